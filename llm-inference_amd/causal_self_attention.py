"""Host-side mirror of the reference's attention module for the MI355X build.

Same public surface as reference causal_self_attention.py, so the reference's own callers
(gpt2.py, transformer_block.py, test_torch_extension.py) read the same against this file:

  ScaledDotProductAttentionBackend      enum, reference :6-26 (the versioned CUDA snapshots
                                        1..12 are a case study of the reference, not part of the path)
  load_attention_backend(backend)       reference :58-92  -- imports the extension module, which
                                        registers torch.ops.causal_multihead_self_attention.*
  scaled_dot_product_attention(...)     reference :95-170 -- backend dispatch
  scaled_dot_product_attention_naive_pytorch(...)   reference :173-194
  CausalSelfAttention                   reference :197-236 -- fused QKV linear, strided q/k/v
                                        views, attention, out-projection

CUSTOM_CUDA keeps its reference name (callers say `ScaledDotProductAttentionBackend.CUSTOM_CUDA`
and `device="cuda"`, which is also how PyTorch-ROCm names HIP devices); CUSTOM_HIP is an alias.
Selecting it runs the hand-written gfx950 kernel or raises -- it never degrades to another backend.
Unlike the reference (TODO at :103) the custom backend also takes batch > 1.
"""
from __future__ import annotations

import importlib
from enum import Enum

import torch
from torch import nn

_EXTENSION_MODULE = "causal_multihead_self_attention"


class ScaledDotProductAttentionBackend(Enum):
    # The custom kernel, the naive formula it is checked against on the same device, and -- so that callers written for
    # the reference (plot_inference_times.py:11, :45) keep working -- the reference's PYTORCH_SDPA_* members (reference
    # :8-11), which select STOCK PyTorch kernels: a second on-device comparator, nothing of this build's.  The reference's
    # CUSTOM_CUDA_VERSION_1..12 members name frozen CUDA snapshots that are not part of this path and have no member here.
    NAIVE_PYTORCH = "NAIVE_PYTORCH"
    CUSTOM_CUDA = "CUSTOM_CUDA"
    CUSTOM_HIP = "CUSTOM_CUDA"      # alias of CUSTOM_CUDA (same value => same member)
    PYTORCH_SDPA_MATH = "PYTORCH_SDPA_MATH"
    PYTORCH_SDPA_FLASH_ATTENTION = "PYTORCH_SDPA_FLASH_ATTENTION"
    PYTORCH_SDPA_EFFICIENT_ATTENTION = "PYTORCH_SDPA_EFFICIENT_ATTENTION"
    PYTORCH_SDPA_CUDNN_ATTENTION = "PYTORCH_SDPA_CUDNN_ATTENTION"


PYTORCH_SDPA_BACKENDS = [b for b in ScaledDotProductAttentionBackend if b.name.startswith("PYTORCH_SDPA_")]      # reference :28-31


def load_attention_backend(backend: ScaledDotProductAttentionBackend) -> None:
    """Import the extension for the custom backend (its import registers the op); no-op otherwise."""
    if backend is ScaledDotProductAttentionBackend.CUSTOM_CUDA:
        importlib.import_module(_EXTENSION_MODULE)   # ImportError propagates: no silent fallback


def _split_heads(t: torch.Tensor, num_heads: int) -> torch.Tensor:
    b, n, d = t.shape
    return t.view(b, n, num_heads, d // num_heads).transpose(1, 2)     # (B, H, T, head_dim)


def _merge_heads(t: torch.Tensor) -> torch.Tensor:
    b, h, n, hd = t.shape
    return t.transpose(1, 2).contiguous().view(b, n, h * hd)


def scaled_dot_product_attention_naive_pytorch(queries: torch.Tensor,
                                               keys: torch.Tensor,
                                               values: torch.Tensor,
                                               num_heads: int,
                                               causal_mask: torch.Tensor) -> torch.Tensor:
    """The reference's plain formula (reference :173-194): mask with -inf, THEN divide by
    sqrt(head_dim), softmax over keys, weight the values.  (B, T, d_model) in and out."""
    if queries.shape[-1] % num_heads != 0:
        raise ValueError("Query dimensionality must be evenly divisible by number of heads.")
    n = queries.shape[1]
    q, k, v = (_split_heads(t, num_heads) for t in (queries, keys, values))
    scores = q @ k.transpose(2, 3)
    scores.masked_fill_(causal_mask.bool()[:n, :n], -torch.inf)
    weights = torch.softmax(scores / (q.shape[-1] ** 0.5), dim=3)
    return _merge_heads(weights @ v)


def _custom_attention(queries, keys, values, num_heads):
    op = torch.ops.causal_multihead_self_attention.causal_multihead_self_attention_torch
    if queries.shape[0] == 1:
        # the reference's exact call: 2-D views in, batch dim restored on the way out (reference :104-106, :144)
        return op(Q=queries.squeeze(0), K=keys.squeeze(0), V=values.squeeze(0), num_heads=num_heads).unsqueeze(0)
    return op(Q=queries, K=keys, V=values, num_heads=num_heads)


def scaled_dot_product_attention(backend: ScaledDotProductAttentionBackend,
                                 queries: torch.Tensor,
                                 keys: torch.Tensor,
                                 values: torch.Tensor,
                                 num_heads: int,
                                 causal_mask: torch.Tensor) -> torch.Tensor:
    """Dispatch on `backend` (reference :95-170).  All tensors are (B, T, d_model); the custom
    kernel derives causality from indices and ignores `causal_mask` (the reference's test passes None,
    test_torch_extension.py:19)."""
    if backend is ScaledDotProductAttentionBackend.NAIVE_PYTORCH:
        return scaled_dot_product_attention_naive_pytorch(queries=queries, keys=keys, values=values,
                                                          num_heads=num_heads, causal_mask=causal_mask)
    if backend is ScaledDotProductAttentionBackend.CUSTOM_CUDA:
        return _custom_attention(queries, keys, values, num_heads)
    if backend in PYTORCH_SDPA_BACKENDS:
        # torch.nn.functional.scaled_dot_product_attention restricted to one of torch's own kernels (reference :113-119,
        # :127-142); whether a kernel exists for fp32 on this device is torch's business (it raises if none does)
        from torch.nn.attention import SDPBackend, sdpa_kernel

        q, k, v = (_split_heads(t, num_heads) for t in (queries, keys, values))
        with sdpa_kernel(getattr(SDPBackend, backend.name[len("PYTORCH_SDPA_"):])):
            return _merge_heads(torch.nn.functional.scaled_dot_product_attention(query=q, key=k, value=v, is_causal=True))
    raise ValueError("Backend not implemented yet")


class CausalSelfAttention(nn.Module):
    """Fused QKV projection -> causal attention -> output projection (reference :197-236).
    Parameter/buffer names match the reference so its state_dicts load unchanged."""

    def __init__(self,
                 d_in: int,
                 d_out: int,
                 max_context_length: int,
                 p_dropout: float,
                 num_heads: int,
                 qkv_bias: bool,
                 scaled_dot_product_attention_backend: ScaledDotProductAttentionBackend,
                 packed_qkv: bool = False):
        super().__init__()
        if d_out % num_heads != 0:
            raise ValueError(f"d_out must be divisible by num_heads. d_out was {d_out} and num_heads was {num_heads}.")
        load_attention_backend(scaled_dot_product_attention_backend)
        self.d_out = d_out
        self.num_heads = num_heads
        self.head_dim = d_out // num_heads
        self.scaled_dot_product_attention_backend = scaled_dot_product_attention_backend
        # opt-in, custom backend only: hand the QKV GEMM's output to the kernel as ONE buffer
        # instead of three column views -- same kernel, same result bits, fewer host-side marshals per call
        self.packed_qkv = packed_qkv and scaled_dot_product_attention_backend is ScaledDotProductAttentionBackend.CUSTOM_CUDA
        self.qkv_layer = nn.Linear(d_in, 3 * d_out, bias=qkv_bias)
        self.out_proj_layer = nn.Linear(d_out, d_out)
        self.dropout = nn.Dropout(p=p_dropout)      # defined but never applied, as in the reference (:216, :222-236)
        self.register_buffer("causal_mask", torch.triu(torch.ones(max_context_length, max_context_length), diagonal=1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        qkv = self.qkv_layer(x)
        if self.packed_qkv:
            return self.out_proj_layer(torch.ops.causal_multihead_self_attention.causal_multihead_self_attention_packed_qkv(
                QKV=qkv, num_heads=self.num_heads))
        # three column slices of ONE buffer: row stride 3*d_out, handed over as views (reference :225-227)
        q, k, v = (qkv[:, :, i * self.d_out:(i + 1) * self.d_out] for i in range(3))
        ctx = scaled_dot_product_attention(backend=self.scaled_dot_product_attention_backend,
                                           queries=q, keys=k, values=v,
                                           num_heads=self.num_heads, causal_mask=self.causal_mask)
        return self.out_proj_layer(ctx)

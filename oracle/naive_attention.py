"""CPU ORACLE (test infrastructure only -- never imported by the product path).

A plain restatement of the reference's naive causal multi-head self-attention,
`scaled_dot_product_attention_naive_pytorch` (reference causal_self_attention.py:173-194):

    view (B,T,H*64) -> (B,H,T,64)                      (:182-184)
    scores = Q @ K^T                                   (:187)
    scores[j > i] = -inf   (mask = triu(ones, 1))      (:188, mask built at :218-219)
    weights = softmax(scores / sqrt(head_dim), dim=-1) (:189)
    ctx = weights @ V, heads merged back to (B,T,H*64) (:191-193)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module.  It is the checker, not the thing shipped: the product path
(`llm-inference_amd/`) fails loudly when the HIP library is missing and never falls back here.

Parity pin: checked equal (bit-for-bit in fp32 for the torch form, <=2e-7 for the fp64 numpy
form) against the reference module itself, imported from /root/reference in the authoring
container by `tests/golden/make_golden.py`, which also wrote the golden vectors in
`tests/golden/`.  The reference holds no golden vectors of its own for this path
(its only pin is the tolerance test test_torch_extension.py:11-22).

Three forms:
  * `naive_attention_torch`  -- fp32 torch-CPU ops in the reference's own operation order;
                                this is "the reference's CPU path" that bench.py times.
  * `naive_attention_f64`    -- numpy float64, used as the high-precision yardstick.
  * `naive_attention_c`      -- ctypes wrapper over oracle/naive_attention.c (plain C fp32).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def causal_mask(context_length: int):
    """triu(ones(T,T), diagonal=1): 1 where key j > query i (reference causal_self_attention.py:218-219)."""
    import torch

    return torch.triu(torch.ones(context_length, context_length), diagonal=1)


def naive_attention_torch(queries, keys, values, num_heads: int):
    """fp32 torch restatement, same op order as reference causal_self_attention.py:178-194.

    queries/keys/values: (B, T, H*head_dim) torch tensors on any device (CPU in practice).
    """
    import torch

    B, T, D = queries.shape
    if D % num_heads != 0:
        raise ValueError("Query dimensionality must be evenly divisible by number of heads.")
    hd = D // num_heads
    q = queries.reshape(B, T, num_heads, hd).transpose(1, 2)
    k = keys.reshape(B, T, num_heads, hd).transpose(1, 2)
    v = values.reshape(B, T, num_heads, hd).transpose(1, 2)
    scores = q @ k.transpose(2, 3)
    masked = torch.triu(torch.ones(T, T, device=queries.device), diagonal=1).bool()
    scores = scores.masked_fill(masked, float("-inf"))
    weights = torch.softmax(scores / (hd ** 0.5), dim=3)
    ctx = weights @ v
    return ctx.transpose(1, 2).contiguous().view(B, T, D)


def naive_attention_f64(queries: np.ndarray, keys: np.ndarray, values: np.ndarray, num_heads: int) -> np.ndarray:
    """float64 numpy restatement of the same formula; inputs (B,T,D) or (T,D) array-likes."""
    q = np.asarray(queries, dtype=np.float64)
    k = np.asarray(keys, dtype=np.float64)
    v = np.asarray(values, dtype=np.float64)
    squeeze = q.ndim == 2
    if squeeze:
        q, k, v = q[None], k[None], v[None]
    B, T, D = q.shape
    if D % num_heads != 0:
        raise ValueError("Query dimensionality must be evenly divisible by number of heads.")
    hd = D // num_heads
    out = np.empty((B, T, D), dtype=np.float64)
    future = np.triu(np.ones((T, T), dtype=bool), k=1)
    for b in range(B):
        for h in range(num_heads):
            sl = slice(h * hd, (h + 1) * hd)
            s = q[b, :, sl] @ k[b, :, sl].T
            s[future] = -np.inf
            s = s / (hd ** 0.5)
            s -= s.max(axis=1, keepdims=True)
            w = np.exp(s)
            w /= w.sum(axis=1, keepdims=True)
            out[b, :, sl] = w @ v[b, :, sl]
    return out[0] if squeeze else out


_clib = None


def _load_c():
    global _clib
    if _clib is None:
        path = os.path.join(_HERE, "libnaive_attention.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.oracle_naive_attention_f32.restype = ctypes.c_int
        lib.oracle_naive_attention_f32.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ]
        _clib = lib
    return _clib


def naive_attention_c(queries: np.ndarray, keys: np.ndarray, values: np.ndarray, num_heads: int,
                      threads: int = 1) -> np.ndarray:
    """Plain-C fp32 restatement (oracle/naive_attention.c). Inputs (B,T,D) float32 C-contiguous."""
    q = np.ascontiguousarray(queries, dtype=np.float32)
    k = np.ascontiguousarray(keys, dtype=np.float32)
    v = np.ascontiguousarray(values, dtype=np.float32)
    squeeze = q.ndim == 2
    if squeeze:
        q, k, v = q[None], k[None], v[None]
    B, T, D = q.shape
    if D % num_heads != 0:
        raise ValueError("Query dimensionality must be evenly divisible by number of heads.")
    out = np.empty((B, T, D), dtype=np.float32)
    rc = _load_c().oracle_naive_attention_f32(
        q.ctypes.data, k.ctypes.data, v.ctypes.data, out.ctypes.data,
        D, D, D, D, T * D, T * D, T * D, T * D,
        B, T, num_heads, D // num_heads, threads)
    if rc != 0:
        raise RuntimeError(f"oracle_naive_attention_f32 failed: {rc}")
    return out[0] if squeeze else out


def decode_step_f64(q, K, V, num_heads: int, head_dim: int = 64):
    """Last row of the reference formula (causal_self_attention.py:173-194) for ONE new token: q (d_model,), K and V
    (n, d_model) = every key/value up to and including the new token's own.  float64 numpy.  What a KV-cache decode
    step must return; the reference itself never computes a single row."""
    import numpy as np

    q = np.asarray(q, dtype=np.float64); K = np.asarray(K, dtype=np.float64); V = np.asarray(V, dtype=np.float64)
    out = np.empty(num_heads * head_dim)
    for h in range(num_heads):
        sl = slice(h * head_dim, (h + 1) * head_dim)
        s = K[:, sl] @ q[sl] / np.sqrt(head_dim)
        w = np.exp(s - s.max()); w /= w.sum()
        out[sl] = w @ V[:, sl]
    return out

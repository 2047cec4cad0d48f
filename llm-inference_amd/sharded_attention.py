"""Multi-GPU forms of the attention path; one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The path shards naturally -- every (batch, head, query block) is independent -- so there are exactly two forms:

  A. data-parallel over prompts / batch rows: no collective on the data path (gpt2_generation.shard_prompts,
     bench.py --gpus N);
  B. head-sharded (BASELINE config 5): rank g owns heads [g*H/W, (g+1)*H/W) -- column slabs of Q, K, V -- computes its
     slab of the context vectors, and ONE all-gather of the (B, T, (H/W)*64) slabs rebuilds (B, T, H*64).

`compute` is the single-device attention: `cmhsa.attention_fwd` on a GPU.  The CPU tests pass the oracle instead;
this module itself never falls back to anything.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


def all_gather_dim0(t: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """ONE all-gather of equal contiguous tensors into (world * t.shape[0], ...).  RCCL ("nccl") moves device memory
    over xGMI directly.  gloo is a host backend (CPU tests, and rehearsals of the N > 1 code path on a one-GPU box):
    device tensors are staged through host memory here, explicitly -- gloo's own handling of device tensors stalls
    after a few dozen collectives on this stack."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        flat = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(flat, host, group=group)
        return flat.to(t.device)
    flat = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(flat, t, group=group)
    return flat


def head_slab(t: torch.Tensor, num_heads: int, rank: int, world: int, head_dim: int = 64) -> torch.Tensor:
    """This rank's heads of a (..., H*head_dim) tensor, as a strided VIEW (the kernel takes row strides)."""
    if num_heads % world != 0:
        raise ValueError(f"num_heads ({num_heads}) must be divisible by the number of ranks ({world}); "
                         "use data-parallel sharding otherwise (GPT2-XL has 25 heads)")
    per = num_heads // world
    return t[..., rank * per * head_dim:(rank + 1) * per * head_dim]


_side_streams: dict = {}


def _side_stream(device: torch.device) -> "torch.cuda.Stream":
    """One extra HIP stream per device for the collectives and copies that overlap the attention kernels."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def head_sharded_attention(Q: torch.Tensor, K: torch.Tensor, V: torch.Tensor, num_heads: int,
                           compute: Callable[[torch.Tensor, torch.Tensor, torch.Tensor, int], torch.Tensor],
                           group=None, head_dim: int = 64, gather: bool = True, chunks: int = 1):
    """Q, K, V: full (B, T, H*head_dim) tensors present on every rank (or at least this rank's head columns valid).
    Returns the full (B, T, H*head_dim) context on every rank (gather=True), else this rank's slab.

    chunks = 1: ONE all-gather into a (W, B, T, slab) buffer (contiguous per-rank slabs), which is then viewed back to
    (B, T, H*head_dim); slabs are column blocks of the output, so the permute is a strided copy.

    chunks > 1 (gather only): the batch is cut into that many pieces and the pieces are pipelined -- the all-gather and
    the permute of piece i run on a second HIP stream while the kernel computes piece i+1 on the caller's stream (over
    xGMI the gather of config 5's slabs takes longer than computing them, so end to end this approaches the gather time
    alone).  The result is ordered on the caller's stream like any other torch op."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = num_heads // world
    q, k, v = (head_slab(t, num_heads, rank, world, head_dim) for t in (Q, K, V))
    if gather and chunks > 1 and Q.dim() == 3 and Q.shape[0] > 1:
        return _head_sharded_pipelined(q, k, v, per, compute, world, group, min(chunks, Q.shape[0]))
    slab = compute(q, k, v, per)                                    # (B, T, per*head_dim), contiguous
    if not gather:
        return slab
    slab = slab.contiguous()
    # output = the W slabs concatenated along dim 0 (the layout every backend's all-gather-into-tensor accepts)
    flat = all_gather_dim0(slab, world, group)
    buf = flat.view((world,) + tuple(slab.shape))
    # (W, B, T, per*hd) -> (B, T, W, per*hd) -> (B, T, H*hd)
    return buf.movedim(0, -2).reshape(*slab.shape[:-1], world * slab.shape[-1])


def _head_sharded_pipelined(q, k, v, per, compute, world, group, chunks):
    B, T, width = q.shape
    out = torch.empty((B, T, world * width), dtype=q.dtype, device=q.device)
    base, extra = divmod(B, chunks)
    on_gpu = q.is_cuda
    if on_gpu:
        main = torch.cuda.current_stream(q.device)
        side = _side_stream(q.device)
        side.wait_stream(main)                                      # `out` (allocated on main) and earlier users of the side stream
        out.record_stream(side)
    lo = 0
    for c in range(chunks):
        hi = lo + base + (1 if c < extra else 0)
        slab = compute(q[lo:hi], k[lo:hi], v[lo:hi], per).contiguous()

        def gather_and_place(slab=slab, lo=lo, hi=hi):
            flat = all_gather_dim0(slab, world, group)               # (W * b, T, width)
            out[lo:hi].view(hi - lo, T, world, width).copy_(flat.view(world, hi - lo, T, width).movedim(0, 2))

        if on_gpu:
            side.wait_stream(main)                                   # this piece's kernel
            slab.record_stream(side)
            with torch.cuda.stream(side):
                gather_and_place()
        else:
            gather_and_place()
        lo = hi
    if on_gpu:
        main.wait_stream(side)
    return out


def batch_sharded_attention(Q: torch.Tensor, K: torch.Tensor, V: torch.Tensor, num_heads: int,
                            compute: Callable[[torch.Tensor, torch.Tensor, torch.Tensor, int], torch.Tensor],
                            group=None, gather: bool = False):
    """Data-parallel over batch rows: rank r takes a contiguous slice of the batch; no collective unless the caller
    asks for the gathered result (uneven slices are padded to the largest for the gather)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B = Q.shape[0]
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    out = compute(Q[lo:hi], K[lo:hi], V[lo:hi], num_heads) if hi > lo else Q.new_empty((0,) + tuple(Q.shape[1:]))
    if not gather:
        return out
    cap = base + (1 if extra else 0)
    pad = Q.new_zeros((cap,) + tuple(Q.shape[1:]))
    pad[:hi - lo] = out
    flat = all_gather_dim0(pad, world, group)
    buf = flat.view((world,) + tuple(pad.shape))
    parts = [buf[r, :base + (1 if r < extra else 0)] for r in range(world)]
    return torch.cat(parts, dim=0)

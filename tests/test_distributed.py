"""world_size-2 gloo tests (CPU) of the multi-GPU forms of the path: head-sharded attention with one
all-gather (BASELINE config 5) and data-parallel batch/prompt sharding (config 4).  The single-device compute is the
CPU oracle here; on the GPU box the same functions run with cmhsa.attention_fwd over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, out_dir):
    for p in (ROOT, os.path.join(ROOT, "llm-inference_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sharded_attention as sa
        from oracle import naive_attention as oracle

        compute = lambda q, k, v, h: oracle.naive_attention_torch(q.contiguous(), k.contiguous(), v.contiguous(), h)
        g = torch.Generator().manual_seed(123)                      # same inputs on every rank
        B, T, H = 3, 70, 4
        x = torch.randn(B, T, 3 * 64 * H, generator=g)
        Q, K, V = x[..., :256], x[..., 256:512], x[..., 512:]
        want = oracle.naive_attention_torch(Q.contiguous(), K.contiguous(), V.contiguous(), H)
        if fn_name == "head":
            slab = sa.head_slab(Q, H, rank, world)
            assert slab.shape == (B, T, 128) and slab.stride(1) == 768          # a strided view, not a copy
            got = sa.head_sharded_attention(Q, K, V, H, compute)
            mine = sa.head_sharded_attention(Q, K, V, H, compute, gather=False)
            assert torch.equal(mine, got[..., rank * 128:(rank + 1) * 128])
        elif fn_name == "batch":
            got = sa.batch_sharded_attention(Q, K, V, H, compute, gather=True)
            local = sa.batch_sharded_attention(Q, K, V, H, compute)
            assert local.shape[0] == (2 if rank == 0 else 1)                     # 3 rows over 2 ranks
        elif fn_name == "bad_heads":
            try:
                sa.head_sharded_attention(Q[..., :192], K[..., :192], V[..., :192], 3, compute)
                got = None
            except ValueError:
                got = want
        err = float((got - want).abs().max())
        np.save(os.path.join(out_dir, f"err_{fn_name}_{rank}.npy"), np.asarray(err))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fn_name", ["head", "batch", "bad_heads"])
def test_two_rank_sharding_matches_single_device(fn_name, tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, fn_name, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert float(np.load(tmp_path / f"err_{fn_name}_{r}.npy")) <= 1e-6


def test_bench_refuses_mismatched_world_size():
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env={**os.environ, "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)

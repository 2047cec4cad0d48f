"""One-GPU look at sharded_attention.head_sharded_attention(chunks=c): world 1 over RCCL, so the all-gather is a device copy
and what shows is the cost of computing the batch in pieces (plans of B/c rows) plus the second stream's bookkeeping.
    python tools/run_pipelined.py > gpurun_out/pipelined_one_gpu.log
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference_amd"))
import cmhsa  # noqa: E402
import sharded_attention as sa  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ws = cmhsa.Workspace(dev)
compute = lambda q, k, v, h: cmhsa.attention_fwd(q, k, v, h, workspace=ws)
for B, H, T in ((32, 2, 4096), (32, 4, 4096), (32, 16, 1024)):
    d = 64 * H
    torch.manual_seed(1)
    x = torch.rand(B, T, 3 * d, device=dev)
    Q, K, V = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
    whole = sa.head_sharded_attention(Q, K, V, H, compute)
    for c in (1, 2, 4, 8):
        fn = lambda: sa.head_sharded_attention(Q, K, V, H, compute, chunks=c)
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            out = fn()
        b.record()
        torch.cuda.synchronize()
        print(f"B{B} H{H} T{T} chunks={c}: {a.elapsed_time(b) / 10:.3f} ms per call, max |diff| vs one gather {float((out - whole).abs().max()):.1e}", flush=True)
dist.destroy_process_group()

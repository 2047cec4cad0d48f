"""Few-row linear kernel, one row: time per call with the weights resident in the 256 MB Infinity Cache (12 weight sets in
rotation, 85-113 MB) against weights that cannot be (60 sets, 425-566 MB: every call streams from HBM) -- the kernel is
one memory round trip, so where the bytes come from is most of its duration."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference_amd"))
import causal_multihead_self_attention  # noqa: E402,F401

dev = torch.device("cuda:0")
op = torch.ops.causal_multihead_self_attention.linear_rows
for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc", 3072, 768), ("fc2", 768, 3072)):
    x = torch.randn(1, K, device=dev)
    b = torch.randn(N, device=dev)
    for sets in (12, 60):
        ws = [torch.randn(N, K, device=dev) * 0.02 for _ in range(sets)]
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for w in ws[:3]:
                op(x, w, b, 0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for w in ws:
                op(x, w, b, 0)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (10 * sets)
        print(f"{name:5s} {N} x {K}, {sets} weight sets ({sets * N * K * 4 / 1e6:.0f} MB): {us:.2f} us per call, {N * K * 4 / us / 1e3:.0f} GB/s", flush=True)
        del ws, g

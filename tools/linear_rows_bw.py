"""Few-row linear kernel (include/gpt2_linear.h) as an HBM kernel: us per call and GB/s of algorithmic bytes for the GPT-2
layer shapes at 1 row, beside torch.nn.functional.linear (bias copy + hipBLASLt).  HIP-graph replays, GPU time only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llm-inference_amd"))
import causal_multihead_self_attention  # noqa: E402,F401
import cmhsa  # noqa: E402

dev = torch.device("cuda:0")
op = torch.ops.causal_multihead_self_attention.linear_rows


def graph_us(fn, reps=50):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(5):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (4 * reps)


for d in (768, 1600):
    for name, N, K in (("qkv", 3 * d, d), ("proj", d, d), ("fc", 4 * d, d), ("fc2", d, 4 * d), ("lm_head", 50257, d)):
        # several weight sets in rotation, as in a model (each layer has its own): no reuse out of the caches beyond what a step has
        ws = [torch.randn(N, K, device=dev) * 0.02 for _ in range(max(2, min(12, int(400e6 / (N * K * 4)))))]
        b = torch.randn(N, device=dev) if name != "lm_head" else None
        for M in (1, 4, 8):
            x = torch.randn(M, K, device=dev)
            i = [0]

            def mine():
                i[0] = (i[0] + 1) % len(ws)
                return op(x, ws[i[0]], b, 0)

            def stock():
                i[0] = (i[0] + 1) % len(ws)
                return torch.nn.functional.linear(x, ws[i[0]], b)

            t_mine, t_stock = graph_us(mine), graph_us(stock)
            nbytes = cmhsa.load_library().gpt2_linear_rows_algorithmic_bytes(M, N, K, 1 if b is not None else 0)
            print(f"d={d} {name:7s} N={N:6d} K={K:5d} rows={M}: few-row kernel {t_mine:7.2f} us ({nbytes / t_mine / 1e3:7.1f} GB/s)   torch linear {t_stock:7.2f} us", flush=True)

"""Stock fp32 GEMM times (torch.nn.functional.linear = hipBLASLt) for the GPT-2 layer shapes at generation sizes:
python tools/gemm_baseline.py [emb_dim]  -> us per call, TFLOP/s, for M = 64..512."""
import sys
import torch

d = int(sys.argv[1]) if len(sys.argv) > 1 else 768
dev = torch.device("cuda:0")
shapes = (("qkv", 3 * d, d), ("proj", d, d), ("fc", 4 * d, d), ("fc2", d, 4 * d))
for name, N, K in shapes:
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    for M in (64, 128, 256, 384, 512):
        x = torch.randn(M, K, device=dev)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(5):
                torch.nn.functional.linear(x, w, b)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()                      # replayed as a HIP graph: GPU time, not torch's dispatch time
        with torch.cuda.graph(g):
            for _ in range(50):
                y = torch.nn.functional.linear(x, w, b)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        print(f"{name:5s} N={N:5d} K={K:5d} M={M:4d}: {us:7.2f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF", flush=True)

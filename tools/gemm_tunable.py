"""What torch's TunableOp (per-shape choice among every rocBLAS / hipBLASLt fp32 algorithm) buys on the GPT-2 layer
shapes at generation sizes, against the default heuristic choice:
    python tools/gemm_tunable.py [emb_dim] [results.csv]  -> us per call, default vs tuned."""
import sys
import time

import torch
import torch.cuda.tunable as tunable

d = int(sys.argv[1]) if len(sys.argv) > 1 else 768
out_csv = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/tunableop_results.csv"
dev = torch.device("cuda:0")
shapes = (("qkv", 3 * d, d), ("proj", d, d), ("fc", 4 * d, d), ("fc2", d, 4 * d))
Ms = (64, 128, 256, 384, 512)


def graph_us(fn):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(5):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 200


data = {}
for name, N, K in shapes:
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    for M in Ms:
        x = torch.randn(M, K, device=dev)
        data[(name, M)] = (x, w, b)
base = {key: graph_us(lambda: torch.nn.functional.linear(x, w, b)) for key, (x, w, b) in data.items()}
print("default algorithms timed", flush=True)

tunable.set_filename(out_csv)
tunable.set_max_tuning_duration(30)
tunable.set_max_tuning_iterations(20)
tunable.enable(True)
tunable.tuning_enable(True)
t0 = time.perf_counter()
for key, (x, w, b) in data.items():
    t1 = time.perf_counter()
    y = torch.nn.functional.linear(x, w, b)           # tunes this shape
    torch.cuda.synchronize()
    print(f"tuned {key} in {time.perf_counter() - t1:.1f} s", flush=True)
print(f"tuning took {time.perf_counter() - t0:.1f} s", flush=True)
tunable.tuning_enable(False)
for key, (x, w, b) in data.items():
    name, M = key
    N, K = w.shape
    us = graph_us(lambda: torch.nn.functional.linear(x, w, b))
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    err = float((torch.nn.functional.linear(x, w, b).double() - ref).abs().max())
    print(f"{name:5s} N={N:5d} K={K:5d} M={M:4d}: default {base[key]:7.2f} us  tuned {us:7.2f} us ({2.0 * M * N * K / us / 1e6:6.1f} TF)  max err vs f64 {err:.1e}", flush=True)
print("validators:", tunable.get_validators(), flush=True)
for r in tunable.get_results():
    print(r, flush=True)

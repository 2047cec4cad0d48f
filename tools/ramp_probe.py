"""How does the per-launch time of BASELINE config 2 evolve from a cold start?  Blocks of 20 launches between HIP events,
no host synchronisation in between; prints the block times at a few points of the first seconds.  (Measurement aid:
bench.py's device ramp is sized from this.)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "llm-inference_amd")]
import torch
import cmhsa

dev = torch.device("cuda:0")
lib = cmhsa.load_library()
ws = cmhsa.Workspace(dev)
B, T, H = 1, 1024, 12
d = 64 * H
qkv = torch.rand((B, T, 3 * d), device=dev)
Q, K, V = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
out = torch.empty((B, T, d), device=dev)
stream = torch.cuda.current_stream(dev)
a = (Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(), Q.stride(1), K.stride(1), V.stride(1), out.stride(1),
     Q.stride(0), K.stride(0), V.stride(0), out.stride(0), B, T, H, 64, ws.ptr, ws.nbytes, ctypes.c_void_p(stream.cuda_stream))
use_probe = len(sys.argv) > 1 and sys.argv[1] == "probe"
if use_probe:
    ms = float(sys.argv[2])
    t0 = time.perf_counter()
    clk = cmhsa.sustained_shader_clock_mhz(dev, iters=int(ms * 1e-3 * 2.3e9 / 64 / 16 / 2))
    torch.cuda.synchronize()
    print(f"clock probe ran {1e3 * (time.perf_counter() - t0):.1f} ms, clock {clk:.0f} MHz")
if len(sys.argv) > 1 and sys.argv[1] == "mem":
    ms = float(sys.argv[2])
    big = torch.zeros(1 << 26, device=dev)            # 256 MiB
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < ms * 1e-3:
        for _ in range(10):
            big.add_(1.0)
        torch.cuda.synchronize()
        n += 10
    print(f"memory ramp: {n} passes over 256 MiB in {1e3 * (time.perf_counter() - t0):.1f} ms")
nblocks = 1500
evs = [torch.cuda.Event(enable_timing=True) for _ in range(nblocks + 1)]
torch.cuda.synchronize()
evs[0].record(stream)
for i in range(nblocks):
    for _ in range(20):
        lib.cmhsa_fp32_fwd_ws(*a)
    evs[i + 1].record(stream)
torch.cuda.synchronize()
us = [evs[i].elapsed_time(evs[i + 1]) * 1e3 / 20 for i in range(nblocks)]
for i in (0, 1, 2, 3, 5, 8, 12, 20, 35, 60, 100, 200, 400, 700, 1000, 1499):
    print(f"launches {20 * i:6d}..{20 * i + 19:6d}: {us[i]:.2f} us/launch")

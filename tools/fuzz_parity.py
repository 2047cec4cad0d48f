"""Soak test on the GPU box: random shapes, layouts and input distributions through every attention entry point of the C ABI
(cmhsa_fp32_fwd, cmhsa_fp32_fwd_ws, cmhsa_fp32_fwd_noalloc, cmhsa_fp32_fwd_packed_qkv), each output pre-filled with NaN and
compared with the oracle's formula (oracle/naive_attention.py) evaluated in float64; the workspace and the no-allocation
entry points must agree bit for bit.  The pytest suite runs a few dozen seeded shapes; this runs for a given number of
seconds and prints a progress line every 25 shapes.

    python tools/fuzz_parity.py --seconds 480 --seed 1 > gpurun_out/fuzz.log
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "llm-inference_amd"))

import cmhsa  # noqa: E402
from oracle import naive_attention as oracle  # noqa: E402


def draw_shape(rng):
    kind = rng.choice(["single", "edge", "mid", "long", "large", "tiny"], p=[0.3, 0.15, 0.2, 0.1, 0.15, 0.1])
    if kind == "single":                         # the generation loop's range (config 3) and its neighbours
        return 1, int(rng.integers(1, 26)), int(rng.integers(1, 1101))
    if kind == "edge":                           # block boundaries
        return int(rng.integers(1, 4)), int(rng.integers(1, 26)), int(32 * rng.integers(1, 40) + rng.integers(-1, 2))
    if kind == "mid":                            # where the planner moves between families
        return int(rng.integers(2, 33)), int(rng.integers(1, 26)), int(rng.integers(33, 1300))
    if kind == "long":
        return 1, int(rng.integers(1, 17)), int(rng.integers(1100, 6000))
    if kind == "large":
        while True:
            B, H, T = int(rng.integers(8, 129)), int(rng.integers(1, 26)), int(rng.integers(16, 700))
            if B * H * T * T <= 1.5e9:
                return B, H, T
    return int(rng.integers(1, 300)), int(rng.integers(1, 5)), int(rng.integers(1, 64))


def make_inputs(rng, B, H, T, dev):
    d = 64 * H
    layout = rng.choice(["packed", "separate", "padded"])
    dist = rng.choice(["randn", "rand", "sharp", "ramp"], p=[0.5, 0.2, 0.2, 0.1])
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))

    def fill(shape):
        if dist == "rand":
            return torch.rand(shape, generator=g, device=dev)
        x = torch.randn(shape, generator=g, device=dev)
        if dist == "sharp":
            x = x * 3.0                            # scores of +-100 and more: a few keys carry each row
        if dist == "ramp":                        # keys grow with position: the running maximum rises at every unit
            x = x + torch.arange(shape[1], device=dev, dtype=torch.float32)[None, :, None] * 0.01
        return x

    if layout == "packed":
        qkv = fill((B, T, 3 * d))
        return (qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]), layout, dist, qkv
    if layout == "separate":
        return tuple(fill((B, T, d)) for _ in range(3)), layout, dist, None
    pad = 4 * int(rng.integers(1, 9))
    bufs = [fill((B, T + 1, d + pad)) for _ in range(3)]
    return tuple(b[:, :T, :d] for b in bufs), layout, dist, None


def oracle_f64(q, k, v, H):
    return torch.cat([oracle.naive_attention_torch(q[b:b + 1].double(), k[b:b + 1].double(), v[b:b + 1].double(), H)
                      for b in range(q.shape[0])])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = cmhsa.load_library()
    ws = cmhsa.Workspace(dev)
    rng = np.random.default_rng(args.seed)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    t0 = time.perf_counter()
    n, worst, families, noalloc_runs, ws_split_runs = 0, 0.0, {}, 0, 0
    while time.perf_counter() - t0 < args.seconds:
        B, H, T = draw_shape(rng)
        (q, k, v), layout, dist, qkv = make_inputs(rng, B, H, T, dev)
        d = 64 * H
        want = oracle_f64(q, k, v, H)
        scale = max(1.0, float(v.abs().max()))
        # fp32 limit of ANY implementation: a score of magnitude S carries ~sqrt(64) roundings of 2^-24 S, and exp() turns
        # an absolute score error into a relative weight error.  S <= |q| |k| / 8 (per head).
        s_max = float(q.reshape(B, T, H, 64).norm(dim=-1).max() * k.reshape(B, T, H, 64).norm(dim=-1).max()) / 8.0
        tol = scale * (2e-5 + 6e-7 * s_max)
        tag = (B, H, T, layout, dist)
        fam = cmhsa.plan_description(B, T, H, ws.nbytes).split(":")[0]
        families[fam] = families.get(fam, 0) + 1
        outs = {}
        for name in ("plain", "ws"):
            o = torch.full((B, T, d), float("nan"), device=dev)
            cmhsa.attention_fwd(q, k, v, H, out=o, workspace=ws if name == "ws" else None)
            outs[name] = o
        need = lib.cmhsa_fp32_fwd_workspace_bytes(B, T, H)
        ws_split_runs += need > 0
        if need <= cmhsa.NOALLOC_SCRATCH_BYTES:
            o = torch.full((B, T, d), float("nan"), device=dev)
            rc = lib.cmhsa_fp32_fwd_noalloc(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), q.stride(1), k.stride(1), v.stride(1),
                                            o.stride(1), q.stride(0), k.stride(0), v.stride(0), o.stride(0), B, T, H, 64, stream)
            assert rc == 0, (tag, rc)
            outs["noalloc"] = o
            noalloc_runs += 1
        if qkv is not None:
            outs["packed_qkv"] = cmhsa.attention_fwd_packed_qkv(qkv, H, workspace=ws)
        torch.cuda.synchronize(dev)
        for name, o in outs.items():
            assert torch.isfinite(o).all(), (tag, name, "an element was not written or is not finite")
            err = float((o.double() - want).abs().max())
            worst = max(worst, err / scale)
            assert err <= tol, (tag, name, err, fam)
        if "noalloc" in outs:
            assert torch.equal(outs["noalloc"], outs["ws"]), (tag, "noalloc != ws")
        if "packed_qkv" in outs:
            assert torch.equal(outs["packed_qkv"], outs["ws"]), (tag, "packed_qkv != ws")
        n += 1
        del q, k, v, qkv, want, outs
        if n % 25 == 0:
            print(f"[{time.perf_counter() - t0:6.0f} s] {n} shapes ok, worst error / max|V| {worst:.2e}, last {tag}", flush=True)
    print(f"DONE seed {args.seed}: {n} shapes, every entry point within (2e-5 + 6e-7 max|score|) * max|V| of the f64 formula (worst {worst:.2e}); "
          f"{ws_split_runs} plans with split blocks, {noalloc_runs} through the no-allocation entry point", flush=True)
    for fam, c in sorted(families.items(), key=lambda x: -x[1]):
        print(f"  {c:5d}  {fam}", flush=True)


if __name__ == "__main__":
    main()

"""bench.py -- the hot path's headline measurement (BASELINE.json metric: attention fp32 TFLOP/s and
% of MI355X fp32 MFMA peak).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config5_shard] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE pass of the hot path over one batch of synthetic input: one call of the C-ABI entry
point cmhsa_fp32_fwd (one kernel launch) on inputs already resident in HBM.  Default workload =
BASELINE.json configs[1]: standalone causal attention fp32, B=1, H=12, T=1024, d_head=64, inputs as
the model hands them over (QKV column slices of one (T, 3*d_model) buffer).  With N > 1 every rank
runs the same workload on its own GPU (independent prompts, data-parallel, no collective on the
data path): weak scaling, value = aggregate over ranks.

Flop convention: F = 2*B*H*d_head*T*(T+1) -- unmasked pairs only, softmax not counted.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "llm-inference_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: 256 CU x 256 flop/clk/CU x 2.4 GHz (v_mfma_f32_32x32x2_f32)

WORKLOADS = {
    # BASELINE.json configs[1]
    "config2": dict(B=1, H=12, T=1024, desc="standalone causal attention fp32 B=1 H=12 T=1024 d_head=64 (QKV-sliced inputs)"),
    # one GPU's share of BASELINE.json configs[4] (B=32, H=16, T=4096 head-sharded 8 ways -> 2 heads per GPU)
    "config5_shard": dict(B=32, H=2, T=4096, desc="one GPU's head shard of B=32 H=16 T=4096 d_head=64 (2 heads)"),
    # BASELINE.json configs[4] itself: heads split over the N ranks, one RCCL all-gather of the context slabs
    # (strong scaling: total work fixed; N=1 runs all 16 heads on one GPU and gathers nothing)
    "config5": dict(B=32, H=16, T=4096, head_sharded=True,
                    desc="synthetic causal attention fp32 B=32 H=16 T=4096 d_head=64, head-sharded over N GPUs + one all-gather"),
}


def cpu_baseline(wl: dict, budget_s: float = 20.0) -> dict:
    """The reference's CPU path (naive PyTorch formula on CPU tensors, restated in oracle/), timed on
    this host's cores on a bounded sample of the same workload."""
    from oracle import naive_attention as oracle

    cores = os.cpu_count() or 1
    cores = min(cores, 16) if "GRAFT_REPO_ROOT" in os.environ else cores   # the GPU box's CPU share is 16
    torch.set_num_threads(cores)
    B = min(wl["B"], 2)                      # bounded sample: at most 2 batch rows of the workload
    H, T = wl["H"], wl["T"]
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.rand((B, T, 64 * H), generator=g) for _ in range(3))
    oracle.naive_attention_torch(q, k, v, H)             # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 9 and (len(times) < 5 or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        oracle.naive_attention_torch(q, k, v, H)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    flops = 2.0 * B * H * 64 * T * (T + 1)
    return {"value": flops / med / 1e12, "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} runs of naive attention (oracle/naive_attention.py, torch {torch.__version__} CPU ops) "
                      f"at B={B} H={H} T={T}, median {med * 1e3:.2f} ms"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="config2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if os.environ.get("CMHSA_BENCH_WATCHDOG"):               # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["CMHSA_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP attention path has no CPU fallback")
    # rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (never for reported numbers):
    # CMHSA_BENCH_BACKEND=gloo lets several ranks share a card (RCCL refuses two ranks on one device)
    backend = os.environ.get("CMHSA_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    import cmhsa

    lib = cmhsa.load_library()
    wl = WORKLOADS[args.workload]
    B, H, T = wl["B"], wl["H"], wl["T"]
    if backend != "nccl":
        B, T = min(B, 2), min(T, 1024)                   # rehearsal only: keep the host-staged collective small
    sharded = bool(wl.get("head_sharded"))
    if sharded:
        if H % world != 0:
            raise SystemExit(f"config5 needs num_heads ({H}) divisible by the number of GPUs ({world})")
        H = H // world                                   # this rank's heads: a column slab of Q, K, V
    d = 64 * H
    torch.manual_seed(1234 + rank)
    qkv = torch.rand((B, T, 3 * d), device=dev)                      # reference test distribution (test_torch_extension.py:16-18)
    Q, K, V = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]      # row stride 3*d_model, as in the model
    out = torch.empty((B, T, d), device=dev)
    stream = torch.cuda.current_stream(dev)
    call_args = (Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(),
                 Q.stride(1), K.stride(1), V.stride(1), out.stride(1),
                 Q.stride(0), K.stride(0), V.stride(0), out.stride(0),
                 B, T, H, 64, ctypes.c_void_p(stream.cuda_stream))
    fwd = lib.cmhsa_fp32_fwd

    gathered = torch.empty((world * B, T, d), device=dev) if sharded and world > 1 else None

    def step() -> None:
        rc = fwd(*call_args)
        if rc != 0:
            raise RuntimeError(cmhsa.status_string(rc))
        if gathered is not None:                         # the one collective of the head-sharded form
            dist.all_gather_into_tensor(gathered, out)

    def fence() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)          # same stream the kernel is launched on
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    fence()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)

    clock_mhz = cmhsa.sustained_shader_clock_mhz(dev)       # outside the timed region; SURVEY 8(d): print the clock
    # parity spot-check outside the timed region: one (batch, head) against the f64 oracle
    from oracle import naive_attention as oracle
    sl = slice(64 * (H - 1), 64 * H)
    want = oracle.naive_attention_f64(Q[B - 1, :, sl].cpu().numpy(), K[B - 1, :, sl].cpu().numpy(), V[B - 1, :, sl].cpu().numpy(), 1)
    max_err = float(abs(out[B - 1, :, sl].cpu().numpy() - want).max())

    breakdown = None
    if gathered is not None:
        # compute-only rate per GPU and the all-gather on its own (outside the timed region)
        def span(fn, reps=5):
            fn()
            fence()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(reps):
                fn()
            b.record(stream)
            fence()
            return a.elapsed_time(b) / reps
        comp_ms = span(lambda: fwd(*call_args))
        ag_ms = span(lambda: dist.all_gather_into_tensor(gathered, out))
        t2 = torch.tensor([comp_ms, ag_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        comp_ms, ag_ms = float(t2[0]), float(t2[1])
        recv_bytes = (world - 1) * out.numel() * 4
        breakdown = {"compute_only_ms": round(comp_ms, 4),
                     "compute_only_tflops_per_gpu": round(cmhsa.algorithmic_flops(B, T, H, 64) / (comp_ms * 1e-3) / 1e12, 2),
                     "all_gather_ms": round(ag_ms, 4), "all_gather_received_bytes_per_rank": recv_bytes,
                     "all_gather_GBps_received_per_rank": round(recv_bytes / (ag_ms * 1e-3) / 1e9, 1),
                     "all_gather_GBps_per_xgmi_link": round(recv_bytes / (world - 1) / (ag_ms * 1e-3) / 1e9, 1)}
    if world > 1:
        tmax = torch.tensor([wall, dev_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(tmax[0]), float(tmax[1])

    flops = cmhsa.algorithmic_flops(B, T, H, 64)              # per rank
    ms_per_step = wall * 1e3 / args.steps
    value = world * flops / (wall / args.steps) / 1e12
    launch_us = dev_ms * 1e3 / args.steps
    achieved = flops / (dev_ms * 1e-3 / args.steps) / 1e12

    # HBM traffic per launch from the committed rocprofv3 counter passes of this same workload
    # (tools/prof_pmc.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, KiB -> bytes)
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", f"pmc_{args.workload}_summary.json")
    if os.path.exists(pmc_path):
        with open(pmc_path) as f:
            pmc = json.load(f)
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            traffic = int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)

    # secondary evidence (N=1 only, outside the timed region): the throughput-regime shape
    others = {}
    if world == 1 and args.workload == "config2":
        for name in ("config5_shard",):
            o = WORKLOADS[name]
            x = torch.rand((o["B"], o["T"], 3 * 64 * o["H"]), device=dev)
            dd = 64 * o["H"]
            oq, ok, ov = x[..., :dd], x[..., dd:2 * dd], x[..., 2 * dd:]
            oo = torch.empty((o["B"], o["T"], dd), device=dev)
            a = (oq.data_ptr(), ok.data_ptr(), ov.data_ptr(), oo.data_ptr(), oq.stride(1), ok.stride(1), ov.stride(1), oo.stride(1),
                 oq.stride(0), ok.stride(0), ov.stride(0), oo.stride(0), o["B"], o["T"], o["H"], 64, ctypes.c_void_p(stream.cuda_stream))
            for _ in range(3):
                fwd(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                fwd(*a)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) * 1e3 / 10
            tf = cmhsa.algorithmic_flops(o["B"], o["T"], o["H"], 64) / us / 1e6
            others[name] = {"workload": o["desc"], "us_per_launch": round(us, 2), "tflops": round(tf, 2),
                            "frac_of_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4)}
            del x, oo
        # SURVEY 8(d) config 2 asks for more than the headline: contiguous as well as QKV-sliced inputs, randn as well
        # as rand, and the latency of ONE launch (launch -> completion seen by the host) next to the back-to-back rate
        def timed(fn, reps):
            for _ in range(5):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(reps):
                fn()
            b.record(stream)
            torch.cuda.synchronize(dev)
            return a.elapsed_time(b) * 1e3 / reps

        variants = {}
        for vname, maker, sliced in (("contiguous_rand", torch.rand, False), ("qkv_sliced_randn", torch.randn, True), ("contiguous_randn", torch.randn, False)):
            if sliced:
                x = maker((B, T, 3 * d), device=dev)
                vq, vk, vv = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
            else:
                vq, vk, vv = (maker((B, T, d), device=dev) for _ in range(3))
            va = (vq.data_ptr(), vk.data_ptr(), vv.data_ptr(), out.data_ptr(), vq.stride(1), vk.stride(1), vv.stride(1), out.stride(1),
                  vq.stride(0), vk.stride(0), vv.stride(0), out.stride(0), B, T, H, 64, ctypes.c_void_p(stream.cuda_stream))
            variants[vname + "_us"] = round(timed(lambda: fwd(*va), 200), 3)
        lat = []
        for _ in range(50):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            fwd(*call_args)
            torch.cuda.synchronize(dev)
            lat.append((time.perf_counter() - t1) * 1e6)
        variants["single_launch_host_latency_us_median"] = round(statistics.median(lat), 2)
        others["config2_variants"] = variants
        # the fused tanh-GELU kernel: HBM-bound, 8 algorithmic bytes per element
        for name, n in (("gelu_gpt2xl_b64_t512", 64 * 512 * 6400), ("gelu_gpt2_t256", 256 * 3072)):
            gx = torch.randn(n, device=dev)
            gy = torch.empty_like(gx)
            ga = (gx.data_ptr(), gy.data_ptr(), n, ctypes.c_void_p(stream.cuda_stream))
            for _ in range(3):
                lib.gpt2_gelu_tanh_f32_fwd(*ga)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record(stream)
            for _ in range(reps):
                lib.gpt2_gelu_tanh_f32_fwd(*ga)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) * 1e3 / reps
            gbs = lib.gpt2_gelu_algorithmic_bytes(n) / us / 1e3
            others[name] = {"workload": f"fused tanh-GELU fp32, {n} elements", "us_per_launch": round(us, 2),
                            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                                         "frac": round(gbs / 8000.0, 4), "traffic": None}}
            del gx, gy
        # the KV-cache decode step (opt-in): HBM/L2-bound, reads every cached K and V row once
        for name, (db, dh, keys) in (("decode_b64_h25_k512", (64, 25, 512)), ("decode_b1_h12_k512", (1, 12, 512))):
            dd = 64 * dh
            kc = torch.randn((db, keys, dd), device=dev)
            vc = torch.randn((db, keys, dd), device=dev)
            new = torch.randn((db, 3 * dd), device=dev)
            do = torch.empty((db, dd), device=dev)
            pos = torch.tensor([keys - 1], dtype=torch.int32, device=dev)
            da = (new.data_ptr(), new.stride(0), kc.data_ptr(), vc.data_ptr(), kc.stride(1), kc.stride(0), do.data_ptr(), do.stride(0),
                  pos.data_ptr(), db, keys, dh, 64, ctypes.c_void_p(stream.cuda_stream))
            for _ in range(3):
                lib.cmhsa_fp32_decode_step(*da)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            e0.record(stream)
            for _ in range(reps):
                lib.cmhsa_fp32_decode_step(*da)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) * 1e3 / reps
            gbs = lib.cmhsa_decode_algorithmic_bytes(db, keys, dh, 64) / us / 1e3
            others[name] = {"workload": f"KV-cache decode step fp32, B={db} H={dh}, {keys} keys", "us_per_launch": round(us, 2),
                            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                                         "frac": round(gbs / 8000.0, 4), "traffic": None}}
            del kc, vc
    result = {
        "metric": "attention fp32 TFLOP/s (causal MHSA forward, F = 2*B*H*d*T*(T+1))",
        "value": round(value, 3),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 6),
        "higher_is_better": True,
        "scaling": "strong" if sharded else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (torch.rand U[0,1), seeded), inputs resident in HBM",
        "config": {"workload": wl["desc"], "batch_per_gpu": B, "num_heads": H, "seq_len": T, "head_dim": 64,
                   "parallelism": (f"heads/{world} per GPU + one all-gather of (B,T,{d}) fp32 slabs" if sharded else
                                   f"dp{world} (independent prompts per GPU, no data-path collective)")},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                     "kernel": "cmhsa_fwd_kernel", "avg_launch_us": round(launch_us, 3),
                     "flops_per_launch": flops, "bytes_per_launch_algorithmic": 4 * B * T * d * 4,
                     "timing": "HIP events on the launch stream around the timed steps (includes inter-launch gaps)"},
        "percent_of_fp32_mfma_peak": round(100 * achieved / FP32_MFMA_PEAK_TFLOPS, 2),
        "max_abs_err_vs_f64_oracle": max_err,
        "peak_assumes_clock_mhz": 2400,
        "sustained_shader_clock_mhz_under_fp32_mfma_load": round(clock_mhz, 0),
        "device": torch.cuda.get_device_name(dev),
    }
    if backend != "nccl":
        result["data"] += f"; REHEARSAL with backend {backend}: ranks may share a GPU, numbers are not measurements"
    if others:
        result["other_workloads"] = others
    if breakdown:
        result["config5_breakdown"] = breakdown
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(wl)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

"""bench.py -- the hot path's headline measurement (BASELINE.json metric: attention fp32 TFLOP/s and
% of MI355X fp32 MFMA peak; GPT2 512-token generation seconds; 1-8 GPUs).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config5_shard|config5]
                    [--no-cpu-baseline] [--no-extras] [--gen-budget SECONDS]

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts N child ranks of itself
(fresh processes, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, one per GPU), relays rank 0's JSON line and exits with the
worst child status; it never touches the GPU.  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` works as well (the ranks then find WORLD_SIZE set).

A "step" is ONE pass of the hot path over one batch of synthetic input: one call of the C-ABI entry point
cmhsa_fp32_fwd_ws (one kernel launch) on inputs already resident in HBM.  Default workload = BASELINE.json configs[1]:
standalone causal attention fp32, B=1, H=12, T=1024, d_head=64, inputs as the model hands them over (QKV column
slices of one (T, 3*d_model) buffer).  With N > 1 every rank runs the same workload on its own GPU (independent
prompts, data-parallel, no collective on the data path): weak scaling, value = aggregate over ranks.

Timing: a device ramp (below), W warm-up steps, barrier + synchronize, a few untimed priming launches that refill the
launch queue (a host fence drains it, and the ~30 us it takes to fill again would otherwise be billed to the first
timed steps), then EXACTLY K steps between two HIP events on the launch stream, barrier + synchronize.  `value` and
`ms_per_step` come from the events (max over ranks); the host wall clock around the same K steps is reported too.

Device ramp: from idle an MI355X runs this kernel at ~2.18 GHz and raises the shader clock to ~2.4 GHz only after
10-20 ms of sustained launches (tools/ramp_probe.py: 20.9 us per launch for the first ~300 launches, 18.7 us from
~1200 on; in-kernel cycle counts are identical, tools/cmhsa_bench_stamps).  A run of W = 5, K = 20 would measure the
power-management transient, not the kernel, so ~100 ms of the same launches run first (`--ramp-ms`); the JSON line
reports the ramp and the time per launch of its first 100 launches next to the steady-state figure.

Flop convention: F = 2*B*H*d_head*T*(T+1) -- unmasked pairs only, softmax not counted.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "llm-inference_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: 256 CU x 256 flop/clk/CU x 2.4 GHz (v_mfma_f32_32x32x2_f32)
HBM_PEAK_GBPS = 8000.0

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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="config2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip other_workloads, generation and the N>1 config legs")
    ap.add_argument("--ramp-ms", type=float, default=100.0,
                    help="device ramp before the warm-up steps: milliseconds of back-to-back launches of the workload (0 = none)")
    ap.add_argument("--gen-budget", type=float, default=120.0,
                    help="seconds the generation legs (GPT2 512-token runs, config 1 on the CPU, config 4 sample) may take in all")
    ap.add_argument("--rendezvous-timeout", type=float, default=120.0,
                    help="seconds torch.distributed may wait for the other ranks (rendezvous, and every collective under gloo; RCCL "
                         "collectives are additionally watched by --extras-limit): a missing rank fails the run in seconds, not torch's 10-30 min")
    ap.add_argument("--extras-limit", type=float, default=420.0,
                    help="seconds the legs after the headline may take before the line is printed without them (a stalled collective "
                         "or a failed rank must not cost the headline measurement)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: this process starts the ranks and never initialises the GPU itself
# ---------------------------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool support dmabuf IPC only; with the legacy mode RCCL's peer-memory
        # set-up (and any device-tensor sharing between processes) fails with `hipIpcGetMemHandle: invalid argument`.  The
        # harness exports it already; it is repeated here so that ranks started from a shell without it still come up.
        env = {**os.environ, "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(args.gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies (no such device, a failed check) would leave the others waiting in a rendezvous or a barrier for
    # torch's time-out: as soon as one exits non-zero, the rest are ended (these exact child processes, nothing else)
    while True:
        codes = [p.poll() for p in procs]
        if any(c not in (None, 0) for c in codes):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        if all(c is not None for c in codes):
            break
        time.sleep(0.2)
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=20))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append(p.wait())
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    worst = max((abs(c) for c in codes), default=0)
    if worst:
        print(f"bench.py: child ranks exited with {codes}", file=sys.stderr)
    return 1 if worst else 0


# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(wl: dict, torch, budget_s: float = 20.0) -> dict:
    """The reference's CPU path (naive PyTorch formula on CPU tensors, restated in oracle/), timed on
    this host's cores on a bounded sample of the same workload."""
    from oracle import naive_attention as oracle

    cores = host_cores()
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


def host_cores() -> int:
    cores = os.cpu_count() or 1
    return min(cores, 16) if "GRAFT_REPO_ROOT" in os.environ else cores   # the GPU box's CPU share is 16


class Bench:
    """Everything one rank does.  `fwd_args(Q, K, V, out)` builds the C-ABI argument tuple once per workload, so the
    timed loop is nothing but ctypes calls."""

    def __init__(self, args):
        import torch

        self.torch = torch
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: launch N>1 either as `python bench.py --gpus N` "
                             "(it starts its own ranks) or with python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP attention path has no CPU fallback")
        # rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (never for reported numbers):
        # CMHSA_BENCH_BACKEND=gloo lets several ranks share a card (RCCL refuses two ranks on one device)
        self.backend = os.environ.get("CMHSA_BENCH_BACKEND", "nccl")
        if self.backend == "nccl" and local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {local_rank} has no GPU: --gpus {args.gpus} on a host with {torch.cuda.device_count()} "
                             "(one rank per GPU; CMHSA_BENCH_BACKEND=gloo rehearses the code path on fewer cards)")
        dev_index = local_rank % torch.cuda.device_count() if self.backend != "nccl" else local_rank
        torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import datetime

            dist.init_process_group(self.backend, timeout=datetime.timedelta(seconds=args.rendezvous_timeout),
                                    **({"device_id": self.dev} if self.backend == "nccl" else {}))
            self.dist = dist
        import cmhsa

        self.cmhsa = cmhsa
        self.lib = cmhsa.load_library()
        self.stream = torch.cuda.current_stream(self.dev)
        self.workspace = cmhsa.Workspace(self.dev)                  # device scratch of the split plans, zeroed once

    # -- helpers --------------------------------------------------------------------------------------------------
    def fence(self) -> None:
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, *vals: float):
        if self.dist is None:
            return vals
        t = self.torch.tensor(vals, device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return tuple(float(x) for x in t)

    def sliced_inputs(self, B: int, T: int, H: int, maker=None):
        torch = self.torch
        d = 64 * H
        qkv = (maker or torch.rand)((B, T, 3 * d), device=self.dev)    # reference test distribution (test_torch_extension.py:16-18)
        return qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]          # row stride 3*d_model, as in the model

    def fwd_args(self, Q, K, V, out, B, T, H):
        return (Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(),
                Q.stride(1), K.stride(1), V.stride(1), out.stride(1),
                Q.stride(0), K.stride(0), V.stride(0), out.stride(0),
                B, T, H, 64, self.workspace.ptr, self.workspace.nbytes, ctypes.c_void_p(self.stream.cuda_stream))

    def launch(self, a) -> None:
        rc = self.lib.cmhsa_fp32_fwd_ws(*a)
        if rc != 0:
            raise RuntimeError(self.cmhsa.status_string(rc))

    def events_us(self, fn, reps: int, warm: int = 20) -> float:
        """Steady-state time per call: `warm` untimed calls, then `reps` calls between two events, no host fence
        in between (the queue stays full across the boundary)."""
        torch = self.torch
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(self.stream)
        for _ in range(reps):
            fn()
        b.record(self.stream)
        torch.cuda.synchronize(self.dev)
        return a.elapsed_time(b) * 1e3 / reps

    def device_ramp(self, step, ramp_ms: float) -> dict:
        """Back-to-back launches of the workload until `ramp_ms` of device time has passed (see the module docstring)."""
        torch = self.torch
        if ramp_ms <= 0:
            return {"ms": 0.0, "launches": 0}
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        step()                                               # first launch: module load, LDS attribute, plan
        torch.cuda.synchronize(self.dev)
        e0.record(self.stream)
        for _ in range(100):
            step()
        e1.record(self.stream)
        torch.cuda.synchronize(self.dev)
        first_us = e0.elapsed_time(e1) * 1e3 / 100
        n = 101
        t_end = time.perf_counter() + ramp_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(200):
                step()
            n += 200
            # keep the host at most a few hundred launches ahead, without letting the device go idle
            e2.record(self.stream)
            while not e2.query() and time.perf_counter() < t_end:
                pass
        return {"ms": ramp_ms, "launches": n, "us_per_launch_first_100_launches_from_idle": round(first_us, 3),
                "why": "the shader clock rises from ~2.18 to ~2.4 GHz only after 10-20 ms of sustained launches"}

    # -- the headline ---------------------------------------------------------------------------------------------
    def headline(self) -> dict:
        torch, args, world = self.torch, self.args, self.world
        wl = WORKLOADS[args.workload]
        B, H, T = wl["B"], wl["H"], wl["T"]
        if self.backend != "nccl":
            B, T = min(B, 2), min(T, 1024)                   # rehearsal only: keep the host-staged collective small
        sharded = bool(wl.get("head_sharded"))
        if sharded:
            if H % world != 0:
                raise SystemExit(f"config5 needs num_heads ({H}) divisible by the number of GPUs ({world})")
            H = H // world                                   # this rank's heads: a column slab of Q, K, V
        d = 64 * H
        torch.manual_seed(1234 + self.rank)
        Q, K, V = self.sliced_inputs(B, T, H)
        out = torch.empty((B, T, d), device=self.dev)
        call_args = self.fwd_args(Q, K, V, out, B, T, H)
        gathered = torch.empty((world * B, T, d), device=self.dev) if sharded and world > 1 else None
        if gathered is not None and self.backend != "nccl":
            import sharded_attention as sa                   # rehearsal: gloo gathers through host memory

        def step() -> None:
            self.launch(call_args)
            if gathered is not None:                         # the one collective of the head-sharded form
                if self.backend == "nccl":
                    self.dist.all_gather_into_tensor(gathered, out)
                else:
                    sa.all_gather_dim0(out, world)

        ramp = self.device_ramp(step, args.ramp_ms)
        for _ in range(args.warmup):
            step()
        self.fence()
        prime = min(8, max(args.warmup, 1))
        for _ in range(prime):                               # refill the launch queue the fence has drained (untimed)
            step()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(self.stream)                              # same stream the kernel is launched on
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ev1.record(self.stream)
        self.fence()
        wall = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)
        wall, dev_ms = self.max_over_ranks(wall, dev_ms)

        clock_mhz = self.cmhsa.sustained_shader_clock_mhz(self.dev)       # outside the timed region; SURVEY 8(d): print the clock
        # parity spot-check outside the timed region: one (batch, head) against the f64 oracle
        from oracle import naive_attention as oracle
        sl = slice(64 * (H - 1), 64 * H)
        want = oracle.naive_attention_f64(Q[B - 1, :, sl].cpu().numpy(), K[B - 1, :, sl].cpu().numpy(), V[B - 1, :, sl].cpu().numpy(), 1)
        max_err = float(abs(out[B - 1, :, sl].cpu().numpy() - want).max())

        flops = self.cmhsa.algorithmic_flops(B, T, H, 64)              # per rank
        step_s = dev_ms * 1e-3 / args.steps
        achieved = flops / step_s / 1e12

        # HBM traffic per launch from the committed rocprofv3 counter passes of this same workload
        # (tools/prof_all.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, KiB -> bytes)
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, "profiles", f"pmc_{args.workload}_summary.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                traffic = int((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
                traffic_source = f"profiles/pmc_{args.workload}_summary.json (rocprofv3 --pmc passes of this workload, committed; not measured in this run)"

        result = {
            "metric": "attention fp32 TFLOP/s (causal MHSA forward, F = 2*B*H*d*T*(T+1))",
            "value": round(world * achieved, 3),
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ramp_launches": ramp["launches"],               # untimed launches of the device ramp that ran BEFORE the warm-up steps
            "ms_per_step": round(step_s * 1e3, 6),
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (torch.rand U[0,1), seeded), inputs resident in HBM",
            "config": {"workload": wl["desc"], "batch_per_gpu": B, "num_heads": H, "seq_len": T, "head_dim": 64,
                       "parallelism": (f"heads/{world} per GPU + one all-gather of (B,T,{d}) fp32 slabs" if sharded else
                                       f"dp{world} (independent prompts per GPU, no data-path collective)")},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "cmhsa_fwd_kernel", "avg_launch_us": round(step_s * 1e6, 3),
                         "flops_per_launch": flops, "bytes_per_launch_algorithmic": 4 * B * T * d * 4,
                         "timing": "HIP events on the launch stream around the K timed steps, launch queue kept full across "
                                   "the warm-up boundary (includes inter-launch gaps)"},
            "timing": {"value_from": "HIP events on the launch stream (max over ranks)",
                       "wall_ms_per_step_incl_final_fence": round(wall * 1e3 / args.steps, 6),
                       "value_from_wall_clock": round(world * flops / (wall / args.steps) / 1e12, 3),
                       "priming_launches_after_fence": prime, "device_ramp": ramp},
            "percent_of_fp32_mfma_peak": round(100 * achieved / FP32_MFMA_PEAK_TFLOPS, 2),
            "percent_of_fp32_mfma_peak_first_100_launches_from_idle": (
                round(100 * flops / (ramp["us_per_launch_first_100_launches_from_idle"] * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 2)
                if ramp.get("us_per_launch_first_100_launches_from_idle") else None),
            "max_abs_err_vs_f64_oracle": max_err,
            "peak_assumes_clock_mhz": 2400,
            "sustained_shader_clock_mhz_under_fp32_mfma_load": round(clock_mhz, 0),
            "entry_point": "cmhsa_fp32_fwd_ws",
            "plan": self.cmhsa.plan_description(B, T, H, self.workspace.nbytes),
            "device": torch.cuda.get_device_name(self.dev),
        }
        if self.backend != "nccl":
            result["data"] += f"; REHEARSAL with backend {self.backend}: ranks may share a GPU, numbers are not measurements"
        if gathered is not None and not args.no_extras:
            result["config5_breakdown"] = self.config5_breakdown(B, T, wl["H"])
        return result

    # -- secondary kernel evidence (N = 1) ------------------------------------------------------------------------
    def other_workloads(self) -> dict:
        torch, lib = self.torch, self.lib
        others = {}
        o = WORKLOADS["config5_shard"]
        q, k, v = self.sliced_inputs(o["B"], o["T"], o["H"])
        oo = torch.empty((o["B"], o["T"], 64 * o["H"]), device=self.dev)
        a = self.fwd_args(q, k, v, oo, o["B"], o["T"], o["H"])
        us = self.events_us(lambda: self.launch(a), reps=20, warm=20)
        tf = self.cmhsa.algorithmic_flops(o["B"], o["T"], o["H"], 64) / us / 1e6
        others["config5_shard"] = {"workload": o["desc"], "us_per_launch": round(us, 2), "tflops": round(tf, 2),
                                   "frac_of_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "warmup": 20, "reps": 20}
        del q, k, v, oo
        # BASELINE configs[4] WHOLE on this one GPU (B=32, H=16, T=4096: the N = 1 point of the head-sharded config; nothing to
        # gather), with one (batch, head) checked against the f64 oracle outside the timed region
        o = WORKLOADS["config5"]
        q, k, v = self.sliced_inputs(o["B"], o["T"], o["H"])
        oo = torch.empty((o["B"], o["T"], 64 * o["H"]), device=self.dev)
        a = self.fwd_args(q, k, v, oo, o["B"], o["T"], o["H"])
        us = self.events_us(lambda: self.launch(a), reps=10, warm=5)
        tf = self.cmhsa.algorithmic_flops(o["B"], o["T"], o["H"], 64) / us / 1e6
        from oracle import naive_attention as oracle
        bb, hh = o["B"] - 1, o["H"] // 2
        sl = slice(64 * hh, 64 * hh + 64)
        want = oracle.naive_attention_f64(q[bb, :, sl].cpu().numpy(), k[bb, :, sl].cpu().numpy(), v[bb, :, sl].cpu().numpy(), 1)
        others["config5_whole_on_one_gpu"] = {"workload": "synthetic causal attention fp32 B=32 H=16 T=4096 d_head=64, all 16 heads on one GPU",
                                              "us_per_launch": round(us, 2), "tflops": round(tf, 2),
                                              "frac_of_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "warmup": 5, "reps": 10,
                                              "max_abs_err_vs_f64_oracle_one_batch_head": float(abs(oo[bb, :, sl].cpu().numpy() - want).max()),
                                              "plan": self.cmhsa.plan_description(o["B"], o["T"], o["H"], self.workspace.nbytes)}
        del q, k, v, oo, want
        torch.cuda.empty_cache()
        # three more shapes of the path, each on the plan family built for it in round 3: config 4's attention shape on one
        # GPU (GPT2-XL, 64 prompts, T = 512: job lists), the launches of a B = 1 generation at T = 256 / 512 (config 3: the
        # 16-key-tile kernel) and a single long prompt (32-row blocks on the 16-key tiles); 2 000 / 300 warm-up launches
        # because the shader clock needs ~20 ms of load to settle (timing.device_ramp)
        for name, (B, T, H), reps, warm in (("config4_attention_shape_b64_h25_t512", (64, 512, 25), 100, 300),
                                            ("config3_launch_b1_h12_t256", (1, 256, 12), 1000, 2000),
                                            ("config3_launch_b1_h12_t512", (1, 512, 12), 1000, 2000),
                                            ("single_long_prompt_b1_h12_t4096", (1, 4096, 12), 100, 300)):
            q, k, v = self.sliced_inputs(B, T, H)
            oo = torch.empty((B, T, 64 * H), device=self.dev)
            a = self.fwd_args(q, k, v, oo, B, T, H)
            us = self.events_us(lambda: self.launch(a), reps=reps, warm=warm)
            tf = self.cmhsa.algorithmic_flops(B, T, H, 64) / us / 1e6
            others[name] = {"us_per_launch": round(us, 2), "tflops": round(tf, 2), "frac_of_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
                            "warmup": warm, "reps": reps, "plan": self.cmhsa.plan_description(B, T, H, self.workspace.nbytes).split(", workspace")[0]}
            del q, k, v, oo
        torch.cuda.empty_cache()
        # SURVEY 8(d) config 2 asks for more than the headline: contiguous as well as QKV-sliced inputs, randn as well
        # as rand, and the latency of ONE launch (launch -> completion seen by the host) next to the back-to-back rate
        c2 = WORKLOADS["config2"]
        B, T, H = c2["B"], c2["T"], c2["H"]
        d = 64 * H
        out = torch.empty((B, T, d), device=self.dev)
        variants = {}
        for vname, maker, sliced in (("qkv_sliced_rand", torch.rand, True), ("contiguous_rand", torch.rand, False),
                                     ("qkv_sliced_randn", torch.randn, True), ("contiguous_randn", torch.randn, False)):
            if sliced:
                vq, vk, vv = self.sliced_inputs(B, T, H, maker)
            else:
                vq, vk, vv = (maker((B, T, d), device=self.dev) for _ in range(3))
            va = self.fwd_args(vq, vk, vv, out, B, T, H)
            variants[vname + "_us"] = round(self.events_us(lambda: self.launch(va), reps=200, warm=20), 3)
        # the entry point without a workspace never splits a query block across workgroups
        vq, vk, vv = self.sliced_inputs(B, T, H)
        na = self.fwd_args(vq, vk, vv, out, B, T, H)[:16] + (ctypes.c_void_p(self.stream.cuda_stream),)
        variants["no_workspace_entry_point_cmhsa_fp32_fwd_us"] = round(self.events_us(lambda: lib.cmhsa_fp32_fwd(*na), reps=200, warm=20), 3)
        lat = []
        va = self.fwd_args(vq, vk, vv, out, B, T, H)
        for _ in range(50):
            torch.cuda.synchronize(self.dev)
            t1 = time.perf_counter()
            self.launch(va)
            torch.cuda.synchronize(self.dev)
            lat.append((time.perf_counter() - t1) * 1e6)
        variants["single_launch_host_latency_us_median"] = round(statistics.median(lat), 2)
        others["config2_variants"] = variants
        sp = ctypes.c_void_p(self.stream.cuda_stream)
        # the fused tanh-GELU kernel: HBM-bound, 8 algorithmic bytes per element
        for name, n in (("gelu_gpt2xl_b64_t512", 64 * 512 * 6400), ("gelu_gpt2_t256", 256 * 3072)):
            gx = torch.randn(n, device=self.dev)
            gy = torch.empty_like(gx)
            ga = (gx.data_ptr(), gy.data_ptr(), n, sp)
            us = self.events_us(lambda: lib.gpt2_gelu_tanh_f32_fwd(*ga), reps=20, warm=20)
            gbs = lib.gpt2_gelu_algorithmic_bytes(n) / us / 1e3
            others[name] = {"workload": f"fused tanh-GELU fp32, {n} elements", "us_per_launch": round(us, 2),
                            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                         "frac": round(gbs / HBM_PEAK_GBPS, 4), "traffic": None}}
            del gx, gy
        # fused residual add + LayerNorm (include/gpt2_layernorm.h): HBM-bound, 16 algorithmic bytes per element
        for name, (rows, cols) in (("add_layernorm_gpt2xl_b64_t512", (64 * 512, 1600)), ("add_layernorm_gpt2_t256", (256, 768))):
            lx, ld = torch.randn((rows, cols), device=self.dev), torch.randn((rows, cols), device=self.dev)
            lw, lb = torch.randn(cols, device=self.dev), torch.randn(cols, device=self.dev)
            ls, ly = torch.empty_like(lx), torch.empty_like(lx)
            la = (lx.data_ptr(), ld.data_ptr(), lw.data_ptr(), lb.data_ptr(), ls.data_ptr(), ly.data_ptr(), rows, cols, cols, cols, 1e-5, sp)
            us = self.events_us(lambda: lib.gpt2_add_layernorm_f32_fwd(*la), reps=20, warm=20)
            gbs = lib.gpt2_add_layernorm_algorithmic_bytes(rows, cols, 1) / us / 1e3
            others[name] = {"workload": f"fused residual add + LayerNorm fp32, {rows} rows of {cols}", "us_per_launch": round(us, 2),
                            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                         "frac": round(gbs / HBM_PEAK_GBPS, 4), "traffic": None}}
            del lx, ld, ls, ly
        # the KV-cache decode step (opt-in): HBM/L2-bound, reads every cached K and V row once
        for name, (db, dh, keys) in (("decode_b64_h25_k512", (64, 25, 512)), ("decode_b1_h12_k512", (1, 12, 512))):
            dd = 64 * dh
            kc = torch.randn((db, keys, dd), device=self.dev)
            vc = torch.randn((db, keys, dd), device=self.dev)
            new = torch.randn((db, 3 * dd), device=self.dev)
            do = torch.empty((db, dd), device=self.dev)
            pos = torch.tensor([keys - 1], dtype=torch.int32, device=self.dev)
            da = (new.data_ptr(), new.stride(0), kc.data_ptr(), vc.data_ptr(), kc.stride(1), kc.stride(0), do.data_ptr(), do.stride(0),
                  pos.data_ptr(), db, keys, dh, 64, sp)
            us = self.events_us(lambda: lib.cmhsa_fp32_decode_step(*da), reps=50, warm=20)
            gbs = lib.cmhsa_decode_algorithmic_bytes(db, keys, dh, 64) / us / 1e3
            others[name] = {"workload": f"KV-cache decode step fp32, B={db} H={dh}, {keys} keys", "us_per_launch": round(us, 2),
                            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                         "frac": round(gbs / HBM_PEAK_GBPS, 4), "traffic": None}}
            del kc, vc
        # the few-row linear layer of the KV-cache step (include/gpt2_linear.h): HBM-bound, the weight matrix read once.
        # Twelve weight sets in rotation, as a model's layers: nothing is served out of a cache the step would not have.
        for name, (n_out, kk, sets) in (("linear_rows_gpt2_qkv_1row", (2304, 768, 12)), ("linear_rows_gpt2_lm_head_1row", (50257, 768, 3))):
            ws = [torch.randn((n_out, kk), device=self.dev) * 0.02 for _ in range(sets)]
            lb = torch.randn(n_out, device=self.dev)
            lx = torch.randn((1, kk), device=self.dev)
            ly = torch.empty((1, n_out), device=self.dev)
            turn = [0]

            def one_linear():
                turn[0] = (turn[0] + 1) % sets
                return lib.gpt2_linear_rows_f32_fwd(lx.data_ptr(), kk, ws[turn[0]].data_ptr(), kk, lb.data_ptr(), ly.data_ptr(), n_out, 1, n_out, kk, 0, sp)

            us = self.events_us(one_linear, reps=60, warm=24)
            gbs = lib.gpt2_linear_rows_algorithmic_bytes(1, n_out, kk, 1) / us / 1e3
            others[name] = {"workload": f"few-row linear fp32, 1 row, weight {n_out} x {kk} (+ bias)", "us_per_launch": round(us, 2),
                            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                         "frac": round(gbs / HBM_PEAK_GBPS, 4), "traffic": None}}
            del ws
        return others

    # -- generation (the metric's first clause: GPT2 512-token seconds; BASELINE configs 1 and 3) -----------------
    def generation(self, budget_s: float) -> dict:
        """GPT2 greedy generation to 512 total tokens from the reference prompt, B=1, seeded random weights (no
        checkpoint offline): the reference loop with the naive formula and with this kernel on the same GPU, the
        HIP-graph loop with the fused GELU, tokens compared; the larger models and config 1 (naive PyTorch on the
        host CPU) as far as the time budget reaches."""
        torch = self.torch
        import gpt2_generation as gen
        from causal_self_attention import ScaledDotProductAttentionBackend as Backend

        t_start = time.perf_counter()
        left = lambda: budget_s - (time.perf_counter() - t_start)
        prompt = list(gen.REFERENCE_PROMPT_IDS)
        res = {"prompt": "reference prompt ids (8 tokens)", "max_tokens": 512, "batch": 1, "weights": "seeded random (no checkpoint offline)",
               "note": "seconds = the second, timed run as reference gpt2.py:104-134 (warm-up run first)"}

        def run(model_name, backend, graphs, fused, fused_all=False):
            # the small model's weights are drawn on the CPU so that config 1 (CPU) and config 3 (GPU) run the SAME model
            model = gen.build_model(model_name, backend, self.dev, seed=0, fused_gelu=fused, fused_add_ln=fused_all,
                                    fused_step_ops=fused_all, init_on_device=model_name != "gpt2")
            g = gen.BucketedGreedyGenerator(model, 1, 512, bucket=64, device=self.dev) if graphs else None
            t = gen.time_greedy_generation(model, prompt, 512, self.dev, g)
            n_new = len(t.tokens[0]) - len(prompt)
            del g, model
            torch.cuda.empty_cache()
            return t.tokens[0], {"seconds": round(t.second_run_seconds, 4), "tokens_generated": n_new,
                                 "tokens_per_second": round(n_new / t.second_run_seconds, 1)}

        small = {}
        tok_naive, small["reference_loop_naive_attention"] = run("gpt2", Backend.NAIVE_PYTORCH, False, False)
        tok_hip, small["reference_loop_hip_kernel"] = run("gpt2", Backend.CUSTOM_CUDA, False, False)
        tok_graph, small["hip_graphs_fused_gelu_hip_kernel"] = run("gpt2", Backend.CUSTOM_CUDA, True, True)
        # + every residual add fused into the LayerNorm behind it (include/gpt2_layernorm.h) and the loop's bookkeeping --
        # embedding sum, next-token step -- as one launch each (include/gpt2_step.h): 33 launches fewer per token
        tok_ln, small["hip_graphs_all_fusions_hip_kernel"] = run("gpt2", Backend.CUSTOM_CUDA, True, True, True)
        small["tokens_identical_naive_vs_hip_vs_graphs"] = bool(tok_naive == tok_hip == tok_graph == tok_ln)
        # attention share: the 504 x 12 attention launches of that generation (T = 8..511, H = 12) replayed on their own
        H, d = 12, 768
        qkv = torch.rand((1, 512, 3 * d), device=self.dev)
        out = torch.empty((1, 512, d), device=self.dev)
        launches = []
        for T in range(8, 512):
            q, k, v = qkv[:, :T, :d], qkv[:, :T, d:2 * d], qkv[:, :T, 2 * d:]
            launches.append(self.fwd_args(q, k, v, out[:, :T], 1, T, H))
        def replay():
            for a in launches:
                for _ in range(12):
                    self.launch(a)
        replay()
        torch.cuda.synchronize(self.dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(self.stream)
        replay()
        e1.record(self.stream)
        torch.cuda.synchronize(self.dev)
        att_s = e0.elapsed_time(e1) * 1e-3
        small["attention_kernel_seconds_6048_launches"] = round(att_s, 4)
        # the same 6 048 launches captured once into a HIP graph and replayed: device time without the host's launch cost
        # (the eager figure above is host-bound wherever a launch is shorter than the ~3-4 us a ctypes call takes)
        cap = torch.cuda.Stream(self.dev)
        cap.wait_stream(self.stream)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=cap):
            cs = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            for a in launches:
                for _ in range(12):
                    self.launch(a[:18] + (cs,))
        graph.replay()
        torch.cuda.synchronize(self.dev)
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record(self.stream)
        graph.replay()
        g1.record(self.stream)
        torch.cuda.synchronize(self.dev)
        small["attention_kernel_seconds_6048_launches_one_hip_graph"] = round(g0.elapsed_time(g1) * 1e-3, 4)
        small["attention_tflops_over_the_6048_launches_hip_graph"] = round(
            sum(12 * self.cmhsa.algorithmic_flops(1, T, H, 64) for T in range(8, 512)) / (g0.elapsed_time(g1) * 1e-3) / 1e12, 2)
        del graph
        small["attention_share_of_reference_loop_hip"] = round(att_s / small["reference_loop_hip_kernel"]["seconds"], 4)
        small["attention_share_of_hip_graph_loop"] = round(att_s / small["hip_graphs_fused_gelu_hip_kernel"]["seconds"], 4)
        # opt-in KV-cache mode (NOT the reference's algorithm, which keeps no cache): one position per token
        if left() >= 15.0:
            model = gen.build_model("gpt2", Backend.CUSTOM_CUDA, self.dev, seed=0, fused_gelu=True, fused_add_ln=True, fused_step_ops=True)
            kv = gen.KVCacheGreedyGenerator(model, 1, 512, device=self.dev)
            t = gen.time_greedy_generation(model, prompt, 512, self.dev, kv)
            n_new = len(t.tokens[0]) - len(prompt)
            small["kv_cache_opt_in_all_fusions"] = {"seconds": round(t.second_run_seconds, 4), "tokens_generated": n_new,
                                                    "tokens_per_second": round(n_new / t.second_run_seconds, 1),
                                                    "tokens_identical_to_reference_loop": bool(t.tokens[0] == tok_naive)}
            del kv
            if left() >= 8.0:                       # + the step's linear layers through the few-row HIP kernel (include/gpt2_linear.h)
                kv = gen.KVCacheGreedyGenerator(model, 1, 512, device=self.dev, few_row_linear=True)
                t = gen.time_greedy_generation(model, prompt, 512, self.dev, kv)
                n_new = len(t.tokens[0]) - len(prompt)
                small["kv_cache_opt_in_all_fusions_few_row_linear"] = {
                    "seconds": round(t.second_run_seconds, 4), "tokens_generated": n_new, "tokens_per_second": round(n_new / t.second_run_seconds, 1),
                    "tokens_identical_to_reference_loop": bool(t.tokens[0] == tok_naive)}
                del kv
            del model
            torch.cuda.empty_cache()
        res["gpt2"] = small
        del qkv, out, launches
        for name, est in (("gpt2-medium", 8.0), ("gpt2-large", 14.0), ("gpt2-xl", 22.0)):
            if left() < est:
                res[name] = "skipped: generation time budget (--gen-budget) spent"
                continue
            _, r = run(name, Backend.CUSTOM_CUDA, True, True, True)
            res[name] = {"hip_graphs_all_fusions_hip_kernel": r}
        # config 1: naive PyTorch attention on the host CPU (reference loop), all of this host's cores
        if left() >= 45.0:
            cores = host_cores()
            torch.set_num_threads(cores)
            model = gen.build_model("gpt2", Backend.NAIVE_PYTORCH, "cpu", seed=0)
            t = gen.time_greedy_generation(model, prompt, 512, "cpu")
            n_new = len(t.tokens[0]) - len(prompt)
            res["config1_cpu_naive"] = {"seconds": round(t.second_run_seconds, 3), "tokens_generated": n_new,
                                        "tokens_per_second": round(n_new / t.second_run_seconds, 2), "cores": cores,
                                        "tokens_identical_to_gpu_naive": bool(t.tokens[0] == tok_naive)}
        else:
            res["config1_cpu_naive"] = "skipped: generation time budget (--gen-budget) spent"
        res["seconds_spent"] = round(time.perf_counter() - t_start, 1)
        return res

    # -- N > 1: config 5 through sharded_attention, config 4 data-parallel -----------------------------------------
    def config5_breakdown(self, B: int, T: int, H_total: int) -> dict:
        """BASELINE configs[4] through llm-inference_amd/sharded_attention.py with the HIP kernel as `compute`:
        compute only (this rank's heads), the all-gather on its own, and end to end including the
        (W, B, T, slab) -> (B, T, H*64) permute."""
        torch, dist, world = self.torch, self.dist, self.world
        import sharded_attention as sa

        d = 64 * H_total
        torch.manual_seed(77)                                   # same inputs on every rank, as head sharding assumes
        Q, K, V = self.sliced_inputs(B, T, H_total)
        compute = lambda q, k, v, h: self.cmhsa.attention_fwd(q, k, v, h, workspace=self.workspace)
        per = H_total // world

        def span(fn, reps=5, warm=2):
            for _ in range(warm):
                fn()
            self.fence()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(self.stream)
            for _ in range(reps):
                fn()
            b.record(self.stream)
            self.fence()
            return a.elapsed_time(b) / reps

        slab = sa.head_sharded_attention(Q, K, V, H_total, compute, gather=False).contiguous()
        flat = torch.empty((world * B, T, 64 * per), device=self.dev)
        comp_ms = span(lambda: sa.head_sharded_attention(Q, K, V, H_total, compute, gather=False))
        ag_ms = span((lambda: dist.all_gather_into_tensor(flat, slab)) if self.backend == "nccl" else (lambda: sa.all_gather_dim0(slab, world)))
        e2e_ms = span(lambda: sa.head_sharded_attention(Q, K, V, H_total, compute))
        # pipelined over batch pieces: gather + permute of piece i on a second stream beside the kernel of piece i+1
        piped_ms = {c: span(lambda c=c: sa.head_sharded_attention(Q, K, V, H_total, compute, chunks=c)) for c in (2, 4) if c <= B}
        comp_ms, ag_ms, e2e_ms, *piped = self.max_over_ranks(comp_ms, ag_ms, e2e_ms, *piped_ms.values())
        piped_ms = dict(zip(piped_ms, piped))
        full = sa.head_sharded_attention(Q, K, V, H_total, compute)
        piped_diff = max((float((sa.head_sharded_attention(Q, K, V, H_total, compute, chunks=c) - full).abs().max()) for c in piped_ms), default=0.0)
        # parity of the gathered result: one head this rank did NOT compute, against the f64 oracle
        from oracle import naive_attention as oracle
        h = ((self.rank + 1) % world) * per
        sl = slice(64 * h, 64 * h + 64)
        want = oracle.naive_attention_f64(Q[0, :, sl].cpu().numpy(), K[0, :, sl].cpu().numpy(), V[0, :, sl].cpu().numpy(), 1)
        err = float(abs(full[0, :, sl].cpu().numpy() - want).max())
        err, piped_diff = self.max_over_ranks(err, piped_diff)
        recv = (world - 1) * slab.numel() * 4
        total_flops = self.cmhsa.algorithmic_flops(B, T, H_total, 64)
        return {"via": "sharded_attention.head_sharded_attention(compute=cmhsa.attention_fwd)",
                "compute_only_ms": round(comp_ms, 4),
                "compute_only_tflops_per_gpu": round(total_flops / world / (comp_ms * 1e-3) / 1e12, 2),
                "all_gather_ms": round(ag_ms, 4), "all_gather_received_bytes_per_rank": recv,
                "all_gather_GBps_received_per_rank": round(recv / (ag_ms * 1e-3) / 1e9, 1),
                "end_to_end_ms_incl_permute": round(e2e_ms, 4),
                "end_to_end_tflops_all_gpus": round(total_flops / (e2e_ms * 1e-3) / 1e12, 2),
                "end_to_end_ms_pipelined_over_batch_pieces": {str(c): round(ms, 4) for c, ms in piped_ms.items()},
                "pipelined_max_abs_diff_vs_one_gather": piped_diff,
                "gathered_max_abs_err_vs_f64_oracle": err,
                "note": "GB/s is per rank over all its links together; the per-link rate depends on how RCCL routes the collective"}

    def config5_leg(self) -> dict | str:
        wl = WORKLOADS["config5"]
        if wl["H"] % self.world != 0:
            return f"skipped: {wl['H']} heads do not divide over {self.world} GPUs"
        B, T = wl["B"], wl["T"]
        if self.backend != "nccl":
            B, T = 2, 512
        return self.config5_breakdown(B, T, wl["H"])

    def config4_dp_leg(self, budget_s: float) -> dict | str:
        """BASELINE configs[3]: GPT2-XL, 64 copies of the reference prompt split evenly over the ranks (data-parallel,
        replicated weights, no collective on the data path), every prompt to 512 total tokens.  Two forms, each labelled:
          kv_cache_512_tokens   the config at its stated length through the opt-in KV-cache loop (one HIP graph per token
                                step; NOT the reference's algorithm, which keeps no cache; ~4 s on one GPU);
          graph_loop            the reference's algorithm (full re-forward per token) as HIP graphs + fused kernels: at its
                                stated 512 tokens it takes ~210 s per run on one GPU (profiles/generation/
                                config4_xl_b64_t512_graphs_fusedgelu.json), so it runs at full length only when
                                --gen-budget allows (>= 500 s left) and otherwise as a 72-token sample, labelled as such."""
        torch = self.torch
        import gpt2_generation as gen
        from causal_self_attention import ScaledDotProductAttentionBackend as Backend

        if budget_s < 25:
            return "skipped: generation time budget (--gen-budget) spent"
        t_start = time.perf_counter()
        left = lambda: budget_s - (time.perf_counter() - t_start)
        n_prompts = 64
        model_name = "gpt2-xl"
        overrides = None
        if self.backend != "nccl":
            n_prompts, overrides = 8, dict(n_layers=2)
        mine = gen.shard_prompts(n_prompts, self.rank, self.world)
        prompts = [list(gen.REFERENCE_PROMPT_IDS) for _ in mine]
        model = gen.build_model(model_name, Backend.CUSTOM_CUDA, self.dev, seed=0, fused_gelu=True, fused_add_ln=True, fused_step_ops=True,
                                init_on_device=True, overrides=overrides)

        def timed(g, max_tokens, label):
            g.generate(prompts)                                  # captures the graphs (warm-up)
            self.fence()
            t0 = time.perf_counter()
            toks = g.generate(prompts)
            torch.cuda.synchronize(self.dev)
            seconds = time.perf_counter() - t0
            generated = float(sum(len(t) - len(gen.REFERENCE_PROMPT_IDS) for t in toks))
            (seconds,) = self.max_over_ranks(seconds)
            if self.dist is not None:
                t = torch.tensor([generated], device=self.dev, dtype=torch.float64)
                self.dist.all_reduce(t)
                generated = float(t[0])
            return toks, {"loop": label, "max_tokens": max_tokens, "full_length_of_the_config": max_tokens == 512,
                          "seconds": round(seconds, 3), "tokens_generated_all_gpus": int(generated),
                          "tokens_per_second_all_gpus": round(generated / seconds, 1),
                          "rows_identical_on_rank0": bool(all(row == toks[0] for row in toks))}

        res = {"workload": f"{model_name} fp32, {n_prompts} prompts over {self.world} GPU(s) ({len(prompts)} on rank 0), 512 total tokens each"}
        kv = gen.KVCacheGreedyGenerator(model, len(prompts), 512, device=self.dev)
        toks_kv, res["kv_cache_512_tokens"] = timed(kv, 512, "opt-in KV cache, one HIP graph per token step, fused GELU / add + LayerNorm / step kernels")
        del kv
        torch.cuda.empty_cache()
        # every rank must take the same branch: decide on the slowest rank's clock
        (room,) = self.max_over_ranks(-left())
        full = -room >= 500.0
        max_tokens = 512 if full else 72
        g = gen.BucketedGreedyGenerator(model, len(prompts), max_tokens, bucket=64, device=self.dev)
        toks_g, res["graph_loop"] = timed(g, max_tokens, "reference algorithm (no KV cache) as HIP graphs per 64-token bucket + fused kernels"
                                          + ("" if full else "; BOUNDED SAMPLE: 72 of the config's 512 tokens"))
        res["graph_loop"]["tokens_identical_to_kv_cache_loop_on_the_common_prefix"] = bool(
            all(a[:max_tokens] == b2[:max_tokens] for a, b2 in zip(toks_g, toks_kv)))
        del g, model
        torch.cuda.empty_cache()
        return res


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if os.environ.get("CMHSA_BENCH_WATCHDOG"):               # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["CMHSA_BENCH_WATCHDOG"]), exit=True)
    b = Bench(args)
    result = b.headline()
    if b.rank == 0 and not args.no_cpu_baseline:
        # (N > 1: a shorter sample -- the other ranks wait for rank 0 at the next barrier)
        result["cpu_baseline"] = cpu_baseline(WORKLOADS[args.workload], b.torch, budget_s=20.0 if b.world == 1 else 6.0)
    if not args.no_extras:
        # The legs below run collectives and minutes of generation that the headline does not depend on.  If they are
        # not done after --extras-limit seconds (a rank died inside a leg, a collective stalled) every rank leaves on
        # its own and rank 0 prints the line as it stood after the headline.
        import threading

        headline_only = json.dumps({**result, "extras": f"abandoned: not finished after {args.extras_limit:.0f} s; "
                                                          "the figures of this line were measured before them"})
        extras_done = threading.Event()

        def leave_without_extras() -> None:
            if extras_done.wait(args.extras_limit):
                return
            if b.rank == 0:
                print(headline_only, flush=True)
            os._exit(0)

        threading.Thread(target=leave_without_extras, daemon=True).start()

        def leg(fn, *a):
            try:
                return fn(*a)
            except Exception as e:                               # noqa: BLE001 -- recorded in the line, the headline stands
                return f"failed: {type(e).__name__}: {e}"[:300]

        t0 = time.perf_counter()
        if os.environ.get("CMHSA_BENCH_STALL_EXTRAS"):           # tests only: stands in for a collective that never returns
            time.sleep(1e6)
        if b.world == 1 and args.workload == "config2":
            result["other_workloads"] = leg(b.other_workloads)
            gen = leg(b.generation, args.gen_budget)
            spent = gen["seconds_spent"] if isinstance(gen, dict) else args.gen_budget
            if isinstance(result["other_workloads"], dict):
                result["other_workloads"]["generation"] = gen
            else:
                result["generation"] = gen
            # the N = 1 point of the data-parallel config-4 sample (the N > 1 lines carry the same leg)
            result["config4_dp"] = leg(b.config4_dp_leg, args.gen_budget - spent)
        if b.world > 1 and args.workload == "config2":
            result["config5"] = leg(b.config5_leg)
            result["config4_dp"] = leg(b.config4_dp_leg, args.gen_budget)
        result["extras_seconds"] = round(time.perf_counter() - t0, 1)
        if b.dist is not None:
            b.dist.barrier()                                     # (still under the limit: a rank that failed a leg may be missing)
        extras_done.set()
    if b.dist is not None:
        b.dist.barrier()
        b.dist.destroy_process_group()
    if b.rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

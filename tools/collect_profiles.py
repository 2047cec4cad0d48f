"""Copies the summaries of one tools/prof_all.sh session (gpurun_out/prof_<round>/) into profiles/ under per-round
names and derives the counter summaries.   usage: python tools/collect_profiles.py r02"""
import collections, csv, glob, json, os, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{rnd}")
dst = os.path.join(root, "profiles")


def newest(pattern):
    """gpurun MERGES each session's files into gpurun_out/ (rocprofv3 names them by process id), so files of earlier
    sessions stay behind: only the newest match belongs to the session being collected."""
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


for name in ("bench_config2", "bench_config5_shard", "bench_config5_n1"):
    shutil.copy(os.path.join(src, name + ".json"), os.path.join(dst, f"{rnd}_{name}.json"))
OURS = ("cmhsa_fwd", "fwd16_kernel")          # the attention kernels: cmhsa_fwd.hip's and cmhsa_fwd16.hip's
for tag in ("bench_py", "config2", "config5_shard", "config3_t128", "config3_t256", "config3_t512", "config4_shape"):
    f = newest(f"trace_{tag}/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv" if tag != "bench_py" else f"{rnd}_bench_py_config2_kernel_stats.csv"))
for tag in ("config2", "config5_shard", "config3_t256", "config4_shape"):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for pass_dir in sorted(glob.glob(os.path.join(src, f"pmc_{tag}", "pass*"))):
        if not os.path.isdir(pass_dir):
            continue
        f = newest(os.path.relpath(os.path.join(pass_dir, "**", "*counter_collection.csv"), src))
        if not f:
            continue
        for row in csv.DictReader(open(f)):
            if not any(n in row["Kernel_Name"] for n in OURS):
                continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[row["Counter_Name"]] += 1
    mean = {k: tot[k] / cnt[k] for k in sorted(tot)}
    f = newest(f"trace_{tag}/**/*kernel_stats.csv")
    if f:
        for row in csv.DictReader(open(f)):
            if any(n in row["Name"] for n in OURS):
                mean["kernel_name"] = row["Name"][:120]
                mean["kernel_avg_ns"] = float(row["AverageNs"])
                mean["kernel_calls"] = int(row["Calls"])
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:        # KiB; FETCH_SIZE reads half on gfx950 (MI355X_MICROARCH.md, HBM)
        mean["hbm_bytes_per_launch_corrected"] = (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024
    if "SQ_INSTS_MFMA" in mean:
        # 64 cycles per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32 (the 16-key-tile kernel); 1024 SIMDs
        mean["mfma_busy_cycles_per_simd"] = mean["SQ_INSTS_MFMA"] * (32 if "fwd16_kernel" in mean.get("kernel_name", "") else 64) / 1024
    mean["round"] = rnd
    json.dump(mean, open(os.path.join(dst, f"pmc_{tag}_summary.json"), "w"), indent=1)
    print(tag, json.dumps(mean, indent=1))
print(open(os.path.join(src, "steps.log")).read())

"""Summarise rocprofv3 --pmc CSVs written by tools/prof_pmc.sh: per-dispatch mean of each counter
for the attention kernel.   usage: python tools/pmc_summary.py gpurun_out/pmc_TAG"""
import csv, glob, sys, collections, json

root = sys.argv[1]
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cmhsa" not in row["Kernel_Name"]:
            continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
mean = {k: tot[k] / cnt[k] for k in tot}
for k in sorted(mean):
    print(f"{k:32s} {mean[k]:16.1f}   (n={cnt[k]})")
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cmhsa" in row["Name"]:
            print("kernel avg ns:", row["AverageNs"], "calls", row["Calls"])
            mean["kernel_avg_ns"] = float(row["AverageNs"])
json.dump(mean, open(root + "/summary.json", "w"), indent=1)

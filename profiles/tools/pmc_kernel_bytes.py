#!/usr/bin/env python3
"""Per-kernel fabric-side read bytes from a `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum
TCC_EA0_RDREQ_128B_sum --kernel-trace` run directory: sum over request sizes, mean per launch, with the launch durations of
the same run.   python profiles/tools/pmc_kernel_bytes.py <dir> [kernel-name-substring ...]"""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
want = sys.argv[2:] or ["k_forward"]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        if any(w in k for w in want):
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        if any(w in k for w in want):
            dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9)
out = {}
for k, c in cnt.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    n32, n128 = m.get("TCC_EA0_RDREQ_32B_sum", 0.0), m.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    n64 = m.get("TCC_EA0_RDREQ_64B_sum", m.get("TCC_EA0_RDREQ_sum", 0.0) - n32 - n128)
    b = 32 * n32 + 64 * n64 + 128 * n128
    t = sum(dur[k]) / max(len(dur[k]), 1)
    out[k] = {"launches": len(dur[k]), "mean_s_profiled": t, "fabric_read_bytes_per_launch": b,
              "fabric_read_gbs": b / t / 1e9 if t else None, "frac_of_8000_gbs": b / t / 8e12 if t else None,
              "frac_of_6300_gbs_achievable": b / t / 6.3e12 if t else None, "counters": m}
print(json.dumps(out, indent=1))

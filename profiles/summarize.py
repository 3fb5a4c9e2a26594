#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<tag>_*) into small committed summaries under profiles/.

    python profiles/summarize.py r01a        # reads gpurun_out/r01a_*, writes profiles/r01a_*.{csv,json}
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


# kernel stats (rocprofv3 --kernel-trace --stats)
for f in glob.glob(os.path.join(G, tag + "_stats*", "*", "*_kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    out = os.path.join(P, "%s_kernel_stats.csv" % os.path.basename(os.path.dirname(os.path.dirname(f))))
    with open(out, "w") as fh:
        fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            fh.write('"%s",%s,%s,%s,%s,%s,%s,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                      r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]))
    print("wrote", out)

# PMC passes: per kernel, per counter, split by grid size (large = the per-GPU batch launches)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(G, tag + "_pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        key = "%s|grid=%s" % (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))
        pmc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in sorted(pmc.items()):
    if not k.startswith("k_"):
        continue
    summary[k] = {c: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in cs.items()}
if summary:
    out = os.path.join(P, "%s_pmc_summary.json" % tag)
    json.dump(summary, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)

# HBM traffic of the headline forward launch for bench.py's roofline.traffic: FETCH_SIZE / WRITE_SIZE are in
# KiB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads -> doubled
# (MI355X_MICROARCH.md, HBM section).  Only the full-batch launches (largest grid) are used.
fwd = [(k, v) for k, v in summary.items() if k.startswith("k_forward_straight_u<double>")]
if fwd:
    k, v = max(fwd, key=lambda kv: int(kv[0].split("grid=")[1]))
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        traffic = 2.0 * v["FETCH_SIZE"]["mean"] * 1024 + v["WRITE_SIZE"]["mean"] * 1024
        out = os.path.join(P, "pmc_forward.json")
        json.dump({"rays_per_launch": 260400, "samples_per_ray": 257, "hbm_bytes_per_launch": traffic,
                   "fetch_size_kib": v["FETCH_SIZE"]["mean"], "write_size_kib": v["WRITE_SIZE"]["mean"],
                   "kernel": k, "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                                          "FETCH_SIZE doubled per the gfx950 correction)" % tag}, open(out, "w"), indent=1)
        print("wrote", out)

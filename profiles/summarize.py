#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<tag>_*) into small committed summaries under profiles/.

    python profiles/summarize.py r02a        # reads gpurun_out/r02a_*, writes profiles/r02a_*.{csv,json} and
                                             # profiles/pmc_counters.json (read by bench.py while csrc is unchanged)
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
LEG_KERNEL = {"forward": "k_forward_bundle<", "f32_forward": "k_forward_bundle_f32", "adjoint": "k_adjoint_binned<double, 0, double",
              "cubic_forward": "k_forward_bundle_lm", "cubic_adjoint": "k_adjoint_binned_lm4", "fermat_cubic": "k_fermat_tec_lm",
              "fermat_linear": "k_fermat_tec<0",
              "fermat_cfg3": "k_fermat_tec_lm"}


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def newest(pattern):
    """one file per run directory: the newest (a directory re-used by a later run of the same tag holds the older run's files too)"""
    by_dir = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


# kernel stats (rocprofv3 --kernel-trace --stats)
for f in newest(os.path.join(G, tag + "_stats_*", "*", "*_kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    out = os.path.join(P, "%s_kernel_stats.csv" % os.path.basename(os.path.dirname(os.path.dirname(f))))
    with open(out, "w") as fh:
        fh.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            fh.write('"%s",%s,%s,%s,%s,%s,%s,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                      r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]))
    print("wrote", out)
for f in sorted(glob.glob(os.path.join(G, tag + "_stats_*.json")) + glob.glob(os.path.join(G, tag + "_bench.json"))):
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        json.dump(json.loads(line), open(os.path.join(P, os.path.basename(f)), "w"), indent=1)
    except Exception as exc:
        print("skip", f, exc)

# PMC passes: per leg, per kernel, per counter (mean over the launches of the pass, largest grid only)
legs = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in newest(os.path.join(G, tag + "_pmc_*", "*", "*_counter_collection.csv")):
    leg = os.path.basename(os.path.dirname(os.path.dirname(f)))[len(tag) + 5:].rsplit("_", 1)[0]
    for r in csv.DictReader(open(f)):
        key = "%s|grid=%s" % (short(r["Kernel_Name"]), r.get("Grid_Size", "?"))
        legs[leg][key][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for leg, ks in legs.items():
    summary[leg] = {k: {c: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in cs.items()}
                    for k, cs in sorted(ks.items()) if k.startswith("k_")}
if summary:
    out = os.path.join(P, "%s_pmc_summary.json" % tag)
    json.dump(summary, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)
    import bench
    pc = {"csrc_sha": bench.csrc_sha(),
          "source": "profiles/%s_pmc_summary.json: rocprofv3 --pmc, one counter set per run of `bench.py --only <leg>`, mean over the "
                    "launches of the leg's kernel (largest grid)" % tag}
    for leg, kname in LEG_KERNEL.items():
        cand = [(k, v) for k, v in summary.get(leg, {}).items() if k.startswith(kname)]
        if not cand:
            continue
        k, v = max(cand, key=lambda kv: int(kv[0].split("grid=")[1]))
        pc[leg] = dict({c: x["mean"] for c, x in v.items()}, rays=bench.NA * bench.ND * bench.NT, Ns=bench.NS, kernel=k)
        if leg.startswith("fermat"):
            pc[leg]["rays"] = 2604 if leg == "fermat_cfg3" else 620000
    json.dump(pc, open(os.path.join(P, "pmc_counters.json"), "w"), indent=1, sort_keys=True)
    print("wrote profiles/pmc_counters.json for csrc", pc["csrc_sha"])

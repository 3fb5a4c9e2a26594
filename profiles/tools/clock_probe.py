#!/usr/bin/env python3
"""Clock the chip held during a kernel: GRBM_GUI_ACTIVE / 8 XCDs over the kernel's own duration (Start/End timestamps of the
same rocprofv3 --pmc --kernel-trace csv), + VALU issue utilisation.   python profiles/tools/clock_probe.py <substring> <dir> ..."""
import collections, csv, glob, json, os, sys
want, dirs = sys.argv[1], sys.argv[2:]
res = {}
for d in dirs:
    cnt, dur = collections.defaultdict(list), []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if not dur:
        continue
    m = {k: sum(v) / len(v) for k, v in cnt.items()}
    cyc, ns = m["GRBM_GUI_ACTIVE"] / 8.0, sum(dur) / len(dur)
    out = {"launches": len(dur), "kernel_us": ns / 1e3, "cycles_per_xcd": cyc, "clock_ghz": cyc / ns}
    if "SQ_ACTIVE_INST_VALU" in m:
        out["valu_busy"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"):
        if k in m:
            out[k] = m[k]
    res[os.path.basename(d.rstrip("/"))] = out
print(json.dumps(res, indent=1))

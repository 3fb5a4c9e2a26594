#!/usr/bin/env python3
"""Mean counter values per launch of the kernels whose name contains a given substring, over every `rocprofv3 --pmc ... --kernel-trace
--output-format csv` run directory given.   python profiles/tools/pmc_kernel_mean.py <substring> <dir> [<dir> ...]"""
import collections, csv, glob, json, os, sys
want, dirs = sys.argv[1], sys.argv[2:]
out = collections.defaultdict(dict)
for d in dirs:
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            if want in k:
                cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in cnt.items():
        for n, v in c.items():
            out[k][n] = sum(v) / len(v)
            out[k]["launches_" + n] = len(v)
print(json.dumps(out, indent=1))

"""Timing-only ablations of the headline forward kernel (what does a wave-load cost, and which part of the path charges it).
Build the variants first (here, hipcc cross-compiles):   bash profiles/tools/ablate_forward.sh build
then on the GPU box:                                      python profiles/tools/ablate_forward.py
Every variant runs in its own child process (its own library: IONOTOMO_LIB); variants 1-5 give WRONG results by design."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch, bench
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
R = w["origins"].shape[0]
e = RayEngine(0)
e.set_grid(w["xvec"], w["yvec"], w["zvec"])
e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
tec = torch.empty(R, dtype=torch.float64, device="cuda")
ts = []
for rnd in range(5):
    for _ in range(3):
        e.forward(o, d, bench.TMAX, bench.NS, out=tec)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        e.forward(o, d, bench.TMAX, bench.NS, out=tec)
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 20)
print(json.dumps({"ms_median": float(np.median(ts)), "ms_min": float(min(ts))}))
""" % ROOT

NAMES = {0: "shipped kernel", 1: "4 loads of 8 B per lane", 2: "2 loads of 16 B", 3: "4 loads, same 4 line runs for every wave (no fills)",
         4: "no loads (VALU + LDS weights only)", 5: "4 loads from the line run of ONE column (1/4 of the fills)",
         6: "shipped loads, slab loop unrolled x 4 at 3 workgroups per CU (16 loads in flight per wave)",
         7: "shipped loads, slab loop unrolled x 2 at 4 workgroups per CU (8 loads in flight per wave)",
         8: "exact: non-temporal loads (nt)", 9: "exact: agent-scope loads (sc1, bypassing the L1)"}
out = {}
runs = [(v, 0) for v in (0, 1, 2, 3, 4, 5, 6, 7)] + [(0, b) for b in (2, 3, 4, 5)] + [(3, b) for b in (2, 4)] + [(5, b) for b in (2, 4)]
if len(sys.argv) > 1:
    runs = [(int(a.split(":")[0]), int(a.split(":")[1]) if ":" in a else 0) for a in sys.argv[1:]]
for v, bpc in runs:
    env = dict(os.environ)
    if v:
        env["IONOTOMO_LIB"] = os.path.join(ROOT, "build_ab", "libionotomo_fwd_abl%d.so" % v)
    if bpc:
        env["IONOTOMO_BLOCKS_PER_CU"] = str(bpc)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    key = "abl%d%s" % (v, "_wg%d" % bpc if bpc else "")
    try:
        out[key] = dict(json.loads(r.stdout.strip().splitlines()[-1]), what=NAMES.get(v, "variant %d" % v), workgroups_per_cu=bpc or "resident (6)")
    except Exception:
        out[key] = {"error": r.stderr[-400:]}
    print(key, out[key], flush=True)
dst = os.path.join(ROOT, "gpurun_out", "ablate_forward.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)

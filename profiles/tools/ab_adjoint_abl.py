#!/usr/bin/env python3
"""Timing-only ablation builds of k_adjoint_binned at the bench shape (results are NOT checked: the ablations change them).
    python profiles/tools/ab_adjoint_abl.py lib ...      ("" = the shipped library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
import bench
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
R = w["origins"].shape[0]
e = RayEngine(0)
e.set_grid(w["xvec"], w["yvec"], w["zvec"])
e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
torch.manual_seed(1)
y = torch.randn(R, dtype=torch.float64, device="cuda")
e.plan_adjoint(o, d, bench.TMAX, bench.NS)
g = torch.zeros(e.shape, dtype=torch.float64, device="cuda")
def run():
    e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g)
bench.SETTLE_MS = 100.0
ks = sorted(bench.time_steps(run, 50, 5, torch, None, 1)[1] for _ in range(5))
print(json.dumps({"lib": os.environ.get("IONOTOMO_LIB", "default"), "adjoint_ms_median_of_5": ks[2] * 1e3, "min": ks[0] * 1e3, "sum": float(g.sum())}))
''' % ROOT
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib:
        env["IONOTOMO_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(line[-1] if line else json.dumps({"lib": lib, "error": r.stderr[-400:]}), flush=True)

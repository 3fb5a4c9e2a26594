#!/usr/bin/env python3
"""A/B of builds of the float32 fast-mode kernel (k_forward_bundle_f32; -DF_CAPCOLS / -DF_WPE variants under build_ab/): the bench
shape, forward ms + plan fit fraction + max rel error vs the float64 kernel.  python profiles/tools/ab_f32.py [lib.so ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
import bench
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
out = {"lib": os.environ.get("IONOTOMO_LIB", "default")}
ref = None
for storage in ("f64", "f32"):
    e = RayEngine(0, storage=storage)
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
    o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
    t = torch.empty(o.shape[0], dtype=torch.float64, device=e.device)
    info = e.plan_forward(o, d, bench.TMAX, bench.NS)
    fn = lambda: e.forward(o, d, bench.TMAX, bench.NS, out=t)
    bench.SETTLE_MS = 100.0
    _, k = bench.time_steps(fn, 40, 5, torch, None, 1)
    out[storage + "_ms"] = k * 1e3
    out[storage + "_fit"] = info[2]
    if ref is None:
        ref = t.clone()
    else:
        out["f32_max_rel_err"] = float(((t - ref).abs() / ref.abs()).max())
print(json.dumps(out))
''' % ROOT

res = []
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib:
        env["IONOTOMO_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    res.append(json.loads(line[-1]) if line else {"lib": lib, "error": r.stderr[-400:]})
    print(json.dumps(res[-1]), file=sys.stderr, flush=True)
print(json.dumps(res))

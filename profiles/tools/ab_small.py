#!/usr/bin/env python3
"""A/B builds (IONOTOMO_LIB) on pipeline-sized forward launches: 62 x 42 x Nt rays, lanes = samples kernel.  python profiles/tools/ab_small.py lib ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
import bench
from ionotomo_amd import synthetic as syn
w = bench.build_workload(0)
out = {"lib": os.environ.get("IONOTOMO_LIB", "default")}
e = bench.engine_with_env({"IONOTOMO_HYBRID_MIN": 65}, 0)
e.set_grid(w["xvec"], w["yvec"], w["zvec"])
e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
for nt in (1, 4, 8, 100):
    o, d = syn.ray_bundle(syn.lofar_enu_km(), syn.rotate_about_pole(syn.facet_directions(bench.ND, 4.0, 1), nt))
    ot, dt = e.tensor(o.reshape(-1, 3)), e.tensor(d.reshape(-1, 3))
    t = torch.empty(ot.shape[0], dtype=torch.float64, device=e.device)
    fn = e.forward_launcher(ot, dt, bench.TMAX, bench.NS, t)
    bench.SETTLE_MS = 20.0
    ks = sorted(bench.time_steps(fn, 100, 5, torch, None, 1)[1] for _ in range(5))
    out["Nt_%%d_us" %% nt] = ks[2] * 1e6
    out["Nt_%%d_sum" %% nt] = float(t.sum())
print(json.dumps(out))
''' % ROOT
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib:
        env["IONOTOMO_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    print(line[-1] if line else json.dumps({"lib": lib, "error": r.stderr[-300:]}), flush=True)

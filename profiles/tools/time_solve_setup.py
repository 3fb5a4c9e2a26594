"""Where the wall time of ONE inversion goes at the bench shape: ShardedRays construction (orders, plan), the solver's own
set-up (active set, normalisations) and the iterations -- first call in a process and a repeat."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd import parallel, solvers
from ionotomo_amd.engine import RayEngine

w = bench.build_workload(0)
Na = bench.NA
o, d = w["origins"].reshape(Na, -1, 3), w["directions"].reshape(Na, -1, 3)
eng = RayEngine(0)
eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
x_true = np.exp(w["m"]) * w["K_ne"] / 1e13
eng.set_values(eng.tensor(x_true))
t = eng.forward(eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3)), bench.TMAX, bench.NS).cpu().numpy().reshape(Na, -1)
dobs = t - t[0]
x0 = eng.tensor(x_true * 0.9)
out = {}


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


for rep in ("first", "again"):
    prob, ms = wall(lambda: parallel.ShardedRays(eng, o, d, bench.TMAX, bench.NS, dobs=dobs, cdct=np.full_like(dobs, 1e-4), i0=0))
    out["sharded_rays_%s_ms" % rep] = ms
    for name, solve in (("cgls", solvers.cgls), ("sirt", solvers.sirt)):
        _, t1 = wall(lambda: solve(prob, x0, n_iter=1))
        _, t51 = wall(lambda: solve(prob, x0, n_iter=51))
        out["%s_setup_plus_1_iteration_%s_ms" % (name, rep)] = t1
        out["%s_per_iteration_%s_ms" % (name, rep)] = (t51 - t1) / 50
print(json.dumps(out, indent=1))

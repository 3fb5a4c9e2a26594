"""Steady-state time per iteration of solvers.steepest_descent_log_model (the reference's own algorithm: objective, update,
line search, stopping rule) at the bench shape: difference of a 25- and a 5-iteration run."""
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
K = w["K_ne"] / 1e13
eng.set_log_model(eng.tensor(w["m"]), K)
t = eng.forward(eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3)), bench.TMAX, bench.NS).cpu().numpy().reshape(Na, -1)
dobs = t - t[0]
prob = parallel.ShardedRays(eng, o, d, bench.TMAX, bench.NS, dobs=dobs, cdct=np.full_like(dobs, 1e-4), i0=0)
m0 = eng.tensor(w["m"] - 0.05)
out = {}
ts = {}
for n in (5, 25):
    solvers.steepest_descent_log_model(prob, m0, K, max_iter=n, min_iter=n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, h = solvers.steepest_descent_log_model(prob, m0, K, max_iter=n, min_iter=n)
    torch.cuda.synchronize()
    ts[n] = time.perf_counter() - t0
    out["iterations_%d" % n] = len(h)
out["ms_per_iteration"] = (ts[25] - ts[5]) / 20 * 1e3
print(json.dumps(out))

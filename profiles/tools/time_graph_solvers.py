"""Eager vs hipGraph-replayed solver iterations (solvers.cgls / sirt, graph=True) at a single-timestep size (config 2:
2 604 rays, 128^3) and at the bench shape (260 400 rays, 256^3)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd import parallel, solvers, synthetic as syn
from ionotomo_amd.engine import RayEngine

out = {}
for name in ("cfg2", "bench"):
    if name == "cfg2":
        w = syn.make_workload("cfg2")
        Na, Ns, tmax = 62, 129, w["tmax"]
        o, d = w["origins"].reshape(Na, -1, 3), w["directions"].reshape(Na, -1, 3)
        xv, yv, zv, x_true = w["xvec"], w["yvec"], w["zvec"], w["ne"] / 1e13
    else:
        w = bench.build_workload(0)
        Na, Ns, tmax = bench.NA, bench.NS, bench.TMAX
        o, d = w["origins"].reshape(Na, -1, 3), w["directions"].reshape(Na, -1, 3)
        xv, yv, zv = w["xvec"], w["yvec"], w["zvec"]
        x_true = np.exp(w["m"]) * w["K_ne"] / 1e13
    eng = RayEngine(0)
    eng.set_grid(xv, yv, zv)
    eng.set_values(eng.tensor(x_true))
    t = eng.forward(eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3)), tmax, Ns).cpu().numpy().reshape(Na, -1)
    dobs = t - t[0]
    prob = parallel.ShardedRays(eng, o, d, tmax, Ns, dobs=dobs, cdct=np.full_like(dobs, 1e-4), i0=0)
    x0 = eng.tensor(x_true * 0.9)
    n = 200
    for sname, solve in (("cgls", solvers.cgls), ("sirt", solvers.sirt)):
        for graph in (False, True):
            solve(prob, x0, n_iter=5, graph=graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            solve(prob, x0, n_iter=n, graph=graph)
            torch.cuda.synchronize()
            out["%s_%s_%s_ms_per_iteration_incl_setup" % (name, sname, "graph" if graph else "eager")] = (time.perf_counter() - t0) / n * 1e3
    del eng, prob
print(json.dumps(out, indent=1))

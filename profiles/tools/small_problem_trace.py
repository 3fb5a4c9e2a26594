"""Where an inversion iteration spends its time at the reference's REAL problem size (one coherence window: config 2's 2 604 rays,
128^3): 300 CGLS iterations under `rocprofv3 --kernel-trace --stats` give the per-kernel device time; this script prints the
wall time per iteration next to it (eager and hipGraph replay).   rocprofv3 ... -- python3 profiles/tools/small_problem_trace.py"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ionotomo_amd import parallel, solvers, synthetic as syn
from ionotomo_amd.engine import RayEngine

w = syn.make_workload("cfg2")
Na, Ns, tmax = 62, 129, w["tmax"]
o, d = w["origins"].reshape(Na, -1, 3), w["directions"].reshape(Na, -1, 3)
eng = RayEngine(0)
eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
x_true = w["ne"] / 1e13
eng.set_values(eng.tensor(x_true))
t = eng.forward(eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3)), tmax, Ns).cpu().numpy().reshape(Na, -1)
dobs = t - t[0]
prob = parallel.ShardedRays(eng, o, d, tmax, Ns, dobs=dobs, cdct=np.full_like(dobs, 1e-4), i0=0)
x0 = eng.tensor(x_true * 0.9)
out = {"forward_plan": prob.forward_plan, "adjoint_plan": prob.plan}
for graph in (False, True):
    solvers.cgls(prob, x0, n_iter=5, graph=graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solvers.cgls(prob, x0, n_iter=300, graph=graph)
    torch.cuda.synchronize()
    out["cgls_us_per_iteration_%s" % ("graph" if graph else "eager")] = (time.perf_counter() - t0) / 300 * 1e6
for name, kw in (("sirt_us_per_iteration_eager", {}), ("sirt_us_per_iteration_separate_passes", {"small_pass": False}), ("cgls_us_per_iteration_separate_passes", {"small_pass": False})):
    fn = solvers.sirt if name.startswith("sirt") else solvers.cgls
    fn(prob, x0, n_iter=5, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(prob, x0, n_iter=300, **kw)
    torch.cuda.synchronize()
    out[name] = (time.perf_counter() - t0) / 300 * 1e6
print(json.dumps(out))

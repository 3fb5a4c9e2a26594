import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd import parallel, solvers
from ionotomo_amd.engine import RayEngine
wb = bench.build_workload(0)
eng = RayEngine(0)
eng.set_grid(wb["xvec"], wb["yvec"], wb["zvec"])
na, P = bench.NA, bench.NT * bench.ND
oo = wb["origins"].reshape(na, P, 3); dd = wb["directions"].reshape(na, P, 3)
x0 = np.exp(wb["m"]) * (wb["K_ne"] / 1e13)
prob = parallel.ShardedRays(eng, oo, dd, bench.TMAX, bench.NS, dobs=np.zeros((na, P)), cdct=np.full((na, P), 1e-6), i0=0)
eng.set_values(eng.tensor(x0 * 1.1))
prob.dobs = prob.forward().clone()
x0t = eng.tensor(x0)
for name in sys.argv[1:] or ["cgls"]:
    fn = getattr(solvers, name)
    fn(prob, x0t, n_iter=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn(prob, x0t, n_iter=30)
    torch.cuda.synchronize(); print(name, "ms/iter", (time.perf_counter() - t0) / 30 * 1e3, flush=True)

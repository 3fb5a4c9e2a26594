"""Effect of the measured partition on the tiled adjoint (bench workload)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
e = RayEngine(0); e.set_grid(w["xvec"], w["yvec"], w["zvec"]); e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
R = o.shape[0]
order = e.locality_order(o, d, bench.TMAX)
y = e.tensor(np.random.default_rng(0).normal(size=R))
g = torch.zeros(e.shape, dtype=torch.float64, device="cuda")
def launch():
    g.zero_(); e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g, order=order)
def timeit(n=20):
    for _ in range(3): launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): launch()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
out = {"before_ms": timeit()}
g_ref = g.clone()
for fr in ((1.0,), (0.75, 0.25)):
    st = e.tune_adjoint_partition(launch, R, fractions=fr)
    out[str(fr)] = dict(st, ms=timeit())
launch(); torch.cuda.synchronize()
out["max_abs_diff_vs_unpartitioned"] = float((g - g_ref).abs().max() / g_ref.abs().max())
print(json.dumps(out, indent=1))

"""Effect of the measured partition on the forward kernel (bench workload)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine
from ionotomo_amd import _lib
w = bench.build_workload(0)
e = RayEngine(0); e.set_grid(w["xvec"], w["yvec"], w["zvec"]); e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
R = o.shape[0]
tec = torch.empty(R, dtype=torch.float64, device="cuda")
def launch():
    e.forward(o, d, bench.TMAX, bench.NS, out=tec)
def timeit(n=50):
    for _ in range(5): launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): launch()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
out = {"before_ms": timeit()}
ref = tec.clone()
for refine in (1, 3, 6):
    st = e.tune_forward_partition(launch, R, refine=refine)
    cyc, _ = e.ctx.walk_cycles(_lib.WALK_FORWARD)
    out["refine%d" % refine] = dict(st, ms=timeit(), imbalance=float(cyc.max() / cyc.mean()))
launch(); torch.cuda.synchronize()
out["max_rel_diff"] = float(((tec - ref).abs() / ref.abs()).max())
print(json.dumps(out))

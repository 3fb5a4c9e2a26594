"""Forward kernel: re-balance only the eight XCD shares of the walk (equal counts within an XCD)."""
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
def timeit(n=100):
    for _ in range(5): launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): launch()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("equal", timeit())
launch(); cyc, nw = e.ctx.walk_cycles(_lib.WALK_FORWARD)
share = np.full(8, 1.0 / 8)
for it in range(5):
    per = nw // 8
    bounds = np.concatenate([[0], np.cumsum(share)]) * R
    starts = np.concatenate([np.linspace(bounds[x], bounds[x + 1], per, endpoint=False) for x in range(8)] + [[R]])
    starts = np.maximum.accumulate(np.rint(starts).astype(np.int64)); starts[0] = 0; starts[-1] = R
    e.ctx.walk_partition_set(_lib.WALK_FORWARD, starts, R)
    t = timeit()
    launch(); cyc, _ = e.ctx.walk_cycles(_lib.WALK_FORWARD)
    c = cyc.astype(float).reshape(8, per)
    xm, xmax = c.mean(1), c.max(1)
    print(it, "ms %.4f" % t, "share", share.round(4).tolist(), "xcd mean", (xm / xm.mean()).round(3).tolist(), "xcd max", (xmax / xmax.mean()).round(3).tolist(), flush=True)
    share = share * (xmax.mean() / xmax) ** 0.7
    share /= share.sum()

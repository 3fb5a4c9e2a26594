"""Ray-stationary (LDS tile per bundle) vs node-stationary (box-binned) back-projection at the bench shape."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


w = bench.build_workload(0)
R = w["origins"].shape[0]
out = {}
for kind in ("linear", "cubic"):
    e = RayEngine(0, interp=kind)
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
    o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
    y = torch.randn(R, dtype=torch.float64, device="cuda")
    g = torch.zeros(e.shape, dtype=torch.float64, device="cuda")
    order = e.locality_order(o, d, bench.TMAX)

    def adj_tile():
        g.zero_()
        e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g, order=order)
    if kind == "linear":
        e.tune_adjoint_partition(adj_tile, R)
    out[kind + "_ray_stationary_ms"] = timeit(adj_tile, 5 if kind == "linear" else 2, 1)
    t0 = time.perf_counter()
    info = e.plan_adjoint(o, d, bench.TMAX, bench.NS)
    out[kind + "_plan_build_s"] = time.perf_counter() - t0
    out[kind + "_plan"] = {"segments": info[0], "units": info[1], "outside_fraction": info[2]}

    def adj_bin():
        g.zero_()
        e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g)
    out[kind + "_node_stationary_ms"] = timeit(adj_bin, 5 if kind == "linear" else 2, 1)
    out[kind + "_memset_ms"] = timeit(lambda: g.zero_(), 10, 2)
    del e
print(json.dumps(out, indent=1))

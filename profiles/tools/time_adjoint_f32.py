"""Back-projection with float32 accumulation (LDS image and global sum in float32) against float64, at the bench shape."""
import json, os, sys
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
e = RayEngine(0)
e.set_grid(w["xvec"], w["yvec"], w["zvec"])
e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
y = torch.randn(R, dtype=torch.float64, device="cuda")
out = {}
res = {}
order = e.locality_order(o, d, bench.TMAX)
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):      # ray-stationary kernel (no plan yet)
    g = torch.zeros(e.shape, dtype=dt, device="cuda")

    def adj_tile():
        g.zero_()
        e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g, order=order)
    out["ray_stationary_%s_ms" % name] = timeit(adj_tile, 5, 1)
e.plan_adjoint(o, d, bench.TMAX, bench.NS)
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    g = torch.zeros(e.shape, dtype=dt, device="cuda")

    def adj():
        g.zero_()
        e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g)
    out["adjoint_%s_ms" % name] = timeit(adj)
    res[name] = g.double().clone()
    out["zero_%s_ms" % name] = timeit(lambda: g.zero_())
out["f32_vs_f64_max_rel"] = float((res["f32"] - res["f64"]).abs().max() / res["f64"].abs().max())
print(json.dumps(out))

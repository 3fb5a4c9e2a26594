"""Tricubic timings at the bench shape (260,400 rays, 256^3) and at config 2: fast (Lekien-Marsden fields) vs general
(216-tap) forward, field rebuild, transpose (8 channel scatters + fold).  Prints one JSON object."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd import synthetic as syn
from ionotomo_amd.engine import RayEngine


def timeit(fn, n=5, warm=2):
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


out = {}
w = bench.build_workload(0)
R = w["origins"].shape[0]
for tier, env in (("fast", None), ("general", "4")):
    if env:
        os.environ["IONOTOMO_VARIANT"] = env
    else:
        os.environ.pop("IONOTOMO_VARIANT", None)
    e = RayEngine(0, interp="cubic")
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    m_t = e.tensor(w["m"])
    e.set_log_model(m_t, w["K_ne"] / 1e13)
    o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
    tec = torch.empty(R, dtype=torch.float64, device="cuda")
    order = e.locality_order(o, d, bench.TMAX)
    out["forward_%s_ms" % tier] = timeit(lambda: e.forward(o, d, bench.TMAX, bench.NS, out=tec))
    if tier == "fast":
        def refresh():
            e.set_log_model(m_t, w["K_ne"] / 1e13)
            e.forward(o, d, bench.TMAX, bench.NS, out=tec)
        out["forward_fast_with_field_rebuild_ms"] = timeit(refresh)
        out["set_log_model_ms"] = timeit(lambda: e.set_log_model(m_t, w["K_ne"] / 1e13))
    y = torch.randn(R, dtype=torch.float64, device="cuda")
    g = torch.zeros(e.shape, dtype=torch.float64, device="cuda")
    if tier == "fast":
        out["adjoint_fast_ordered_ms"] = timeit(lambda: e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g, order=order), 3, 1)
    else:
        sub = slice(0, 2604 * 4)
        os_, ds_, ys_ = o[sub].contiguous(), d[sub].contiguous(), y[sub].contiguous()
        out["adjoint_general_10416_rays_ms"] = timeit(lambda: e.adjoint(os_, ds_, ys_, bench.TMAX, bench.NS, out=g), 2, 1)
    assert not e.check_oob()
    del e
os.environ.pop("IONOTOMO_VARIANT", None)
w2 = syn.make_workload("cfg2")
e = RayEngine(0, interp="cubic")
e.set_grid(w2["xvec"], w2["yvec"], w2["zvec"])
e.set_log_model(e.tensor(w2["m"]), w2["K_ne"] / 1e13)
o, d = e.tensor(w2["origins"].reshape(-1, 3)), e.tensor(w2["directions"].reshape(-1, 3))
out["cfg2_forward_fast_us"] = timeit(lambda: e.forward(o, d, w2["tmax"], w2["Ns"]), 20, 3) * 1e3
out["rays"] = R
out["forward_fast_ray_integrals_per_s"] = R / out["forward_fast_ms"] * 1e3
print(json.dumps(out, indent=1))

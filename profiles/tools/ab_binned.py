"""Planned back-projections at the bench shape for ONE build of the library (IONOTOMO_LIB selects an A/B build): trilinear and
tricubic, float atomics and fixed point, plan figures, checked against the unplanned kernel of the same build.

    IONOTOMO_LIB=build_ab/libionotomo_r4bin.so python profiles/tools/ab_binned.py      (one JSON line)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine


def timeit(fn, n=10, warm=3):
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
torch.manual_seed(1)
out = {"lib": os.environ.get("IONOTOMO_LIB", "default")}
for interp in ("linear", "cubic"):
    e = RayEngine(0, interp=interp)
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
    o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
    y = torch.randn(R, dtype=torch.float64, device="cuda")
    ref = e.adjoint(o, d, y, bench.TMAX, bench.NS).clone()                     # unplanned
    info = e.plan_adjoint(o, d, bench.TMAX, bench.NS)
    out[interp + "_plan"] = {"segments": info[0], "units": info[1], "outside": info[2], "lanes": e.plan_segment_lanes()}
    g = torch.zeros(e.shape, dtype=torch.float64, device="cuda")
    for name, on in (("float", False), ("fixed", True)):
        e.set_deterministic(on)

        def run():
            g.zero_()
            e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g)
        out["%s_%s_ms" % (interp, name)] = timeit(run)
        out["%s_%s_rel_err_vs_unplanned" % (interp, name)] = float((g - ref).abs().max() / ref.abs().max())
    e.set_deterministic(False)
    assert not e.check_oob()
print(json.dumps(out))

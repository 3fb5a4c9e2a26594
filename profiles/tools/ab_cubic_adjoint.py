"""Planned tricubic transpose at the bench shape: folds + zeroing over the tiles the plan's rays reach (default) against the
whole grid (IONOTOMO_VARIANT=23).  Prints times and the largest difference between the two results."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine

w = bench.build_workload(0)
R = w["origins"].shape[0]
res, out = {}, {}
for name, env in (("tiles", None), ("whole_grid", "23"), ("tiles_deterministic", None)):
    if env:
        os.environ["IONOTOMO_VARIANT"] = env
    else:
        os.environ.pop("IONOTOMO_VARIANT", None)
    e = RayEngine(0, interp="cubic")
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
    o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
    torch.manual_seed(1)
    y = torch.randn(R, dtype=torch.float64, device="cuda")
    e.plan_adjoint(o, d, bench.TMAX, bench.NS)
    e.set_deterministic(name.endswith("deterministic"))
    g = torch.zeros(e.shape, dtype=torch.float64, device="cuda")
    for _ in range(2):
        e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(5):
        e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g)
    b.record()
    torch.cuda.synchronize()
    out[name + "_ms"] = a.elapsed_time(b) / 5
    g.zero_()
    e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g)
    res[name] = g.clone()
    assert not e.check_oob()
    del e
os.environ.pop("IONOTOMO_VARIANT", None)
out["max_abs_diff"] = float((res["tiles"] - res["whole_grid"]).abs().max())
out["max_abs_diff_deterministic_vs_float"] = float((res["tiles_deterministic"] - res["tiles"]).abs().max())
out["max_abs"] = float(res["whole_grid"].abs().max())
out["nonzero_fraction"] = float((res["whole_grid"] != 0).double().mean())
print(json.dumps(out))

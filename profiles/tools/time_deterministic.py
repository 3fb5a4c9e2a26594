"""Planned trilinear back-projection at the bench shape: float atomics (default) against the deterministic fixed-point mode
(iono_set_deterministic).  Prints times, the largest difference, and whether two deterministic launches agree bit for bit."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine

w = bench.build_workload(0)
R = w["origins"].shape[0]
e = RayEngine(0)
e.set_grid(w["xvec"], w["yvec"], w["zvec"])
e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
torch.manual_seed(1)
y = torch.randn(R, dtype=torch.float64, device="cuda")
e.plan_adjoint(o, d, bench.TMAX, bench.NS)
g = torch.zeros(e.shape, dtype=torch.float64, device="cuda")
out, res = {}, {}
for name, on in (("float_atomics", False), ("fixed_point", True)):
    e.set_deterministic(on)
    for _ in range(3):
        e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g)
    b.record()
    torch.cuda.synchronize()
    out[name + "_ms"] = a.elapsed_time(b) / 10
    res[name] = [e.adjoint(o, d, y, bench.TMAX, bench.NS).clone() for _ in range(2)]
assert not e.check_oob()
out["float_runs_bit_equal"] = bool(torch.equal(*res["float_atomics"]))
out["fixed_runs_bit_equal"] = bool(torch.equal(*res["fixed_point"]))
out["max_rel_diff_fixed_vs_float"] = float((res["fixed_point"][0] - res["float_atomics"][0]).abs().max() / res["float_atomics"][0].abs().max())
out["max_rel_diff_float_vs_float"] = float((res["float_atomics"][1] - res["float_atomics"][0]).abs().max() / res["float_atomics"][0].abs().max())
print(json.dumps(out))

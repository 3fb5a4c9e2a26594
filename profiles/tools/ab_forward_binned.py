"""A/B: direct forward (one wave per ray, 4 global dwordx4 gathers per 64 samples) vs node-stationary forward on the
ray plan (box image staged in LDS).  Parity of both against each other and against the C oracle sample; timing."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine
from oracle import oracle_c as OC, oracle as O


def timeit(fn, n=20, warm=3):
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
res = {}
for name, env in (("direct", "0"), ("binned", "1")):
    os.environ["IONOTOMO_FWD_PLAN"] = env
    e = RayEngine(0)
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
    o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
    tec = torch.empty(R, dtype=torch.float64, device="cuda")
    e.plan_adjoint(o, d, bench.TMAX, bench.NS)
    out[name + "_ms"] = timeit(lambda: e.forward(o, d, bench.TMAX, bench.NS, out=tec))
    assert not e.check_oob()
    res[name] = tec.cpu().numpy()
    del e
out["max_rel_diff_binned_vs_direct"] = float(np.max(np.abs(res["binned"] - res["direct"]) / np.abs(res["direct"])))
idx = np.arange(0, R, 401)
ref = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], O.ne_from_log_model(w["m"], w["K_ne"]), w["origins"][idx], w["directions"][idx], bench.TMAX, bench.NS)
out["max_rel_err_binned_vs_oracle"] = float(np.max(np.abs(res["binned"][idx] - ref) / np.abs(ref)))
print(json.dumps(out, indent=1))

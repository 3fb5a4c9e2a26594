"""Cost of the geometry-only set-up of an inversion at the bench shape: the back-projection plan (device-built), the
walk orders."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
e = RayEngine(0)
e.set_grid(w["xvec"], w["yvec"], w["zvec"])
e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
out = {}


def timed(name, fn, n=4):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    out[name + "_first_ms"], out[name + "_again_ms"] = ts[0], min(ts[1:])
    return r


info = timed("plan_adjoint", lambda: e.plan_adjoint(o, d, bench.TMAX, bench.NS))
out["plan"] = {"segments": info[0], "units": info[1]}
timed("coherent_order", lambda: e.coherent_order(o, d))
timed("locality_order", lambda: e.locality_order(o, d, bench.TMAX))
print(json.dumps(out))

"""Tricubic forward at the bench shape: bundle plan (k_forward_bundle_lm) against lanes = samples (k_forward_straight_lm).
python profiles/tools/time_cubic_bundle.py            (IONOTOMO_LIB selects an A/B build)"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ionotomo_amd.engine import RayEngine

w = bench.build_workload(0)
eng = RayEngine(0, interp="cubic")
eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
eng.set_values(eng.tensor(np.exp(w["m"])))
ot, dt = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
out = {"lib": os.environ.get("IONOTOMO_LIB", "default")}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for name in ("direct", "bundle"):
    if name == "bundle":
        out["plan"] = eng.plan_forward(ot, dt, bench.TMAX, bench.NS)
    for _ in range(5):
        eng.forward(ot, dt, bench.TMAX, bench.NS)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(30):
        eng.forward(ot, dt, bench.TMAX, bench.NS)
    torch.cuda.synchronize()
    out[name + "_ms"] = (time.perf_counter() - t) / 30 * 1e3
    # with new values every call: the field arrays are rebuilt
    x = eng.tensor(np.exp(w["m"]))
    t = time.perf_counter()
    for _ in range(10):
        eng.set_values(x)
        eng.forward(ot, dt, bench.TMAX, bench.NS)
    torch.cuda.synchronize()
    out[name + "_with_field_rebuild_ms"] = (time.perf_counter() - t) / 10 * 1e3
print(json.dumps(out))

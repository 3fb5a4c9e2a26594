import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
R = w["origins"].shape[0]
for storage in ("f64", "f32"):
    e = RayEngine(0, storage=storage, interp="cubic")
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
    o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
    out = torch.empty(R, dtype=torch.float64, device="cuda")
    for _ in range(2):
        e.forward(o, d, bench.TMAX, bench.NS, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5):
        e.forward(o, d, bench.TMAX, bench.NS, out=out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print("tricubic %s: %.3f ms per %d-ray launch -> %.3e ray-integrals/s; oob=%s" % (storage, ms, R, R / ms * 1e3, e.check_oob()), flush=True)

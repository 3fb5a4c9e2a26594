"""Forward kernel on a grid larger than the 256 MiB Infinity Cache (512^3 float64 = 1 GiB, Ns = 513):
the regime where the gathers really come from HBM."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd import synthetic as syn
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
R = w["origins"].shape[0]
for n, storage in ((512, "f64"), (512, "f32"), (640, "f32")):
    Ns = n + 1
    xv = np.linspace(w["xvec"][0], w["xvec"][-1], n)
    yv = np.linspace(w["yvec"][0], w["yvec"][-1], n)
    zv = np.linspace(w["zvec"][0], w["zvec"][-1], n)
    e = RayEngine(0, storage=storage)
    e.set_grid(xv, yv, zv)
    prof = syn.chapman_profile(np.maximum(zv, 0.0)) / 1e13
    m = torch.from_numpy(np.log(prof)).cuda()[None, None, :].expand(n, n, n).contiguous()
    e.set_log_model(m.reshape(-1), 1.0)
    del m
    o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
    out = torch.empty(R, dtype=torch.float64, device="cuda")
    for _ in range(2):
        e.forward(o, d, bench.TMAX, Ns, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(5):
        e.forward(o, d, bench.TMAX, Ns, out=out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    esz = 8 if storage == "f64" else 4
    alg = R * (Ns * 8 * esz + 56)
    print("grid %d^3 %s (%.2f GiB), Ns=%d: %.3f ms -> %.3e ray-integrals/s, algorithmic %.1f TB/s, oob=%s" % (
        n, storage, n ** 3 * esz / 2 ** 30, Ns, ms, R / ms * 1e3, alg / ms / 1e9, e.check_oob()), flush=True)
    del e, o, d, out
    torch.cuda.empty_cache()

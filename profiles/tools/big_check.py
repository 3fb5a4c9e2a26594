"""Scale check beyond the bench shape: 512^3 float64 grid (1 GiB), 620 000 rays (config 4's ray count), Ns = 513 -- dot-product
test of forward and planned back-projection, planned vs ray-stationary back-projection, float32 block layout, C-oracle
sample; timings."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ionotomo_amd import synthetic as syn
from ionotomo_amd.engine import RayEngine
from oracle import oracle_c as OC


def timeit(fn, n=5, warm=1):
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
w = syn.make_workload("cfg4")                       # 62 x 100 x 100 rays, 256^3 axes
n = 512
xv, yv, zv = (np.linspace(w[k][0], w[k][-1], n) for k in ("xvec", "yvec", "zvec"))
Ns = n + 1
e = RayEngine(0)
e.set_grid(xv, yv, zv)
rng = np.random.default_rng(0)
prof = syn.chapman_profile(np.maximum(zv, 0.0)) / 1e13
M = torch.from_numpy(prof).cuda()[None, None, :] * (1.0 + 0.1 * torch.rand(n, n, n, dtype=torch.float64, device="cuda"))
e.set_values(M.reshape(-1))
o, d = e.tensor(w["origins"].reshape(-1, 3)), e.tensor(w["directions"].reshape(-1, 3))
R = o.shape[0]
out["rays"], out["grid"], out["Ns"] = R, n, Ns
tec = torch.empty(R, dtype=torch.float64, device="cuda")
order = e.coherent_order(o, d)
out["forward_ms"] = timeit(lambda: e.forward(o, d, w["tmax"], Ns, out=tec, order=order))
assert not e.check_oob()
# bundle-stationary forward on its geometry plan (round 3)
t0 = time.perf_counter()
out["forward_plan"] = e.plan_forward(o, d, w["tmax"], Ns)
out["forward_plan_build_ms"] = (time.perf_counter() - t0) * 1e3
tecp = torch.empty_like(tec)
out["forward_planned_ms"] = timeit(lambda: e.forward(o, d, w["tmax"], Ns, out=tecp))
out["planned_vs_unplanned_max_rel"] = float(((tecp - tec).abs() / tec.abs()).max())
out["forward_planned_algorithmic_gbs"] = R * (Ns * 64 + 56) / out["forward_planned_ms"] / 1e6
e.clear_forward_plan()
sel = rng.choice(R, 400, replace=False)
ref = OC.forward_tec_straight(xv, yv, zv, M.cpu().numpy(), w["origins"].reshape(-1, 3)[sel], w["directions"].reshape(-1, 3)[sel], w["tmax"], Ns)
out["forward_vs_c_oracle_max_rel"] = float(np.max(np.abs(tec.cpu().numpy()[sel] - ref) / np.abs(ref)))
y = torch.randn(R, dtype=torch.float64, device="cuda")
g0 = torch.zeros(e.shape, dtype=torch.float64, device="cuda")
mo = e.locality_order(o, d, w["tmax"])
out["adjoint_ray_stationary_ms"] = timeit(lambda: (g0.zero_(), e.adjoint(o, d, y, w["tmax"], Ns, out=g0, order=mo)), 3, 1)
t0 = time.perf_counter()
info = e.plan_adjoint(o, d, w["tmax"], Ns)
torch.cuda.synchronize()
out["plan_build_ms"], out["plan"] = (time.perf_counter() - t0) * 1e3, {"segments": info[0], "units": info[1], "outside": info[2]}
g1 = torch.zeros(e.shape, dtype=torch.float64, device="cuda")
out["adjoint_node_stationary_ms"] = timeit(lambda: (g1.zero_(), e.adjoint(o, d, y, w["tmax"], Ns, out=g1)), 3, 1)
out["planned_vs_ray_stationary_max_rel"] = float((g1 - g0).abs().max() / g0.abs().max())
lhs, rhs = float((tec * y).sum()), float((M * g1).sum())
out["dot_product_rel"] = abs(lhs - rhs) / abs(lhs)
del g0, g1
e32 = RayEngine(0, storage="f32")
e32.set_grid(xv, yv, zv)
e32.set_values(M.reshape(-1))
t32 = torch.empty_like(tec)
out["forward_f32_blocks_ms"] = timeit(lambda: e32.forward(o, d, w["tmax"], Ns, out=t32, order=order))
out["f32_vs_f64_max_rel"] = float(((t32 - tec).abs() / tec.abs()).max())
print(json.dumps(out, indent=1))

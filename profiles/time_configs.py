#!/usr/bin/env python3
"""Timings of the BASELINE.json configs other than the bench line (these are parity-test cases, not
bench lines; this script records how fast they run on one MI355X).  Prints one JSON object.

cfg2  62 x 42 x 1, 128^3, trilinear and tricubic forward (Ns = 129)
cfg3  same geometry, Fermat curved-ray tracer (true bending, tricubic n) + TEC along the traced rays
cfg5  50 CGLS and 50 SIRT iterations (forward + adjoint each) on the 256^3 grid with this GPU's share
      of config 4 (62 x 42 x 100 rays), synthetic data from a perturbed model; objective history
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ionotomo_amd import parallel, solvers, synthetic as syn  # noqa: E402
from ionotomo_amd.engine import RayEngine  # noqa: E402


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


out = {}
# ---------------------------------------------------------------- cfg2 / cfg3
w = syn.make_workload("cfg2")
R = w["origins"].reshape(-1, 3).shape[0]
for kind in ("linear", "cubic"):
    eng = RayEngine(0, interp=kind)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    eng.set_values(eng.tensor(w["ne"] / 1e13))
    o, d = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
    ms = timeit(lambda: eng.forward(o, d, w["tmax"], w["Ns"]), 50, 5)
    out["cfg2_%s_forward_us" % kind] = ms * 1e3
    out["cfg2_%s_ray_integrals_per_s" % kind] = R / ms * 1e3
# cfg3: same rays, grid with a wider margin (this synthetic, strongly turbulent ionosphere bends 120 MHz
# rays by kilometres; rays leaving the grid raise, exactly like the reference's bounds_error=True)
w = syn.make_workload("cfg2", margin_cells=16)
eng = RayEngine(0, interp="linear")
eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
eng.set_values(eng.tensor(w["ne"]))                      # ne [m^-3] for the tracer
o, d = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
rays = torch.empty((R, 4, w["Ns"]), dtype=torch.float64, device="cuda")
for kind, sub in (("cubic", 4), ("linear", 4)):
    ms = timeit(lambda: eng.trace_fermat(o, d, w["tmax"], w["Ns"], 120e6, bend=True, kind=kind, substeps=sub, out=rays), 10, 2)
    out["cfg3_fermat_trace_%s_ms" % kind] = ms
    out["cfg3_fermat_%s_rays_per_s" % kind] = R / ms * 1e3
ms = timeit(lambda: eng.forward_rays(rays), 50, 5)
out["cfg3_tec_along_traced_rays_us"] = ms * 1e3
eng.trace_fermat(o, d, w["tmax"], w["Ns"], 120e6, bend=True, kind="cubic", substeps=4, out=rays)
straight = eng.forward(o, d, w["tmax"], w["Ns"])
curved = eng.forward_rays(rays)
out["cfg3_max_lateral_bending_km"] = float((rays[:, 0, -1] - (o[:, 0] + d[:, 0] / d[:, 2] * (w["tmax"] - o[:, 2]))).abs().max())
out["cfg3_max_rel_tec_change_from_bending_120MHz"] = float(((curved - straight).abs() / straight.abs()).max())
assert not eng.check_oob()
# throughput of the tracer once the batch fills the chip: the same fan x 100 (260 400 rays, jittered origins)
ob = (o[None] + 0.05 * torch.randn((100, 1, 3), dtype=torch.float64, device="cuda") * torch.tensor([1.0, 1.0, 0.0], device="cuda",
                                                                                            dtype=torch.float64)).reshape(-1, 3).contiguous()
db = d[None].expand(100, -1, -1).reshape(-1, 3).contiguous()
big = torch.empty((ob.shape[0], 4, w["Ns"]), dtype=torch.float64, device="cuda")
for kind in ("linear", "cubic"):
    ms = timeit(lambda: eng.trace_fermat(ob, db, w["tmax"], w["Ns"], 120e6, bend=True, kind=kind, substeps=4, out=big), 3, 1)
    out["fermat_full_batch_%s_ms" % kind] = ms
    out["fermat_full_batch_%s_rays_per_s" % kind] = ob.shape[0] / ms * 1e3
assert not eng.check_oob()
del big

# ---------------------------------------------------------------- fused curved-ray forward / transpose (round 3: no rays[R,4,Ns])
w3 = syn.make_workload("cfg2", margin_cells=16)
e3 = RayEngine(0)
e3.set_grid(w3["xvec"], w3["yvec"], w3["zvec"])
e3.set_values(e3.tensor(w3["ne"]))
o3, d3 = e3.tensor(w3["origins"].reshape(-1, 3)), e3.tensor(w3["directions"].reshape(-1, 3))
for kind in ("linear", "cubic"):
    out["cfg3_fused_forward_%s_ms" % kind] = timeit(lambda: e3.forward_fermat(o3, d3, w3["tmax"], w3["Ns"], 120e6, bend=True, kind=kind, substeps=4, fused=True), 10, 2)
    out["cfg3_forward_fermat_%s_default_path_ms" % kind] = timeit(lambda: e3.forward_fermat(o3, d3, w3["tmax"], w3["Ns"], 120e6, bend=True, kind=kind, substeps=4), 10, 2)
y3 = torch.randn(o3.shape[0], dtype=torch.float64, device="cuda")
g3 = torch.zeros(e3.shape, dtype=torch.float64, device="cuda")
out["cfg3_fused_adjoint_linear_ms"] = timeit(lambda: e3.adjoint_fermat(o3, d3, y3, w3["tmax"], w3["Ns"], 120e6, bend=True, kind="linear", substeps=4, out=g3), 5, 1)
w4 = syn.make_workload("cfg4", margin_cells=16)
e4 = RayEngine(0)
e4.set_grid(w4["xvec"], w4["yvec"], w4["zvec"])
e4.set_values(e4.tensor(w4["ne"]))
o4, d4 = e4.tensor(w4["origins"].reshape(-1, 3)), e4.tensor(w4["directions"].reshape(-1, 3))
t4 = torch.empty(o4.shape[0], dtype=torch.float64, device="cuda")
out["fused_620k_bending_rays_256_forward_ms"] = timeit(lambda: e4.forward_fermat(o4, d4, w4["tmax"], w4["Ns"], 150e6, bend=True, kind="linear", substeps=2, out=t4), 3, 1)
out["fused_620k_bending_rays_per_s"] = o4.shape[0] / out["fused_620k_bending_rays_256_forward_ms"] * 1e3
y4 = torch.randn(o4.shape[0], dtype=torch.float64, device="cuda")
g4 = torch.zeros(e4.shape, dtype=torch.float64, device="cuda")
out["fused_620k_bending_rays_256_adjoint_ms"] = timeit(lambda: e4.adjoint_fermat(o4, d4, y4, w4["tmax"], w4["Ns"], 150e6, bend=True, kind="linear", substeps=2, out=g4), 2, 1)
# round 4: 620 000 bending rays through a TRICUBIC index without rays[R,4,Ns] (k_fermat_tec_lm; trilinear integrand of this engine)
out["fused_620k_bending_rays_256_cubic_index_forward_ms"] = timeit(lambda: e4.forward_fermat(o4, d4, w4["tmax"], w4["Ns"], 150e6, bend=True, kind="cubic", substeps=2, out=t4), 2, 1)
del e4, o4, d4, t4, y4, g4, w4

# ---------------------------------------------------------------- PCIe-inclusive facade call (host numpy in/out)
import ionotomo_amd as it  # noqa: E402
w2 = syn.make_workload("cfg2")
rays_h = it.calc_rays(w2["origins"][:, 0, 0, :], w2["directions"][0], [0.0], None, None, None,
                      it.TriCubic(w2["xvec"], w2["yvec"], w2["zvec"], w2["ne"]), 120e6, True, w2["tmax"], w2["Ns"])
m_tci = it.TriCubic(w2["xvec"], w2["yvec"], w2["zvec"], w2["m"])
from ionotomo_amd import _lib as _il  # noqa: E402
ts = []
for _ in range(5):                       # every call from a cold cache: grid + 10.7 MB of rays over PCIe
    _il.default_context().forget()
    t0 = time.perf_counter()
    it.forward_equation(rays_h, w2["K_ne"], m_tci, 0)
    ts.append(time.perf_counter() - t0)
dt = float(np.median(ts))
out["cfg2_facade_forward_equation_ms_pcie_inclusive"] = dt * 1e3
out["cfg2_facade_ray_integrals_per_s_pcie_inclusive"] = rays_h.shape[0] * rays_h.shape[2] / dt
it.forward_equation(rays_h, w2["K_ne"], m_tci, 0)
ts = []
for _ in range(50):                      # the same rays and model again (a line search's repeated evaluations): resident operands
    t0 = time.perf_counter()
    it.forward_equation(rays_h, w2["K_ne"], m_tci, 0)
    ts.append(time.perf_counter() - t0)
out["cfg2_facade_forward_equation_us_resident_second_call"] = float(np.median(ts)) * 1e6
ts = []
for k in range(20):                      # a new model every call (m + alpha dm), the same rays
    m2 = it.TriCubic(w2["xvec"], w2["yvec"], w2["zvec"], w2["m"] + 1e-3 * k)
    t0 = time.perf_counter()
    it.forward_equation(rays_h, w2["K_ne"], m2, 0)
    ts.append(time.perf_counter() - t0)
out["cfg2_facade_forward_equation_us_resident_rays_new_model"] = float(np.median(ts)) * 1e6

# ---------------------------------------------------------------- cfg5 (one GPU's share)
wb = bench.build_workload(0)
eng = RayEngine(0)
eng.set_grid(wb["xvec"], wb["yvec"], wb["zvec"])
na, P = bench.NA, bench.NT * bench.ND
oo = wb["origins"].reshape(na, P, 3)
dd = wb["directions"].reshape(na, P, 3)
x0 = np.exp(wb["m"]) * (wb["K_ne"] / 1e13)                               # prior model (TECU/km)
rng = np.random.default_rng(3)
X, Y, Z = np.meshgrid(wb["xvec"], wb["yvec"], wb["zvec"], indexing="ij")
blob = 1.0 + 0.3 * np.exp(-((X - 5) ** 2 + (Y + 8) ** 2) / 15.0 ** 2 - ((Z - 300) / 80.0) ** 2)
x_true = x0 * blob
prob = parallel.ShardedRays(eng, oo, dd, bench.TMAX, bench.NS, dobs=np.zeros((na, P)), cdct=np.full((na, P), 1e-6), i0=0)
eng.set_values(eng.tensor(x_true))
prob.dobs = prob.forward() + eng.tensor(rng.normal(size=na * P) * 1e-3)
for name, fn in (("cgls", lambda: solvers.cgls(prob, eng.tensor(x0), n_iter=50)),
                 ("sirt", lambda: solvers.sirt(prob, eng.tensor(x0), n_iter=50))):
    fn3 = {"cgls": lambda: solvers.cgls(prob, eng.tensor(x0), n_iter=3), "sirt": lambda: solvers.sirt(prob, eng.tensor(x0), n_iter=3)}
    fn3[name]()                                   # warm-up (allocator, caches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, hist = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    err0 = float(np.linalg.norm(x0 - x_true) / np.linalg.norm(x_true - x0 + 1e-300))
    err = float((x - eng.tensor(x_true)).norm() / (eng.tensor(x0) - eng.tensor(x_true)).norm())
    out["cfg5_%s_50_iterations_s" % name] = dt
    out["cfg5_%s_ms_per_iteration" % name] = dt / 50 * 1e3
    out["cfg5_%s_objective_first_last" % name] = [hist[0], hist[-1]]
    out["cfg5_%s_model_error_vs_prior_error" % name] = err
print(json.dumps(out, indent=1))

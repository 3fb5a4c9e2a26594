"""Phase observable at the bench shape (260 400 rays, Ns = 257, 256^3): forward g[Na,Nt,Nd,Nf] and its adjoint w.r.t. the
log-model, for 1, 2 and 4 frequencies."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine


def timeit(fn, n=10, warm=2):
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
e = RayEngine(0)
e.set_grid(w["xvec"], w["yvec"], w["zvec"])
e.set_log_model(e.tensor(w["m"]), w["K_ne"])                       # ne in m^-3: the phase model needs physical densities
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
order = e.locality_order(o, d, bench.TMAX)
clock = torch.zeros(bench.NA, bench.NT, dtype=torch.float64, device="cuda")
const = torch.zeros(bench.NA, dtype=torch.float64, device="cuda")
out = {}
for planned in (False, True):
  if planned:
    e.plan_adjoint(o, d, bench.TMAX, bench.NS)
  for nf in (1, 2, 4, 8):
    freqs = np.linspace(120e6, 160e6, nf)
    g = torch.empty(bench.NA, bench.NT, bench.ND, nf, dtype=torch.float64, device="cuda")
    if not planned:
        out["forward_phase_nf%d_ms" % nf] = timeit(lambda: e.forward_phase(o, d, bench.NA, bench.NT, bench.ND, bench.TMAX, bench.NS, freqs, clock, const, 0, out=g))
        g0 = g.clone()
        e.plan_forward(o, d, bench.TMAX, bench.NS)            # bundle plan (round 3): the same observable from LDS windows
        out["forward_phase_bundle_nf%d_ms" % nf] = timeit(lambda: e.forward_phase(o, d, bench.NA, bench.NT, bench.ND, bench.TMAX, bench.NS, freqs, clock, const, 0, out=g))
        out["forward_phase_bundle_vs_direct_nf%d_max_abs" % nf] = float((g - g0).abs().max())
        out["forward_phase_nf%d_max_abs_value" % nf] = float(g0.abs().max())
        e.clear_forward_plan()
    y = torch.randn(bench.NA, bench.NT * bench.ND, nf, dtype=torch.float64, device="cuda")
    grad = torch.zeros(e.shape, dtype=torch.float64, device="cuda")

    def adj():
        grad.zero_()
        e.adjoint_phase(o, d, y, bench.NA, bench.TMAX, bench.NS, freqs, 0, order=order, out=grad)
    out["adjoint_phase_%s_nf%d_ms" % ("node_stationary" if planned else "ray_stationary", nf)] = timeit(adj, 5, 1)
print(json.dumps(out))

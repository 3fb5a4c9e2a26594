"""Variants of the partition update rule."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
e = RayEngine(0); e.set_grid(w["xvec"], w["yvec"], w["zvec"]); e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
R = o.shape[0]
order = e.locality_order(o, d, bench.TMAX)
y = e.tensor(np.random.default_rng(0).normal(size=R))
g = torch.zeros(e.shape, dtype=torch.float64, device="cuda")
def launch():
    g.zero_(); e.adjoint(o, d, y, bench.TMAX, bench.NS, out=g, order=order)
def timeit(n=20):
    for _ in range(3): launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): launch()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def tune(rounds, alpha, smooth):
    e.ctx.adjoint_partition_set(None, R)
    launch(); cyc = e.ctx.adjoint_block_cycles().astype(float); nb = cyc.size
    base, rem = divmod(R, nb)
    starts = np.array([b * base + min(b, rem) for b in range(nb + 1)], dtype=np.int64)
    dens = None
    hist = []
    for it in range(rounds):
        launch(); cyc = e.ctx.adjoint_block_cycles().astype(float)
        hist.append(round(cyc.max() / cyc.mean(), 3))
        lens = np.maximum(np.diff(starts), 1)
        d_now = cyc / lens                                    # cost per ray in each chunk
        # per-RAY density estimate kept across rounds on a fine grid (resolution 32 rays), relaxed towards the new measurement
        fine = np.repeat(d_now, np.diff(starts)) if dens is None else dens + alpha * (np.repeat(d_now, np.diff(starts)) - dens)
        dens = fine
        if smooth:
            k = np.ones(smooth) / smooth
            fine = np.convolve(np.pad(fine, smooth // 2, mode="edge"), k, mode="valid")[:R]
        cum = np.concatenate([[0.0], np.cumsum(fine)])
        tgt = cum[-1] * np.arange(nb + 1) / nb
        new = np.searchsorted(cum, tgt).astype(np.int64)
        new = np.maximum.accumulate(new); new[0] = 0; new[-1] = R
        starts = new
        e.ctx.adjoint_partition_set(starts, R)
    return hist
for rounds, alpha, smooth in ((6, 1.0, 0), (10, 0.5, 0), (10, 0.7, 65), (16, 0.4, 33)):
    h = tune(rounds, alpha, smooth)
    print(rounds, alpha, smooth, "ms %.4f" % timeit(), h, flush=True)

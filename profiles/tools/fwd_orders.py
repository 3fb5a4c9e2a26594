"""Forward kernel under different walk orders: what matters is that ONE wave's consecutive rays are coherent (L1 reuse
in time) while concurrently running waves are far apart (no L2 channel hammering)."""
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
Na, Nt, Nd = bench.NA, bench.NT, bench.ND
idx = torch.arange(R, dtype=torch.int32, device="cuda").reshape(Na, Nt, Nd)
orders = {"memory [Na][Nt][Nd]": None,
          "[Na][Nd][Nt]": idx.permute(0, 2, 1).reshape(-1).contiguous(),
          "[Nd][Na][Nt]": idx.permute(2, 0, 1).reshape(-1).contiguous(),
          "[Nt][Nd][Na]": idx.permute(1, 2, 0).reshape(-1).contiguous(),
          "morton": e.locality_order(o, d, bench.TMAX)}
tec = torch.empty(R, dtype=torch.float64, device="cuda")
ref = e.forward(o, d, bench.TMAX, bench.NS).clone()
for name, od in orders.items():
    for _ in range(5): e.forward(o, d, bench.TMAX, bench.NS, out=tec, order=od)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): e.forward(o, d, bench.TMAX, bench.NS, out=tec, order=od)
    torch.cuda.synchronize()
    print("%-22s %.4f ms  maxdiff %.1e" % (name, (time.perf_counter() - t0) / 50 * 1e3, float((tec - ref).abs().max())), flush=True)

"""Tricubic forward (LM fields) under different walks: memory order / Morton order, per-wave chunks / interleaved waves."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
R = w["origins"].shape[0]
res = {}
for walk in ("0", "1"):
    for bpc in ("0", "5", "6"):
        os.environ["IONOTOMO_WALK"] = walk
        os.environ["IONOTOMO_VARIANT"] = bpc
        e = RayEngine(0, interp="cubic")
        e.set_grid(w["xvec"], w["yvec"], w["zvec"])
        e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
        o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
        tec = torch.empty(R, dtype=torch.float64, device="cuda")
        order = e.locality_order(o, d, bench.TMAX)
        # [Na][Nd][Nt]: consecutive rays = consecutive timesteps of one (antenna, direction)
        perm = torch.arange(R, device="cuda").reshape(bench.NA, bench.NT, bench.ND).permute(0, 2, 1).reshape(-1).to(torch.int32).contiguous()
        for name, ordr in (("memory", None), ("morton", order), ("time_inner", perm)):
            for _ in range(2):
                e.forward(o, d, bench.TMAX, bench.NS, out=tec, order=ordr)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                e.forward(o, d, bench.TMAX, bench.NS, out=tec, order=ordr)
            b.record(); torch.cuda.synchronize()
            res["walk%s_bpc%s_%s" % (walk, bpc, name)] = round(a.elapsed_time(b) / 5, 3)
        del e
print(json.dumps(res, indent=1))

#!/usr/bin/env python3
"""Where a wave of k_forward_bundle spends its cycles: in-kernel s_memtime stamps of a -DIONO_B_STAMP build (timing-only).
    IONOTOMO_LIB=build_ab/libiono_stamp.so python profiles/tools/bundle_stamps.py"""
import ctypes, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ionotomo_amd import _lib  # noqa: E402
from ionotomo_amd.engine import RayEngine  # noqa: E402
w = bench.build_workload(0)
e = RayEngine(0)
e.set_grid(w["xvec"], w["yvec"], w["zvec"])
e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
nb = e.plan_forward(o, d, bench.TMAX, bench.NS)[0]
out = torch.empty(o.shape[0], dtype=torch.float64, device="cuda")
for _ in range(5):
    e.forward(o, d, bench.TMAX, bench.NS, out=out)
torch.cuda.synchronize()
lib = _lib.load()
n = min(nb, 8192) * 4 * 8
buf = np.zeros(n, dtype=np.uint64)
fn = lib.iono_debug_bundle_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
assert fn(e.ctx._h, buf.ctypes.data, n) == 0
s = buf.reshape(-1, 4, 8).astype(np.float64)
names = sys.argv[1:6] if len(sys.argv) > 5 else ["bundle prologue", "registers -> image (+ load wait)", "weights / next record wait", "next window load issue", "samples"]
life = s[:, :, 5]
res = {"workgroups": int(s.shape[0]), "wave_lifetime_cycles_mean": float(life.mean()),
       "kernel_span_cycles": float(s[:, :, 7].max() - s[:, :, 6].min()),
       "phase_cycles_mean_per_wave": {nm: float(s[:, :, i].mean()) for i, nm in enumerate(names)},
       "phase_fraction_of_lifetime": {nm: float(s[:, :, i].sum() / life.sum()) for i, nm in enumerate(names)},
       "chunks_per_wave": 8.25}
res["unaccounted_fraction"] = 1.0 - sum(res["phase_fraction_of_lifetime"].values())
print(json.dumps(res, indent=1))

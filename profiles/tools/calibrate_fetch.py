"""Known-byte-count launch of the forward kernel for calibrating the L2 fabric-side counters on THIS access pattern
(z-contiguous 16 B / lane gathers): vertical rays through a 512^3 f64 grid (1 GiB, beyond the 256 MiB Infinity Cache),
one ray per 2 x 2 block of columns, so every grid line under the fan is read exactly once from HBM per launch:
compulsory bytes = 510 x 510 columns x 512 nodes x 8 B.  Run under
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 profiles/tools/calibrate_fetch.py
and compare the counters of k_forward_straight_u<double> with `known_bytes`."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ionotomo_amd.engine import RayEngine
n = 512
xv = yv = np.linspace(0.0, n - 1.0, n)
zv = np.linspace(0.0, n - 1.0, n)
eng = RayEngine(0)
eng.set_grid(xv, yv, zv)
eng.set_values(torch.rand(n ** 3, dtype=torch.float64, device="cuda"))
a = np.arange(0, n - 2, 2) + 0.5
X, Y = np.meshgrid(a, a, indexing="ij")
o = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], 1)
d = np.tile(np.array([0.0, 0.0, 1.0]), (len(o), 1))
ot, dt = eng.tensor(o), eng.tensor(d)
flush = torch.empty(1 << 27, dtype=torch.float64, device="cuda")        # 1 GiB: evicts the Infinity Cache between launches
for _ in range(3):
    flush.fill_(1.0)
    tec = eng.forward(ot, dt, float(n - 1), n)
torch.cuda.synchronize()
assert not eng.check_oob()
print(json.dumps({"rays": len(o), "known_bytes": int(len(a) * 2) ** 2 * n * 8, "grid_bytes": n ** 3 * 8}))

import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ionotomo_amd.engine import RayEngine
from ionotomo_amd.ionosphere.covariance import Covariance
n = 256
eng = RayEngine(0)
eng.set_grid(np.linspace(-80, 80, n), np.linspace(-95, 95, n), np.linspace(-17, 1017, n))
C = Covariance(dx=160 / 255, dy=190 / 255, dz=1034 / 255)
phi = torch.randn(n, n, n, dtype=torch.float64, device="cuda")
out, work = torch.empty_like(phi), torch.empty_like(phi)
for _ in range(3):
    eng.smooth(phi, C.kx, C.ky, C.kz, out=out, work=work)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(10):
    eng.smooth(phi, C.kx, C.ky, C.kz, out=out, work=work)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print("smooth 256^3, stencil half width %d: %.3f ms  (compulsory 6 x 134 MB = 0.805 GB -> %.0f GB/s)" % (C.h, ms, 0.805 / ms * 1e3))

import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import oracle_c as OC
from ionotomo_amd.engine import RayEngine
rng = np.random.default_rng(2024)
n = 48
xv, yv, zv = np.linspace(-40, 40, n), np.linspace(-35, 45, n), np.linspace(-2, 210, n)
eng = RayEngine(0)
eng.set_grid(xv, yv, zv)
eng.set_values(eng.tensor(rng.uniform(1, 2, size=(n, n, n))))
for case in range(24):
    R = int(rng.choice([1, 3, 16, 17, 63, 64, 65, 130, 400, 1000]))
    Ns = int(rng.choice([9, 64, 65, 70, 129]))
    nant = int(rng.integers(1, 6))
    ants = np.stack([rng.uniform(-8, 8, nant), rng.uniform(-8, 8, nant), rng.uniform(0, 1.0, nant)], -1)
    spread = float(rng.choice([0.002, 0.02, 0.1]))
    a = rng.integers(0, nant, R)
    o = ants[a] + rng.normal(scale=0.01, size=(R, 3)) * [1, 1, 0]
    d = np.stack([rng.normal(scale=spread, size=R), rng.normal(scale=spread, size=R), np.ones(R)], -1)
    y = rng.normal(size=R)
    y[rng.random(R) < 0.2] = 0.0
    ref = OC.adjoint_straight(xv, yv, zv, o, d, y, 200.0, Ns)
    ot, dt, yt = eng.tensor(o), eng.tensor(d), eng.tensor(y)
    order = eng.locality_order(ot, dt, 200.0) if case % 2 else None
    g = eng.adjoint(ot, dt, yt, 200.0, Ns, order=order).cpu().numpy()
    err = np.max(np.abs(g - ref)) / max(np.max(np.abs(ref)), 1e-300)
    oob = eng.check_oob()
    end = o + d * ((200.0 - o[:, 2]) / d[:, 2])[:, None]
    print(case, R, Ns, spread, "err %.3e" % err, "oob", oob, "end x [%.1f, %.1f] y [%.1f, %.1f]" % (end[:, 0].min(), end[:, 0].max(), end[:, 1].min(), end[:, 1].max()), flush=True)

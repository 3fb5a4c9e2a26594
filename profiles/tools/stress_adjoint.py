"""One-off stress of the tiled adjoint (bundle ladder, partition, guided chunks) against the C oracle."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_c as OC
from ionotomo_amd.engine import RayEngine
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
worst = 0.0
for case in range(60):
    n = int(rng.choice([24, 40, 64]))
    xv, yv, zv = np.linspace(-60, 60, n), np.linspace(-55, 65, n + 3), np.linspace(-2, 210, n + 5)
    eng = RayEngine(0); eng.set_grid(xv, yv, zv); eng.set_values(eng.tensor(rng.uniform(1, 2, size=(n, n + 3, n + 5))))
    R = int(rng.choice([5, 64, 257, 1000, 5000, 20000]))
    Ns = int(rng.choice([8, 9, 33, 64, 65, 72, 73, 129, 200, 257]))
    nant = int(rng.integers(1, 12))
    ants = np.stack([rng.uniform(-15, 15, nant), rng.uniform(-15, 15, nant), rng.uniform(0, 1.0, nant)], -1)
    spread = float(rng.choice([0.001, 0.01, 0.03, 0.08]))
    a = rng.integers(0, nant, R)
    o = ants[a] + rng.normal(scale=0.02, size=(R, 3)) * [1, 1, 0]
    d = np.stack([np.clip(rng.normal(scale=spread, size=R), -0.15, 0.15), np.clip(rng.normal(scale=spread, size=R), -0.15, 0.15), np.ones(R)], -1)
    y = rng.normal(size=R); y[rng.random(R) < 0.1] = 0.0
    ref = OC.adjoint_straight(xv, yv, zv, o, d, y, 200.0, Ns)
    ot, dt, yt = eng.tensor(o), eng.tensor(d), eng.tensor(y)
    order = eng.locality_order(ot, dt, 200.0) if case % 3 else None
    scale = max(np.max(np.abs(ref)), 1e-300)
    g = eng.adjoint(ot, dt, yt, 200.0, Ns, order=order).cpu().numpy()
    e1 = np.max(np.abs(g - ref)) / scale
    st = eng.tune_adjoint_partition(lambda: eng.adjoint(ot, dt, yt, 200.0, Ns, order=order), R, fractions=tuple(rng.dirichlet(np.ones(int(rng.integers(1, 5))))))
    g2 = eng.adjoint(ot, dt, yt, 200.0, Ns, order=order).cpu().numpy()
    e2 = np.max(np.abs(g2 - ref)) / scale
    assert not eng.check_oob()
    worst = max(worst, e1, e2)
    assert e1 < 1e-11 and e2 < 1e-11, (case, n, R, Ns, nant, spread, e1, e2)
    print(case, n, R, Ns, nant, spread, "%.1e %.1e" % (e1, e2), None if st is None else st["chunks"], flush=True)
    del eng
print("worst", worst)

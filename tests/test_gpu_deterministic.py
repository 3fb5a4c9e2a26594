"""Deterministic (fixed-point) back-projection -- iono_set_deterministic / k_adjoint_binned<.., FIX>: the same bits run after run,
the float64 answer to the resolution the header states, every solver iterate reproducible; what the mode does not serve fails
loudly.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest
import torch

from ionotomo_amd import parallel, solvers

pytestmark = pytest.mark.gpu
SOAK = int(__import__("os").environ.get("IONO_SOAK", "1"))      # IONO_SOAK=20: twenty times the seeds


def engine(xv, yv, zv, **kw):
    from ionotomo_amd.engine import RayEngine
    eng = RayEngine(0, **kw)
    eng.set_grid(xv, yv, zv)
    return eng


def geometry(seed):
    rng = np.random.default_rng(seed)
    n = [int(v) for v in rng.integers(8, 70, 3)]
    xv, yv, zv = (np.linspace(0.0, float(rng.uniform(20, 200)), m) for m in n)
    R = int(rng.integers(50, 900))
    Ns = int(rng.choice([9, 17, 33, 65, 100, 257]))
    steep = float(rng.choice([0.02, 0.3, 1.5]))
    zlo, zhi = zv[0] + 0.1 * (zv[-1] - zv[0]), zv[-1] - 0.1 * (zv[-1] - zv[0])
    o = np.stack([rng.uniform(xv[0], xv[-1], R), rng.uniform(yv[0], yv[-1], R), np.full(R, zlo)], 1)
    d = np.stack([rng.normal(size=R) * steep, rng.normal(size=R) * steep, np.ones(R)], 1)
    w = rng.normal(size=R) * 10.0 ** rng.uniform(-6, 6)
    return n, xv, yv, zv, o, d, w, zhi, Ns


@pytest.mark.parametrize("seed", range(SOAK * 6))
def test_fixed_point_back_projection_is_reproducible_and_equals_the_float_one(seed):
    from oracle import oracle_c as OC
    n, xv, yv, zv, o, d, w, zhi, Ns = geometry(seed)
    eng = engine(xv, yv, zv)
    eng.set_values(eng.tensor(np.ones(n)))
    ot, dt, wt = eng.tensor(o), eng.tensor(d), eng.tensor(w)
    segments = eng.plan_adjoint(ot, dt, zhi, Ns)[0]
    gf = eng.adjoint(ot, dt, wt, zhi, Ns).cpu().numpy()
    oob = eng.check_oob()
    eng.set_deterministic(True)
    if segments == 0:
        # no ray of this seed is inside the grid (the reference raises on every one of them): there is nothing to plan, and the mode
        # says so instead of running the float-atomic kernels
        assert oob and float(np.abs(gf).max()) == 0.0
        with pytest.raises(ValueError, match="deterministic"):
            eng.adjoint(ot, dt, wt, zhi, Ns)
        return
    runs = [eng.adjoint(ot, dt, wt, zhi, Ns).clone() for _ in range(4)]
    assert eng.check_oob() == oob
    for g in runs[1:]:
        assert torch.equal(g, runs[0])                       # bit for bit
    g = runs[0].cpu().numpy()
    scale = np.max(np.abs(gf))
    assert np.max(np.abs(g - gf)) < 1e-10 * scale
    # accumulating into an existing float32 / float64 grid, twice: out += G^T w both times
    acc = torch.ones(tuple(n), dtype=torch.float64, device="cuda")
    eng.adjoint(ot, dt, wt, zhi, Ns, out=acc)
    eng.adjoint(ot, dt, wt, zhi, Ns, out=acc)
    assert np.max(np.abs(acc.cpu().numpy() - 1.0 - 2.0 * g)) < 1e-12 * max(scale, 1.0)
    g32 = eng.adjoint(ot, dt, wt, zhi, Ns, accum=torch.float32).cpu().numpy()
    assert np.max(np.abs(g32 - gf)) < 2e-6 * scale
    if not oob:
        gref = OC.adjoint_straight(xv, yv, zv, o, d, w, zhi, Ns)
        assert np.max(np.abs(g - gref)) < 1e-10 * np.max(np.abs(gref))
    eng.check_oob()
    # zero weights: nothing is added; a NaN weight: NaN where its ray goes, not a plausible number
    z = eng.adjoint(ot, dt, torch.zeros_like(wt), zhi, Ns)
    assert float(z.abs().max()) == 0.0
    wn = wt.clone()
    wn[::3] = float("nan")
    end = o + d * ((zhi - o[:, 2]) / d[:, 2])[:, None]
    inside = np.flatnonzero((end[:, 0] >= xv[0]) & (end[:, 0] <= xv[-1]) & (end[:, 1] >= yv[0]) & (end[:, 1] <= yv[-1]))
    if inside.size:                                         # (a steep seed can leave every third ray outside the grid: poison one inside)
        wn[int(inside[0])] = float("nan")
        assert bool(torch.isnan(eng.adjoint(ot, dt, wn, zhi, Ns)).any())
    else:
        eng.adjoint(ot, dt, wn, zhi, Ns)
    eng.check_oob()
    # ... and the integer grid is clean again afterwards
    assert torch.equal(eng.adjoint(ot, dt, wt, zhi, Ns), runs[0])
    eng.check_oob()


def test_what_the_mode_does_not_serve_fails_loudly():
    n, xv, yv, zv, o, d, w, zhi, Ns = geometry(3)
    eng = engine(xv, yv, zv)
    eng.set_values(eng.tensor(np.ones(n)))
    ot, dt, wt = eng.tensor(o), eng.tensor(d), eng.tensor(w)
    eng.set_deterministic(True)
    with pytest.raises(Exception, match="deterministic"):
        eng.adjoint(ot, dt, wt, zhi, Ns)                    # no plan for these rays
    with pytest.raises(Exception, match="deterministic"):
        eng.adjoint_fermat(ot, dt, wt, zhi, Ns, 150e6)      # the curved-ray transpose adds with float atomics
    eng.set_deterministic(False)
    eng.adjoint(ot, dt, wt, zhi, Ns)
    eng.check_oob()


def test_cgls_and_sirt_iterates_are_reproducible():
    """50 iterations twice: identical objective histories and iterates, bit for bit (with float atomics the late CG iterations differ
    from run to run by 1e-4 of the initial objective, tests/test_gpu_configs.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import oracle as Or, solvers as OS
    from problems import small_problem
    from test_gpu_engine import make_engine
    pb = small_problem(na=6, nd=6, nt=4, n=18, Ns=19)
    w = pb["w"]
    rays = Or.straight_rays(pb["o"], pb["d"], pb["tmax"], pb["Ns"])
    G, A = OS.dense_operator(rays, w["xvec"], w["yvec"], w["zvec"], pb["i0"])
    d = A @ pb["x_true"].ravel() + pb["rng"].normal(size=A.shape[0]) * 1e-3
    cd = np.full(A.shape[0], 1e-6)
    eng = make_engine(w)
    eng.set_deterministic(True)
    prob = parallel.ShardedRays(eng, pb["o"], pb["d"], pb["tmax"], pb["Ns"], dobs=d.reshape(pb["na"], pb["P"]),
                                cdct=cd.reshape(pb["na"], pb["P"]), i0=pb["i0"])
    x0 = eng.tensor(pb["x0"])
    for solve, ref in ((solvers.cgls, lambda: OS.cgls(A, d, cd, pb["x0"].ravel(), 50)),
                       (solvers.sirt, lambda: OS.sirt(G, A, d, cd, pb["x0"].ravel(), pb["na"], pb["P"], pb["i0"], 50))):
        x1, h1 = solve(prob, x0, n_iter=50)
        x2, h2 = solve(prob, x0, n_iter=50)
        assert torch.equal(x1, x2) and list(h1) == list(h2)
        xr, hr = ref()
        assert np.allclose(np.array(h1)[:12], np.array(hr)[:12], rtol=1e-6)
        assert np.max(np.abs(np.array(h1) - np.array(hr))) < 1e-3 * hr[0]


@pytest.mark.parametrize("seed", range(SOAK * 4))
def test_tricubic_transpose_in_deterministic_mode(seed):
    """The planned tricubic transpose with fixed-point channel images (k_adjoint_binned_lm4<.., FIX> + the z fold reading integers): the
    same bits twice, the float transpose's numbers, accumulation into an existing result, and a clean state afterwards."""
    rng = np.random.default_rng(40 + seed)
    n = [int(v) for v in rng.integers(14, 60, 3)]
    n[2] += n[2] & 1
    xv, yv, zv = (np.linspace(0.0, float(m - 1), m) for m in n)
    R, Ns = int(rng.integers(60, 500)), int(rng.choice([17, 33, 65, 129]))
    steep = float(rng.choice([0.02, 0.2]))
    o = np.stack([rng.uniform(4, n[0] - 5, R), rng.uniform(4, n[1] - 5, R), np.full(R, 2.3)], 1)
    d = np.stack([rng.normal(size=R) * steep, rng.normal(size=R) * steep, np.ones(R)], 1)
    tmax = zv[-1] - 4.6
    w = rng.normal(size=R) * 10.0 ** rng.uniform(-3, 3)
    eng = engine(xv, yv, zv, interp="cubic")
    eng.set_values(eng.tensor(np.ones(n)))
    ot, dt, wt = eng.tensor(o), eng.tensor(d), eng.tensor(w)
    eng.plan_adjoint(ot, dt, tmax, Ns)
    gf = eng.adjoint(ot, dt, wt, tmax, Ns)
    oob = eng.check_oob()
    eng.set_deterministic(True)
    g1 = eng.adjoint(ot, dt, wt, tmax, Ns)
    g2 = eng.adjoint(ot, dt, wt, tmax, Ns)
    assert eng.check_oob() == oob
    assert torch.equal(g1, g2)
    scale = float(gf.abs().max())
    assert scale > 0 and float((g1 - gf).abs().max()) < 1e-10 * scale
    acc = torch.full(tuple(n), 2.0, dtype=torch.float64, device="cuda")
    eng.adjoint(ot, dt, wt, tmax, Ns, out=acc)
    assert float((acc - 2.0 - g1).abs().max()) < 1e-12 * max(scale, 1.0)
    eng.set_deterministic(False)
    g3 = eng.adjoint(ot, dt, wt, tmax, Ns)                      # the float path on the same scratch buffers
    assert float((g3 - gf).abs().max()) < 1e-11 * scale
    eng.check_oob()

"""Tricubic on the hot path (round 2): the Lekien-Marsden derivative-field forward and the channel-scatter + fold
transpose (ionotomo_amd/csrc/iono_cubic_kernels.h) against the oracle's 6 x 6 x 6 tensor-product form
(oracle.tricubic: notebooks/TricubicInterpolation.ipynb c0:138-1257), through the C-ABI.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest
import torch

from ionotomo_amd import _lib, parallel, synthetic as syn

pytestmark = pytest.mark.gpu
SOAK = int(__import__("os").environ.get("IONO_SOAK", "1"))      # IONO_SOAK=20: twenty times the seeds (a soak run on the GPU box)


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def rel(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


def engine(w, **kw):
    from ionotomo_amd.engine import RayEngine
    eng = RayEngine(0, interp="cubic", **kw)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    return eng


@pytest.fixture(scope="module")
def cfg2():
    # config 2: 62 LOFAR stations x 42 directions x 1 time, 128^3, TriCubic interpolation
    return syn.make_workload("cfg2")


def test_config2_tricubic_forward_all_rays(cfg2, O, monkeypatch):
    w = cfg2
    M = w["ne"] / 1e13
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    eng = engine(w)
    eng.set_values(eng.tensor(M))
    ot, dt = eng.tensor(o), eng.tensor(d)
    for Ns, rule in ((129, "avg"), (128, "avg"), (128, "scipy"), (200, "avg"), (70, "trapz")):
        eng.rule = _lib.quad_rule(rule)
        tec = eng.forward(ot, dt, w["tmax"], Ns).cpu().numpy()
        assert not eng.check_oob()
        rays = O.straight_rays(o, d, w["tmax"], Ns)
        ref = O.forward_tec(rays, w["xvec"], w["yvec"], w["zvec"], M, rule=_lib.quad_rule(rule), kind=O.INTERP_TRICUBIC)
        assert rel(tec, ref) < 1e-11, (Ns, rule)
    # the general tier (216 taps per sample, any grid) agrees with the fast one
    monkeypatch.setenv("IONOTOMO_VARIANT", "4")
    gen = engine(w)
    gen.set_values(gen.tensor(M))
    eng.rule = gen.rule = _lib.quad_rule("avg")
    a = eng.forward(ot, dt, w["tmax"], 129).cpu().numpy()
    b = gen.forward(gen.tensor(o), gen.tensor(d), w["tmax"], 129).cpu().numpy()
    assert rel(a, b) < 1e-12
    # values change -> the derivative fields are rebuilt
    eng.set_values(eng.tensor(2.0 * M))
    assert rel(eng.forward(ot, dt, w["tmax"], 129).cpu().numpy(), 2.0 * a) < 1e-13
    # float32 storage of the node values (the fields themselves stay float64)
    e32 = engine(w, storage="f32")
    e32.set_values(e32.tensor(M))
    c = e32.forward(e32.tensor(o), e32.tensor(d), w["tmax"], 129).cpu().numpy()
    assert rel(c, a) < 2e-6
    # rays that leave the tricubic domain g[2] .. g[n-3] set the flag (scipy-style bounds error in the facade)
    eng.forward(ot, dt, w["zvec"][-2], 129)
    assert eng.check_oob() and not eng.check_oob()


def test_lm_fields_reproduce_a_cubic_polynomial_exactly(O):
    """The interpolant is exact for tri-quadratic fields (cubic Hermite with 4th-order slopes differentiates
    polynomials up to degree 4 exactly and reproduces cubics): TEC = the analytic line integral, Simpson exact."""
    xv, yv, zv = np.linspace(-30, 30, 41), np.linspace(-25, 35, 37), np.linspace(0, 100, 51)
    X, Y, Z = np.meshgrid(xv, yv, zv, indexing="ij")
    M = 1.0 + 0.01 * X - 0.02 * Y + 0.003 * Z + 1e-3 * X * Y - 2e-4 * Y * Z + 3e-4 * X * Z + 1e-5 * X * Y * Z
    w = dict(xvec=xv, yvec=yv, zvec=zv)
    eng = engine(w)
    eng.set_values(eng.tensor(M))
    rng = np.random.default_rng(0)
    R = 300
    o = np.stack([rng.uniform(-10, 10, R), rng.uniform(-10, 10, R), np.full(R, zv[2])], 1)
    d = np.stack([rng.uniform(-0.1, 0.1, R), rng.uniform(-0.1, 0.1, R), np.ones(R)], 1)
    tmax, Ns = zv[-3], 97
    tec = eng.forward(eng.tensor(o), eng.tensor(d), tmax, Ns).cpu().numpy()
    # the field restricted to a straight ray is a cubic in the path parameter: Simpson integrates it exactly
    rays = O.straight_rays(o, d, tmax, Ns)
    f = lambda x, y, z: 1.0 + 0.01 * x - 0.02 * y + 0.003 * z + 1e-3 * x * y - 2e-4 * y * z + 3e-4 * x * z + 1e-5 * x * y * z
    vals = f(rays[:, 0], rays[:, 1], rays[:, 2])
    ref = O.simps(vals, rays[:, 3])
    assert rel(tec, ref) < 1e-12


def test_tricubic_adjoint_general_tier_small(O):
    """Explicit-sample and straight-ray transposes on a NON-uniform grid (216 atomics per sample) vs the oracle."""
    rng = np.random.default_rng(1)
    xv = np.cumsum(rng.uniform(0.5, 1.5, 16))
    yv = np.cumsum(rng.uniform(0.5, 1.5, 15))
    zv = np.cumsum(rng.uniform(0.5, 1.5, 18))
    M = rng.normal(size=(16, 15, 18))
    R = 40
    o = np.stack([rng.uniform(xv[4], xv[10], R), rng.uniform(yv[4], yv[9], R), np.full(R, zv[2])], 1)
    d = np.stack([rng.uniform(-0.05, 0.05, R), rng.uniform(-0.05, 0.05, R), np.ones(R)], 1)
    c = _lib.Context(0)
    c.set_grid(xv, yv, zv, M)
    y = rng.normal(size=R)
    for Ns in (9, 12):
        rays = O.straight_rays(o, d, zv[-3], Ns)
        ref = O.adjoint_tec(rays, xv, yv, zv, y, kind=O.INTERP_TRICUBIC)
        g1 = c.adjoint_rays(rays, y, kind="cubic")
        g2 = c.adjoint_straight(o, d, y, zv[-3], Ns, kind="cubic")
        assert rel(g1, ref) < 1e-11 and rel(g2, ref) < 1e-11
        t = c.forward_tec_rays(rays, kind="cubic")
        assert abs(np.dot(t.ravel(), y) - np.sum(g1 * M)) < 1e-11 * np.linalg.norm(t) * np.linalg.norm(y)
    gs = c.adjoint_straight(o, d, y, zv[-3], 9, kind="cubic", scale_by_grid=True)
    assert rel(gs, O.adjoint_tec(O.straight_rays(o, d, zv[-3], 9), xv, yv, zv, y, kind=O.INTERP_TRICUBIC) * M) < 1e-11
    c.close()


def test_config2_tricubic_adjoint_fast_tier(cfg2, O, monkeypatch):
    """Channel scatter (LDS-tiled) + fold vs the oracle's 216-tap transpose, a 400-ray sample of config 2 at 128^3;
    with / without walk order; fused-residual and differential-weight modes; float32 accumulation; general tier."""
    w = cfg2
    M = w["ne"] / 1e13
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    idx = np.sort(np.random.default_rng(2).choice(len(o), 400, replace=False))
    Ns = 129
    rays = O.straight_rays(o[idx], d[idx], w["tmax"], Ns)
    y = np.random.default_rng(3).normal(size=len(idx))
    ref = O.adjoint_tec(rays, w["xvec"], w["yvec"], w["zvec"], y, kind=O.INTERP_TRICUBIC)
    eng = engine(w)
    eng.set_values(eng.tensor(M))
    ot, dt, yt = eng.tensor(o[idx]), eng.tensor(d[idx]), eng.tensor(y)
    order = eng.locality_order(ot, dt, w["tmax"])
    for ordr in (None, order):
        g = eng.adjoint(ot, dt, yt, w["tmax"], Ns, order=ordr).cpu().numpy()
        assert rel(g, ref) < 1e-11
    g32 = eng.adjoint(ot, dt, yt, w["tmax"], Ns, order=order, accum=torch.float32).cpu().numpy()
    assert rel(g32, ref) < 1e-6
    monkeypatch.setenv("IONOTOMO_VARIANT", "4")
    gen = engine(w)
    gen.set_values(gen.tensor(M))
    gg = gen.adjoint(gen.tensor(o[idx]), gen.tensor(d[idx]), gen.tensor(y), w["tmax"], Ns).cpu().numpy()
    assert rel(gg, ref) < 1e-11
    monkeypatch.delenv("IONOTOMO_VARIANT")
    # all 2,604 rays in layout [Na][P]: fused residual and differential modes against the separate steps
    na, P = 62, 42
    rng = np.random.default_rng(4)
    dobs, cdct = rng.normal(size=(na, P)) * 0.1, rng.uniform(0.5, 2.0, size=(na, P))
    O4, D4 = w["origins"].reshape(na, P, 3), w["directions"].reshape(na, P, 3)
    for i0 in (0, 7):
        prob = parallel.ShardedRays(eng, O4, D4, w["tmax"], Ns, dobs=dobs, cdct=cdct, i0=i0, tune=False)
        tec = prob.forward_tec()
        t = tec.cpu().numpy().reshape(na, P)
        dd = (t - t[i0] - dobs) / (cdct + 1e-15)
        wd = O.differential_weights(dd, i0)
        sep = eng.adjoint(prob.origins, prob.dirs, eng.tensor(wd.ravel()), w["tmax"], Ns, order=prob.order).cpu().numpy()
        fused = prob.gradient_from_tec(tec).cpu().numpy()
        assert rel(fused, sep) < 1e-11
        v, sc = rng.normal(size=(na, P)), rng.uniform(0.5, 2.0, size=(na, P))
        wd2 = O.differential_weights(v * sc, i0)
        sep2 = eng.adjoint(prob.origins, prob.dirs, eng.tensor(wd2.ravel()), w["tmax"], Ns, order=prob.order).cpu().numpy()
        dif = eng.adjoint_differential(prob.origins, prob.dirs, eng.tensor(v.ravel()), eng.tensor(sc.ravel()), na, i0, w["tmax"],
                                       Ns, order=prob.order).cpu().numpy()
        assert rel(dif, sep2) < 1e-11
    assert not eng.check_oob()


def test_full_batch_tricubic_dot_product_256_cubed():
    """BASELINE size: 260,400 rays through 256^3 with the tricubic -- <G x, y> = <x, G^T y> and linearity."""
    import bench
    w = bench.build_workload(0)
    eng = engine(w)
    rng = np.random.default_rng(5)
    x = np.exp(w["m"])
    eng.set_values(eng.tensor(x))
    ot, dt = eng.tensor(w["origins"]), eng.tensor(w["directions"])
    tec = eng.forward(ot, dt, bench.TMAX, bench.NS)
    assert not eng.check_oob()
    yt = eng.tensor(rng.normal(size=ot.shape[0]))
    order = eng.locality_order(ot, dt, bench.TMAX)
    g = eng.adjoint(ot, dt, yt, bench.TMAX, bench.NS, order=order)
    lhs, rhs = float(torch.dot(tec, yt)), float((g * eng.tensor(x)).sum())
    assert abs(lhs - rhs) < 1e-10 * float(tec.norm()) * float(yt.norm())
    # trilinear and tricubic TEC of a smooth field agree to the interpolation error, not to rounding
    lin = engine(w)
    lin.kind = _lib.interp_kind("linear")
    lin.set_values(lin.tensor(x))
    tl = lin.forward(ot, dt, bench.TMAX, bench.NS)
    dev = float(((tl - tec).abs() / tec.abs()).max())
    assert 1e-9 < dev < 5e-2
    assert not eng.check_oob()


def test_solvers_with_the_tricubic_operator():
    """CGLS / SIRT on the tricubic forward + its transpose: the fused solver passes (search direction read in place, so the
    derivative fields are rebuilt every iteration; plan-binned channel transposes) give the iterates of the dense-vector
    form on the same engine, and the objective falls."""
    from ionotomo_amd import solvers
    from problems import small_problem
    pb = small_problem(na=6, nd=5, nt=3, n=24, Ns=33)
    w = pb["w"]
    eng = engine(w)
    rng = np.random.default_rng(7)
    prob = parallel.ShardedRays(eng, pb["o"], pb["d"], pb["tmax"], pb["Ns"], dobs=np.zeros((pb["na"], pb["P"])),
                                cdct=np.full((pb["na"], pb["P"]), 1e-6), i0=pb["i0"], tune=False)
    assert prob.plan and prob.plan[0] > 0
    eng.set_values(eng.tensor(pb["x_true"]))
    prob.dobs = prob.forward() + eng.tensor(rng.normal(size=pb["na"] * pb["P"]) * 1e-3)
    x0 = eng.tensor(pb["x0"])
    for name in ("cgls", "sirt"):
        xf, hf = getattr(solvers, name)(prob, x0, n_iter=6)
        xd, hd = getattr(solvers, "_%s_dense" % name)(prob, x0, n_iter=6)
        assert np.allclose(hf, hd, rtol=1e-7) and hf[-1] < 0.5 * hf[0]
        assert float((xf - xd).abs().max()) < 1e-7 * float(xd.abs().max())
    assert not eng.check_oob()


@pytest.mark.parametrize("seed", range(SOAK * 8))
def test_tricubic_fast_forward_random_geometry(seed, O, monkeypatch):
    """The wave kernel shares node records between neighbouring lanes (a lane's upper node = the next lane's lower node when
    the next sample sits one cell higher in the same column): every way that assumption can fail must fall back to the
    lane's own loads -- rays so oblique that the column changes at every sample, more or less than one cell per sample, two
    samples in one cell, Ns below / off a multiple of 64 (masked tail), one ray, walk orders and modes."""
    rng = np.random.default_rng(500 + seed)
    nx, ny, nz = (int(v) for v in rng.integers(9, 48, 3))
    xv, yv, zv = np.linspace(-40.0, 35.0, nx), np.linspace(-28.0, 44.0, ny), np.linspace(-6.0, 110.0, nz)
    M = rng.uniform(0.5, 2.0, size=(nx, ny, nz))
    R = int(rng.choice([1, 5, 37, 200]))
    Ns = int(rng.choice([3, 17, 63, 64, 65, 127, 130, 257, 300]))
    slope = [0.02, 0.3, 1.5, 4.0][seed % 4]              # lateral cells per vertical cell: up to 4 columns per sample
    lo, hi = np.array([xv[2], yv[2], zv[2]]), np.array([xv[-3], yv[-3], zv[-3]])       # the tricubic domain
    z0 = rng.uniform(lo[2], lo[2] + 5.0, R)
    z1 = hi[2] - rng.uniform(1e-6, 20.0)
    # choose both end points inside the domain, then the direction between them
    a = np.stack([rng.uniform(lo[0], hi[0], R), rng.uniform(lo[1], hi[1], R), z0], -1)
    reach = slope * (z1 - z0)[:, None] * np.array([(xv[1] - xv[0]) / (zv[1] - zv[0]), (yv[1] - yv[0]) / (zv[1] - zv[0])])
    b_xy = np.clip(a[:, :2] + rng.uniform(-1, 1, (R, 2)) * reach, lo[:2] + 1e-6, hi[:2] - 1e-6)     # (end point recomputed on the device)
    d = np.concatenate([b_xy - a[:, :2], (z1 - z0)[:, None]], -1)
    d *= rng.uniform(0.5, 2.0, (R, 1))                                                # not normalised on purpose
    from ionotomo_amd.engine import RayEngine
    eng = RayEngine(0, interp="cubic")
    eng.set_grid(xv, yv, zv)
    eng.set_values(eng.tensor(M))
    ot, dt = eng.tensor(a), eng.tensor(d)
    rays = O.straight_rays(a, d, z1, Ns)
    ref = O.forward_tec(rays, xv, yv, zv, M, rule=_lib.quad_rule("avg"), kind=O.INTERP_TRICUBIC)
    order = torch.from_numpy(rng.permutation(R).astype(np.int32)).cuda()
    for ordr in (None, order):
        tec = eng.forward(ot, dt, z1, Ns, order=ordr).cpu().numpy()
        assert not eng.check_oob()
        assert np.max(np.abs(tec - ref) / np.abs(ref)) < 1e-11, (seed, Ns, R, slope)


@pytest.mark.parametrize("seed", range(SOAK * 6))
def test_planned_tricubic_transpose_folds_only_the_tiles_the_rays_reach(seed, monkeypatch):
    """The planned tricubic transpose zeroes, scatters into and folds only the 8 x 8 x 16-node tiles its rays reach (k_lm_*_tiles).
    On grids whose sizes are no multiples of the tile, with pencils of rays that leave most of the grid untouched, it must equal the
    unplanned transpose (whose folds run over the whole grid), also when the scratch buffers hold an EARLIER launch's
    values outside the new plan's tiles, and leave the rest of an accumulated result alone."""
    from ionotomo_amd.engine import RayEngine
    rng = np.random.default_rng(100 + seed)
    n = [int(v) for v in rng.integers(12, 75, 3)]
    if seed % 2:
        n[2] |= 1                                                  # (any parity of nz: the forward / transpose plans take both)
    xv, yv, zv = (np.linspace(0.0, float(m - 1), m) for m in n)
    Ns = int(rng.choice([17, 33, 65, 129]))

    def pencil(R, centre, steep):
        o = np.stack([centre[0] + rng.normal(size=R) * 0.7, centre[1] + rng.normal(size=R) * 0.7, np.full(R, zv[0] + 2.3)], 1)
        d = np.stack([rng.normal(size=R) * steep, rng.normal(size=R) * steep, np.ones(R)], 1)
        return o, d
    tmax = zv[-1] - 2.3 - (zv[0] + 2.3)
    cx = [float(rng.uniform(4, m - 5)) for m in n[:2]]
    o1, d1 = pencil(int(rng.integers(40, 400)), cx, 0.02)
    o2, d2 = pencil(int(rng.integers(40, 400)), [float(rng.uniform(4, m - 5)) for m in n[:2]], 0.05)

    def run(o, d, y, variant, planned, eng=None, base=None):
        if variant:
            monkeypatch.setenv("IONOTOMO_VARIANT", variant)
        else:
            monkeypatch.delenv("IONOTOMO_VARIANT", raising=False)
        if eng is None:
            eng = RayEngine(0, interp="cubic")
            eng.set_grid(xv, yv, zv)
            eng.set_values(eng.tensor(np.ones(n)))
        ot, dt = eng.tensor(o), eng.tensor(d)
        if planned:
            eng.plan_adjoint(ot, dt, tmax, Ns)
        else:
            eng.clear_adjoint_plan()
        out = None if base is None else base.clone()
        g = eng.adjoint(ot, dt, eng.tensor(y), tmax, Ns, out=out)
        oob = eng.check_oob()
        return g, eng, oob
    y1, y2 = rng.normal(size=len(o1)), rng.normal(size=len(o2))
    g_tiles, eng, oob1 = run(o1, d1, y1, None, True)
    g_free, _, _ = run(o1, d1, y1, None, False)
    scale = float(g_free.abs().max())
    assert scale > 0
    assert float((g_tiles - g_free).abs().max()) < 1e-11 * scale
    # the same engine, another pencil somewhere else: the first launch's channel values are still in the scratch buffers
    base = torch.full(tuple(n), 3.0, dtype=torch.float64, device="cuda")
    g2, _, _ = run(o2, d2, y2, None, True, eng=eng, base=base)
    g2_free, _, _ = run(o2, d2, y2, None, False)
    assert float((g2 - 3.0 - g2_free).abs().max()) < 1e-11 * max(float(g2_free.abs().max()), 1.0)


def test_planned_forward_rebuilds_only_the_fields_its_rays_read(monkeypatch):
    """Round 5: with new node values the planned tricubic forward rebuilds its derivative fields only on the node lines the plan's
    windows hold (k_lm_touch_lines -> k_lm_fields_yx with an x range per line), not over the whole grid.  On a grid most of which
    the rays never reach: every new set of values gives the unplanned kernel's numbers (whose own field array is always rebuilt in
    full); a NEW plan for other rays on the same engine sees fields for ITS lines; rays edited in place after the restricted rebuild
    are recomputed straight from the node values (exact) + the stale flag, never from fields that were not rebuilt."""
    from ionotomo_amd.engine import RayEngine
    rng = np.random.default_rng(42)
    n = (52, 61, 47)
    xv, yv, zv = (np.linspace(0.0, float(m - 1), m) for m in n)
    tmax, Ns = 40.0, 97

    def pencil(R, cx, cy, steep):
        o = np.stack([cx + rng.normal(size=R) * 0.8, cy + rng.normal(size=R) * 0.8, np.full(R, 2.5)], 1)
        d = np.stack([rng.normal(size=R) * steep, rng.normal(size=R) * steep, np.ones(R)], 1)
        return o, d
    o1, d1 = pencil(900, 14.0, 40.0, 0.03)
    o2, d2 = pencil(700, 37.0, 17.0, 0.05)

    def make(variant=None):
        monkeypatch.setenv("IONOTOMO_HYBRID_MIN", "1")                                     # the bundle kernel whatever the plan's size
        if variant is not None:
            monkeypatch.setenv("IONOTOMO_VARIANT", variant)
        e = RayEngine(0, interp="cubic")
        monkeypatch.delenv("IONOTOMO_HYBRID_MIN")
        monkeypatch.delenv("IONOTOMO_VARIANT", raising=False)
        e.set_grid(xv, yv, zv)
        return e
    eng = make()
    free = RayEngine(0, interp="cubic")
    free.set_grid(xv, yv, zv)
    ot, dt = eng.tensor(o1), eng.tensor(d1)
    nb, _, fit = eng.plan_forward(ot, dt, tmax, Ns)
    assert nb > 0 and fit == 1.0
    for it in range(3):
        M = eng.tensor(rng.uniform(1.0, 2.0, size=n))
        eng.set_values(M), free.set_values(M)
        a = eng.forward(ot, dt, tmax, Ns)
        b = free.forward(ot, dt, tmax, Ns)
        assert not eng.check_oob() and not eng.plan_stale()
        assert float((a - b).abs().max()) < 1e-12 * float(b.abs().max()), it
    # another plan on the same engine, same values: its own lines are rebuilt (the first plan's are not enough)
    ot2, dt2 = eng.tensor(o2), eng.tensor(d2)
    assert eng.plan_forward(ot2, dt2, tmax, Ns)[0] > 0
    a2 = eng.forward(ot2, dt2, tmax, Ns)
    b2 = free.forward(ot2, dt2, tmax, Ns)
    assert float((a2 - b2).abs().max()) < 1e-12 * float(b2.abs().max())
    # ... and back: plan 1 again after plan 2's restricted rebuild
    eng.plan_forward(ot, dt, tmax, Ns)
    a = eng.forward(ot, dt, tmax, Ns)
    assert float((a - b).abs().max()) < 1e-12 * float(b.abs().max())
    # rays edited in place (one far outside the planned pencil): exact from the node values + the stale flag
    keep = ot.clone()
    ot[5, 0] += 20.0
    ot[77, 1] -= 15.0
    stale = eng.forward(ot, dt, tmax, Ns)
    assert eng.plan_stale()
    want = free.forward(ot.clone(), dt.clone(), tmax, Ns)
    assert not free.check_oob() and bool(torch.isfinite(stale).all())
    assert float((stale - want).abs().max()) < 1e-12 * float(want.abs().max())
    assert float((stale[5] - b[5]).abs()) > 1e-6 * float(b.abs().max())                 # (the edit did change that ray)
    ot.copy_(keep)
    again = eng.forward(ot, dt, tmax, Ns)
    assert not eng.plan_stale() and float((again - b).abs().max()) < 1e-12 * float(b.abs().max())



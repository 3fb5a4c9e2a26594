"""Bundle-stationary forward -- iono_forward_plan_dev / k_forward_bundle (the voxel neighbourhood of <= 64 nearly coincident rays
staged in LDS) -- against the C oracle and the direct-load kernels: random geometries incl. steep rays whose windows do not
fit the LDS image, odd nz (columns that start on 8-byte boundaries only), rays that leave the grid, every quadrature rule, sample counts below one chunk,
rays on the grid faces, the bench shape, and the two properties the design promises: TEC does not depend on how the rays were
bundled (bit for bit), and a stale plan is never used.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest
import torch

from ionotomo_amd import synthetic as syn

pytestmark = pytest.mark.gpu
SOAK = int(__import__("os").environ.get("IONO_SOAK", "1"))


@pytest.fixture(scope="module")
def OC():
    from oracle import oracle_c
    return oracle_c


def engine(xv, yv, zv, force_bundle=True, env=None, **kw):
    """``force_bundle``: IONOTOMO_HYBRID_MIN=1 while the context is created -- a planned launch then runs k_forward_bundle for EVERY
    bundle whatever their number and size (by default the plan decides which bundles are worth a workgroup -- all, those of >= T
    rays with the others' rays going to the lanes = samples kernel in the same call, or none: iono_forward_plan_split).
    ``env``: further variables read at context creation."""
    import os
    from ionotomo_amd.engine import RayEngine
    env = dict(env or {})
    if force_bundle:
        env.setdefault("IONOTOMO_HYBRID_MIN", "1")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        eng = RayEngine(0, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    eng.set_grid(xv, yv, zv)
    return eng


def random_rays(rng, xv, yv, zv, R, steep, cluster):
    """R rays from the z = zlo plane; ``cluster``: feet and slopes drawn around a few centres (compact bundles) or uniformly."""
    zlo, zhi = zv[0] + rng.uniform(0, 0.3) * (zv[-1] - zv[0]), zv[-1] - rng.uniform(0, 0.2) * (zv[-1] - zv[0])
    if cluster:
        nc = int(rng.integers(1, 6))
        cx, cy = rng.uniform(xv[0], xv[-1], nc), rng.uniform(yv[0], yv[-1], nc)
        sx, sy = rng.normal(size=nc) * steep, rng.normal(size=nc) * steep
        pick = rng.integers(0, nc, R)
        hx, hy = xv[1] - xv[0], yv[1] - yv[0]
        o = np.stack([cx[pick] + rng.normal(size=R) * 2 * hx, cy[pick] + rng.normal(size=R) * 2 * hy, np.full(R, zlo)], 1)
        d = np.stack([sx[pick] + rng.normal(size=R) * 0.01, sy[pick] + rng.normal(size=R) * 0.01, np.ones(R)], 1)
    else:
        o = np.stack([rng.uniform(xv[0], xv[-1], R), rng.uniform(yv[0], yv[-1], R), np.full(R, zlo)], 1)
        d = np.stack([rng.normal(size=R) * steep, rng.normal(size=R) * steep, np.ones(R)], 1)
    o[:, 2] += rng.uniform(0, 0.4, R) * (zv[1] - zv[0])                    # antennas at slightly different heights
    end = o + d * ((zhi - o[:, 2]) / d[:, 2])[:, None]
    inside = ((o[:, 0] >= xv[0]) & (o[:, 0] <= xv[-1]) & (o[:, 1] >= yv[0]) & (o[:, 1] <= yv[-1]) &
              (end[:, 0] >= xv[0]) & (end[:, 0] <= xv[-1]) & (end[:, 1] >= yv[0]) & (end[:, 1] <= yv[-1]))
    return o, d, zhi, inside


@pytest.mark.parametrize("seed", range(SOAK * 12))
def test_bundle_forward_random_geometries(seed, OC):
    rng = np.random.default_rng(1000 + seed)
    n = [int(v) for v in rng.integers(6, 70, 3)]
    if seed % 2 == 0:
        n[2] |= 1                                                         # odd nz: columns start on 8-byte boundaries only
    elif seed % 4 == 1:
        n[2] += n[2] & 1                                                  # even nz; every fourth case keeps any nz
    xv, yv, zv = (np.linspace(0.0, float(rng.uniform(20, 200)), m) for m in n)
    R = int(rng.integers(1, 1500))
    Ns = int(rng.choice([2, 3, 7, 8, 9, 16, 17, 33, 64, 65, 100, 257]))
    steep = float(rng.choice([0.02, 0.3, 1.5]))
    o, d, zhi, inside = random_rays(rng, xv, yv, zv, R, steep, cluster=bool(seed % 2))
    quad = ["avg", "scipy", "trapz"][seed % 3]
    eng = engine(xv, yv, zv, quad=quad)
    M = rng.uniform(1, 2, size=n)
    eng.set_values(eng.tensor(M))
    ot, dt = eng.tensor(o), eng.tensor(d)
    direct = eng.forward(ot, dt, zhi, Ns).cpu().numpy()                    # no plan: lanes = samples
    assert eng.check_oob() == (not inside.all())
    nb, nchunks, fit = eng.plan_forward(ot, dt, zhi, Ns)
    # (odd nz too since round 5: the window rows are staged with 16-byte loads from 8-byte aligned addresses)
    assert nb >= (R + 63) // 64 and nchunks == (Ns + 7) // 8 and 0.0 <= fit <= 1.0
    tec = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    assert eng.check_oob() == (not inside.all())                           # rays leaving the grid: NaN + flag
    assert np.all(np.isnan(tec[~inside])) and np.all(np.isfinite(tec[inside]))
    if inside.any():
        assert np.max(np.abs(tec[inside] - direct[inside])) <= 2e-13 * np.max(np.abs(direct[inside])), (n, R, Ns, steep, fit)
        if quad == "avg" or (quad == "scipy" and Ns % 2 == 1):
            ref = OC.forward_tec_straight(xv, yv, zv, M, o[inside], d[inside], zhi, Ns)
            assert np.max(np.abs(tec[inside] - ref)) <= 1e-12 * np.max(np.abs(ref)), (n, R, Ns, steep, fit)
    eng.clear_forward_plan()
    again = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    eng.check_oob()
    assert np.array_equal(again, direct, equal_nan=True)
    # without the override the plan itself decides which bundles are worth a workgroup (iono_forward_plan_split): at these sizes none
    # -- one lanes = samples launch is modelled (and measured) faster than any bundle launch -- so the plan only orders the walk and
    # the result carries the lanes = samples kernel's bits
    eng2 = engine(xv, yv, zv, force_bundle=False, quad=quad)
    eng2.set_values(eng2.tensor(M))
    nb2, _, fit2 = eng2.plan_forward(ot, dt, zhi, Ns)
    sp2 = eng2.forward_plan_split()
    assert sp2["bundles_served"] == nb2 and sp2["rays_served"] + sp2["rays_tail"] == R and sp2["bundles_cut"] == nb
    small = eng2.forward(ot, dt, zhi, Ns).cpu().numpy()
    eng2.check_oob()
    if nb2 == 0 or fit2 < 0.5:
        assert sp2["min_rays_per_served_bundle"] == 65 or fit2 < 0.5
        assert np.array_equal(small, direct, equal_nan=True)
    elif inside.any():
        assert np.max(np.abs(small[inside] - direct[inside])) <= 2e-13 * np.max(np.abs(direct[inside]))


def mixed_rays(rng, xv, yv, zv, n_dense, n_sparse, cubic):
    """Dense clusters (their bundles fill) + scattered rays (bundles of one or two): what a few timesteps of a sparse array give."""
    lo = 2 if cubic else 0
    x0, x1, y0, y1 = xv[lo], xv[-1 - lo], yv[lo], yv[-1 - lo]
    zlo, zhi = zv[lo] + 1e-9 + 0.1 * (zv[-1] - zv[0]), zv[-1 - lo] - 1e-9 - 0.1 * (zv[-1] - zv[0])
    hx, hy = xv[1] - xv[0], yv[1] - yv[0]
    nc = int(rng.integers(1, 5))
    cx, cy = rng.uniform(x0 + 0.3 * (x1 - x0), x0 + 0.7 * (x1 - x0), nc), rng.uniform(y0 + 0.3 * (y1 - y0), y0 + 0.7 * (y1 - y0), nc)
    pick = rng.integers(0, nc, n_dense)
    o1 = np.stack([cx[pick] + rng.normal(size=n_dense) * hx, cy[pick] + rng.normal(size=n_dense) * hy, np.full(n_dense, zlo)], 1)
    d1 = np.stack([rng.normal(size=n_dense) * 0.004, rng.normal(size=n_dense) * 0.004, np.ones(n_dense)], 1)
    o2 = np.stack([rng.uniform(x0, x1, n_sparse), rng.uniform(y0, y1, n_sparse), np.full(n_sparse, zlo)], 1)
    d2 = np.stack([rng.normal(size=n_sparse) * 0.2, rng.normal(size=n_sparse) * 0.2, np.ones(n_sparse)], 1)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    p = rng.permutation(len(o))
    o, d = o[p], d[p]
    end = o + d * ((zhi - o[:, 2]) / d[:, 2])[:, None]
    inside = ((o[:, 0] >= x0) & (o[:, 0] <= x1) & (o[:, 1] >= y0) & (o[:, 1] <= y1) &
              (end[:, 0] >= x0) & (end[:, 0] <= x1) & (end[:, 1] >= y0) & (end[:, 1] <= y1))
    return o, d, zhi, inside


@pytest.mark.parametrize("interp", ["linear", "cubic"])
@pytest.mark.parametrize("seed", range(SOAK * 6))
def test_hybrid_dispatch_mixed_geometries(seed, interp, OC):
    """the choice between the bundle kernels and the lanes = samples kernels is made per BUNDLE.  On geometries that
    mix well-filled bundles with singletons the served bundles go to k_forward_bundle / k_forward_bundle_lm, the other rays to
    k_forward_straight_u / _lm in the SAME call: equal to the unplanned launch to 2e-13 (the tail rays bit for bit:
    the same kernel), to the C oracle to 1e-12, rays that leave the grid NaN + flag on either side of the split."""
    from oracle import oracle as O
    rng = np.random.default_rng(4200 + seed)
    cubic = interp == "cubic"
    n = [int(v) for v in rng.integers(24, 64, 3)]
    n[2] += n[2] & 1 if cubic else 0
    xv, yv, zv = (np.linspace(0.0, float(rng.uniform(60, 200)), m) for m in n)
    hmin = int(rng.choice([2, 8, 24]))
    n_dense, n_sparse = int(rng.integers(300, 2500)), int(rng.integers(50, 1500))
    o, d, zhi, inside = mixed_rays(rng, xv, yv, zv, n_dense, n_sparse, cubic)
    R, Ns = len(o), int(rng.choice([9, 33, 64, 65, 129]))
    quad = ["avg", "scipy", "trapz"][seed % 3]
    eng = engine(xv, yv, zv, force_bundle=False, quad=quad, interp=interp,
                 env={"IONOTOMO_HYBRID_MIN": hmin})
    M = rng.uniform(1, 2, size=n)
    eng.set_values(eng.tensor(M))
    ot, dt = eng.tensor(o), eng.tensor(d)
    direct = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    assert eng.check_oob() == (not inside.all())
    nb, _, fit = eng.plan_forward(ot, dt, zhi, Ns)
    sp = eng.forward_plan_split(histogram=True)
    assert sp["bundles_served"] == nb and sp["rays_served"] + sp["rays_tail"] == R and sp["min_rays_per_served_bundle"] == hmin
    hist = np.array(sp["bundles_by_ray_count"])
    assert hist.sum() == sp["bundles_cut"] and (hist * np.arange(65)).sum() == R
    assert sp["bundles_served"] == hist[hmin:].sum() and sp["rays_served"] == (hist * np.arange(65))[hmin:].sum()
    # a genuinely mixed launch whenever the cut left bundles on both sides of the threshold (it need not: samples coarser than the
    # z cells fit no window and every ray stays alone -- the plan then serves nothing --, and a small grid can bundle every ray)
    if hist[hmin:].sum() > 0 and hist[1:hmin].sum() > 0:
        assert 0 < sp["rays_tail"] < R, sp
    else:
        assert sp["rays_tail"] in (0, R), sp
    got = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    assert eng.check_oob() == (not inside.all())
    assert np.all(np.isnan(got[~inside])) and np.all(np.isfinite(got[inside]))
    scale = np.max(np.abs(direct[inside]))
    assert np.max(np.abs(got[inside] - direct[inside])) <= (1e-12 if cubic else 2e-13) * scale, (n, R, Ns, sp)
    assert np.sum(got[inside] == direct[inside]) >= min(sp["rays_tail"], inside.sum()) - (~inside).sum()    # the tail: the same kernel's bits
    if not cubic and (quad == "avg" or (quad == "scipy" and Ns % 2 == 1)):
        ref = OC.forward_tec_straight(xv, yv, zv, M, o[inside], d[inside], zhi, Ns)
        assert np.max(np.abs(got[inside] - ref)) <= 1e-12 * np.max(np.abs(ref))
    if cubic and Ns % 2 == 1 and quad != "trapz":
        sel = np.flatnonzero(inside)[:300]
        ref = O.forward_tec(O.straight_rays(o[sel], d[sel], zhi, Ns), xv, yv, zv, M, kind=O.INTERP_TRICUBIC)
        assert np.max(np.abs(got[sel] - ref)) < 1e-11 * np.max(np.abs(ref))
    # new node values: both halves of the launch see them (the tricubic field arrays of BOTH layouts are rebuilt)
    M2 = rng.uniform(1, 2, size=n)
    eng.set_values(eng.tensor(M2))
    got2 = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    eng.clear_forward_plan()
    direct2 = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    eng.check_oob()
    assert np.max(np.abs(got2[inside] - direct2[inside])) <= (1e-12 if cubic else 2e-13) * np.max(np.abs(direct2[inside]))
    assert not eng.plan_stale()


def test_hybrid_dispatch_phase_observable():
    """The phase forward on a mixed geometry: served bundles through k_forward_bundle<NF>, the tail through k_forward_phase_u with the
    plan's walk as its ray list; equal to the unplanned launch for 1, 3 and 8 frequencies."""
    rng = np.random.default_rng(99)
    n = (40, 44, 48)
    xv, yv, zv = (np.linspace(0.0, 150.0, m) for m in n)
    na, nt, nd = 6, 5, 40
    o, d, zhi, inside = mixed_rays(rng, xv, yv, zv, 1200, 500, False)
    keep = np.flatnonzero(inside)[:na * nt * nd]
    o, d = o[keep], d[keep]
    assert len(o) == na * nt * nd
    eng = engine(xv, yv, zv, force_bundle=False, env={"IONOTOMO_HYBRID_MIN": 8})
    eng.set_values(eng.tensor(rng.uniform(1e11, 2e11, size=n)))
    ot, dt = eng.tensor(o), eng.tensor(d)
    clock, const = eng.tensor(rng.normal(size=(na, nt)) * 1e-9), eng.tensor(rng.normal(size=na))
    for nf in (1, 3, 8):
        freqs = np.linspace(110e6, 170e6, nf)
        eng.clear_forward_plan()
        direct = eng.forward_phase(ot, dt, na, nt, nd, zhi, 65, freqs, clock, const, 1).cpu().numpy()
        eng.plan_forward(ot, dt, zhi, 65)
        sp = eng.forward_plan_split()
        assert sp["bundles_served"] > 0 and sp["rays_tail"] > 0
        got = eng.forward_phase(ot, dt, na, nt, nd, zhi, 65, freqs, clock, const, 1).cpu().numpy()
        assert np.all(np.isfinite(got)) and np.max(np.abs(got - direct)) < 1e-11 * np.max(np.abs(direct)), nf
    assert not eng.check_oob()


TEC_RTOL_F32_FAST = 1e-6      # north_star's bound; float32 storage + packed-float32 interpolation measures ~1e-7 (printed below)


@pytest.mark.parametrize("seed", range(SOAK * 10))
def test_f32_fast_mode_random_geometries(seed, OC):
    """VERDICT r5 item 2 / SURVEY 7 step 4: storage="f32" + a forward plan = the float32 fast mode (k_forward_bundle_f32: float32 window
    images, packed-float32 interpolation, float64 sums of the chunk sums).  Against the C oracle on the float64 values at north_star's
    1e-6, against the unplanned float32 kernels (float64 arithmetic on the same rounded values) likewise; rays that leave the grid NaN +
    flag; grids of every nz alignment (windows start on multiples of four levels: unaligned 16-byte loads otherwise), windows that
    do not fit (steep rays: direct loads), every quadrature rule, hybrid splits."""
    rng = np.random.default_rng(9100 + seed)
    n = [int(v) for v in rng.integers(8, 70, 3)]
    xv, yv, zv = (np.linspace(0.0, float(rng.uniform(20, 200)), m) for m in n)
    R = int(rng.integers(1, 1500))
    Ns = int(rng.choice([2, 7, 8, 9, 17, 33, 64, 65, 100, 257]))
    steep = float(rng.choice([0.02, 0.3, 1.5]))
    o, d, zhi, inside = random_rays(rng, xv, yv, zv, R, steep, cluster=bool(seed % 2))
    quad = ["avg", "scipy", "trapz"][seed % 3]
    eng = engine(xv, yv, zv, force_bundle=seed % 4 != 3, env={} if seed % 4 != 3 else {"IONOTOMO_HYBRID_MIN": 4}, quad=quad, storage="f32")
    M = rng.uniform(1, 2, size=n)
    eng.set_values(eng.tensor(M))
    ot, dt = eng.tensor(o), eng.tensor(d)
    direct = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    assert eng.check_oob() == (not inside.all())
    nb, nchunks, fit = eng.plan_forward(ot, dt, zhi, Ns)
    assert nchunks == (Ns + 7) // 8 and 0.0 <= fit <= 1.0
    tec = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    assert eng.check_oob() == (not inside.all())
    assert np.all(np.isnan(tec[~inside])) and np.all(np.isfinite(tec[inside]))
    if inside.any():
        scale = np.max(np.abs(direct[inside]))
        assert np.max(np.abs(tec[inside] - direct[inside])) <= TEC_RTOL_F32_FAST * scale, (n, R, Ns, steep, fit, nb)
        if quad == "avg" or (quad == "scipy" and Ns % 2 == 1):
            ref = OC.forward_tec_straight(xv, yv, zv, M, o[inside], d[inside], zhi, Ns)
            assert np.max(np.abs(tec[inside] - ref) / np.abs(ref)) <= TEC_RTOL_F32_FAST, (n, R, Ns, steep, fit, nb)
    assert not eng.plan_stale()


def test_f32_fast_mode_bench_shape(OC):
    """The bench shape through the float32 fast mode: every ray against the float64 bundle kernel's TEC at 1e-6 (measured ~1e-7,
    printed), dTEC against SURVEY 8(d)'s gate 1e-6 max|TEC|, a sample against the C oracle; edited rays fall back to direct loads."""
    import bench
    w = bench.build_workload(0)
    ne = np.exp(w["m"]) * (w["K_ne"] / 1e13)
    e64 = engine(w["xvec"], w["yvec"], w["zvec"], force_bundle=False)
    e32 = engine(w["xvec"], w["yvec"], w["zvec"], force_bundle=False, storage="f32")
    for e in (e64, e32):
        e.set_values(e.tensor(ne))
    ot, dt = e64.tensor(w["origins"]), e64.tensor(w["directions"])
    e64.plan_forward(ot, dt, bench.TMAX, bench.NS)
    ref = e64.forward(ot, dt, bench.TMAX, bench.NS)
    unplanned = e32.forward(ot, dt, bench.TMAX, bench.NS).clone()
    nb, _, fit = e32.plan_forward(ot, dt, bench.TMAX, bench.NS)
    assert nb > 2000 and fit > 0.99
    got = e32.forward(ot, dt, bench.TMAX, bench.NS)
    assert not e32.check_oob() and not e32.plan_stale()
    rel = float(((got - ref).abs() / ref.abs()).max())
    rel_u = float(((unplanned - ref).abs() / ref.abs()).max())
    print("float32 fast mode: max rel TEC error vs float64 %.3g (float32 storage with float64 arithmetic: %.3g)" % (rel, rel_u))
    assert rel <= 2e-7              # at the bench shape even tests/test_gpu_parity.py's TEC_RTOL_F32 holds (measured 1.06e-7)
    na = bench.NA
    dg, dr = got.view(na, -1) - got.view(na, -1)[0:1], ref.view(na, -1) - ref.view(na, -1)[0:1]
    assert float((dg - dr).abs().max()) <= 1e-6 * float(ref.abs().max())            # dTEC atol = 1e-6 max|TEC|
    sel = np.random.default_rng(1).choice(len(w["origins"]), 2000, replace=False)
    oc = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], ne, w["origins"][sel], w["directions"][sel], bench.TMAX, bench.NS)
    assert np.max(np.abs(got.cpu().numpy()[sel] - oc) / np.abs(oc)) <= TEC_RTOL_F32_FAST
    # rays edited in place: those bundles from the arrays with direct loads, flag raised
    o2 = w["origins"].copy()
    o2[5] = o2[9000]
    ot.copy_(e64.tensor(o2))
    got2 = e32.forward(ot, dt, bench.TMAX, bench.NS)
    assert e32.plan_stale()
    assert bool(torch.isfinite(got2).all())
    keep = torch.ones(len(o2), dtype=torch.bool, device=got.device)
    keep[5] = False
    assert float(((got2 - got).abs() / got.abs())[keep].max()) <= 1e-6


def test_tec_does_not_depend_on_the_bundling():
    """The same rays inside different batches (hence different bundles, windows and LDS / direct-load decisions) give the same
    bits: per-ray arithmetic is position-for-position that of the direct loads and the four z-parts add in a fixed order."""
    w = syn.make_workload("cfg2")
    eng = engine(w["xvec"], w["yvec"], w["zvec"])
    eng.set_values(eng.tensor(w["ne"] / 1e13))
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    rng = np.random.default_rng(5)
    # many rays per antenna: jitter the directions so that bundles fill
    o = np.repeat(o, 6, axis=0)
    d = np.repeat(d, 6, axis=0) + rng.normal(size=(o.shape[0], 3)) * np.array([5e-4, 5e-4, 0.0])
    ot, dt = eng.tensor(o), eng.tensor(d)
    nb, _, fit = eng.plan_forward(ot, dt, w["tmax"], w["Ns"])
    assert nb > 0 and fit > 0.5
    full = eng.forward(ot, dt, w["tmax"], w["Ns"]).cpu().numpy()
    for k in range(3):
        sel = np.sort(rng.choice(o.shape[0], size=o.shape[0] // (2 + k), replace=False))
        os_, ds_ = eng.tensor(o[sel]), eng.tensor(d[sel])
        nb2, _, _ = eng.plan_forward(os_, ds_, w["tmax"], w["Ns"])
        assert nb2 > 0
        part = eng.forward(os_, ds_, w["tmax"], w["Ns"]).cpu().numpy()
        assert np.array_equal(part, full[sel])
    assert not eng.check_oob()


def test_stale_or_foreign_plan_is_never_used(OC):
    w = syn.make_workload("cfg1")
    eng = engine(w["xvec"], w["yvec"], w["zvec"])
    M = w["ne"] / 1e13
    eng.set_values(eng.tensor(M))
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    ot, dt = eng.tensor(o), eng.tensor(d)
    ref = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], M, o, d, w["tmax"], w["Ns"])
    assert eng.plan_forward(ot, dt, w["tmax"], w["Ns"])[0] > 0
    # other tensors / other Ns / other tmax: the plan does not apply, the launch is served by the direct kernels
    o2, d2 = eng.tensor(o[::-1].copy()), eng.tensor(d[::-1].copy())
    t2 = eng.forward(o2, d2, w["tmax"], w["Ns"]).cpu().numpy()
    assert np.max(np.abs(t2 - ref[::-1])) < 1e-12 * np.max(np.abs(ref))
    t3 = eng.forward(ot, dt, w["tmax"], w["Ns"] + 2).cpu().numpy()
    ref3 = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], M, o, d, w["tmax"], w["Ns"] + 2)
    assert np.max(np.abs(t3 - ref3)) < 1e-12 * np.max(np.abs(ref3))
    # the planned launch itself
    t1 = eng.forward(ot, dt, w["tmax"], w["Ns"]).cpu().numpy()
    assert np.max(np.abs(t1 - ref)) < 1e-12 * np.max(np.abs(ref))
    # new values on the same axes keep the plan (geometry only) ...
    import ctypes
    nbun = ctypes.c_int64(-1)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    eng.set_values(eng.tensor(2.0 * M))
    eng.ctx.call("iono_forward_plan_info", ctypes.byref(nbun), None, None)
    assert nbun.value > 0
    t4 = eng.forward(ot, dt, w["tmax"], w["Ns"]).cpu().numpy()
    assert np.max(np.abs(t4 - 2.0 * ref)) < 1e-12 * np.max(np.abs(ref))
    # ... new axes clear it
    xs = w["xvec"] - 0.37 * (w["xvec"][1] - w["xvec"][0])
    eng.set_grid(xs, w["yvec"], w["zvec"])
    eng.set_values(eng.tensor(M))
    eng.ctx.call("iono_forward_plan_info", ctypes.byref(nbun), None, None)
    assert nbun.value == 0
    t5 = eng.forward(ot, dt, w["tmax"], w["Ns"]).cpu().numpy()
    ref5 = OC.forward_tec_straight(xs, w["yvec"], w["zvec"], M, o, d, w["tmax"], w["Ns"])
    assert np.max(np.abs(t5 - ref5)) < 1e-12 * np.max(np.abs(ref5))
    assert not eng.check_oob()


@pytest.mark.parametrize("storage", ["f64", "f32"])
def test_rays_on_the_grid_faces_and_the_last_bundle(OC, storage):
    """Vertical rays exactly on the low / high x and y faces, feet on the bottom face, ends on the top face: windows that touch
    the padded plane / row beyond the grid (weight-0 corners), and a ray count that leaves a one-ray last bundle."""
    n = (12, 10, 16)
    xv, yv, zv = np.linspace(-3.0, 8.0, n[0]), np.linspace(0.0, 9.0, n[1]), np.linspace(0.0, 30.0, n[2])
    rng = np.random.default_rng(3)
    M = rng.uniform(1, 2, size=n)
    feet = [(xv[0], yv[0]), (xv[-1], yv[-1]), (xv[0], yv[-1]), (xv[-1], yv[0]), (xv[3], yv[-1]), (xv[-1], yv[4]), (xv[5], yv[5])]
    o = np.array([[x, y, zv[0]] for x, y in feet] * 19 + [[xv[2], yv[2], zv[0]]])             # 134 rays
    d = np.tile([0.0, 0.0, 1.0], (o.shape[0], 1))
    for Ns in (16, 17, 31):
        eng = engine(xv, yv, zv, storage=storage)            # ("f32": the float32 fast mode's windows touch the same padded plane / row)
        eng.set_values(eng.tensor(M))
        ot, dt = eng.tensor(o), eng.tensor(d)
        assert eng.plan_forward(ot, dt, zv[-1], Ns)[0] >= 3
        assert eng.describe("forward", ot, dt, zv[-1], Ns)[0].startswith("k_forward_bundle_f32" if storage == "f32" else "k_forward_bundle<0>")
        tec = eng.forward(ot, dt, zv[-1], Ns).cpu().numpy()
        assert not eng.check_oob()
        ref = OC.forward_tec_straight(xv, yv, zv, M, o, d, zv[-1], Ns)
        assert np.max(np.abs(tec - ref)) < (1e-12 if storage == "f64" else TEC_RTOL_F32_FAST) * np.max(np.abs(ref))


def test_bundle_forward_bench_shape(OC):
    import bench
    w = bench.build_workload(0)
    eng = engine(w["xvec"], w["yvec"], w["zvec"], force_bundle=False)            # the default dispatch: 4 838 bundles >= 2 per CU
    ne = np.exp(w["m"]) * (w["K_ne"] / 1e13)
    eng.set_values(eng.tensor(ne))
    ot, dt = eng.tensor(w["origins"]), eng.tensor(w["directions"])
    direct = eng.forward(ot, dt, bench.TMAX, bench.NS).cpu().numpy()
    nb, nchunks, fit = eng.plan_forward(ot, dt, bench.TMAX, bench.NS)
    R = w["origins"].shape[0]
    assert R / 64 <= nb <= R / 40 and nchunks == 33 and fit > 0.99
    tec = eng.forward(ot, dt, bench.TMAX, bench.NS).cpu().numpy()
    assert not eng.check_oob()
    assert np.max(np.abs(tec - direct) / np.abs(direct)) < 1e-13
    sel = np.random.default_rng(0).choice(R, 3000, replace=False)
    ref = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], ne, w["origins"][sel], w["directions"][sel], bench.TMAX, bench.NS)
    assert np.max(np.abs(tec[sel] - ref) / np.abs(ref)) < 1e-12
    # bound to a caller-owned values buffer (what the solvers do): same kernel, same numbers
    padded, view = eng.new_grid_buffer()
    view.copy_(eng.tensor(ne))
    eng.bind_values(padded)
    tec2 = eng.forward(ot, dt, bench.TMAX, bench.NS).cpu().numpy()
    assert np.array_equal(tec2, tec)


def test_phase_observable_on_the_bundle_plan(OC):
    """iterative_newton.forward_equation's observable (per-frequency integrals of 1 - sqrt(1 - ne / n_p)) through k_forward_bundle<NF>:
    equal to the direct-load kernels for 1, 3 and 8 frequencies, rays that leave the grid give NaN + flag."""
    import bench
    w = bench.build_workload(0)
    na, nt, nd = bench.NA, 12, bench.ND
    sel = np.arange(bench.NA * bench.NT * bench.ND).reshape(bench.NA, bench.NT, bench.ND)[:, :nt].reshape(-1)
    o, d = w["origins"][sel].copy(), w["directions"][sel].copy()
    eng = engine(w["xvec"], w["yvec"], w["zvec"])
    eng.set_log_model(eng.tensor(w["m"]), w["K_ne"])                    # ne [m^-3]
    ot, dt = eng.tensor(o), eng.tensor(d)
    rng = np.random.default_rng(0)
    clock = eng.tensor(rng.normal(size=(na, nt)) * 1e-9)
    const = eng.tensor(rng.normal(size=na))
    for nf in (1, 3, 8):
        freqs = np.linspace(110e6, 170e6, nf)
        eng.clear_forward_plan()
        direct = eng.forward_phase(ot, dt, na, nt, nd, bench.TMAX, bench.NS, freqs, clock, const, 2).cpu().numpy()
        nb, _, fit = eng.plan_forward(ot, dt, bench.TMAX, bench.NS)
        assert nb > 0 and fit > 0.9
        planned = eng.forward_phase(ot, dt, na, nt, nd, bench.TMAX, bench.NS, freqs, clock, const, 2).cpu().numpy()
        assert planned.shape == (na, nt, nd, nf) and np.all(np.isfinite(planned))
        assert np.max(np.abs(planned - direct)) < 1e-11 * np.max(np.abs(direct)), nf
    assert not eng.check_oob()
    bad = d.copy()
    bad[7, :2] = [3.0, 3.0]                                              # one ray out through the side
    bt = eng.tensor(bad)
    eng.plan_forward(ot, bt, bench.TMAX, bench.NS)
    g = eng.forward_phase(ot, bt, na, nt, nd, bench.TMAX, bench.NS, np.array([150e6]), clock, const, 2).cpu().numpy().reshape(-1)
    assert eng.check_oob() and np.isnan(g[7]) and np.isfinite(g[8])


@pytest.mark.parametrize("seed", range(SOAK * 8))
def test_bundle_tricubic_forward_random_geometries(seed):
    """k_forward_bundle_lm (one Lekien-Marsden field pair per wave, windows staged from the pair-major arrays) against the
    lanes = samples tricubic kernel and the numpy oracle: random clustered / scattered rays incl. rays outside the tricubic domain
    (NaN + flag), windows that do not fit (steep rays), changed grid values (the pair arrays are rebuilt), every quadrature rule."""
    from oracle import oracle as O
    rng = np.random.default_rng(7000 + seed)
    n = [int(v) for v in rng.integers(8, 60, 3)]
    n[2] += n[2] & 1
    xv, yv, zv = (np.linspace(0.0, float(rng.uniform(20, 200)), m) for m in n)
    R = int(rng.integers(1, 1200))
    Ns = int(rng.choice([2, 7, 8, 9, 17, 64, 65, 129]))
    steep = float(rng.choice([0.02, 0.3, 1.0]))
    o, d, zhi, _ = random_rays(rng, xv, yv, zv, R, steep, cluster=bool(seed % 2))
    zhi = min(zhi, zv[-3] - 1e-9)
    o[:, 2] = np.maximum(o[:, 2], zv[2] + 1e-9)
    end = o + d * ((zhi - o[:, 2]) / d[:, 2])[:, None]
    lo, hi = np.array([xv[2], yv[2]]), np.array([xv[-3], yv[-3]])
    inside = np.all((o[:, :2] >= lo) & (o[:, :2] <= hi) & (end[:, :2] >= lo) & (end[:, :2] <= hi), axis=1)
    quad = ["avg", "scipy", "trapz"][seed % 3]
    eng = engine(xv, yv, zv, quad=quad, interp="cubic")
    M = rng.uniform(1, 2, size=n)
    eng.set_values(eng.tensor(M))
    ot, dt = eng.tensor(o), eng.tensor(d)
    direct = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    assert eng.check_oob() == (not inside.all())
    nb, nchunks, fit = eng.plan_forward(ot, dt, zhi, Ns)
    assert nb > 0
    got = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    assert eng.check_oob() == (not inside.all())
    assert np.all(np.isnan(got[~inside])) and np.all(np.isnan(direct[~inside]))
    if inside.any():
        scale = np.max(np.abs(direct[inside]))
        assert np.max(np.abs(got[inside] - direct[inside])) < 1e-12 * scale, (n, R, Ns, steep, fit)
        if Ns % 2 == 1 and quad != "trapz" and inside.sum() <= 400:
            rays = O.straight_rays(o[inside], d[inside], zhi, Ns)
            ref = O.forward_tec(rays, xv, yv, zv, M, kind=O.INTERP_TRICUBIC)
            assert np.max(np.abs(got[inside] - ref)) < 1e-11 * np.max(np.abs(ref))
    M2 = rng.uniform(1, 2, size=n)
    eng.set_values(eng.tensor(M2))
    got2 = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    eng.clear_forward_plan()
    direct2 = eng.forward(ot, dt, zhi, Ns).cpu().numpy()
    eng.check_oob()
    if inside.any():
        assert np.max(np.abs(got2[inside] - direct2[inside])) < 1e-12 * np.max(np.abs(direct2[inside]))


def test_bundle_tricubic_bench_shape():
    import bench
    w = bench.build_workload(0)
    eng = engine(w["xvec"], w["yvec"], w["zvec"], force_bundle=False, interp="cubic")
    eng.set_values(eng.tensor(np.exp(w["m"])))
    ot, dt = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
    direct = eng.forward(ot, dt, bench.TMAX, bench.NS)
    nb, _, fit = eng.plan_forward(ot, dt, bench.TMAX, bench.NS)
    assert nb > 2000 and fit > 0.9
    got = eng.forward(ot, dt, bench.TMAX, bench.NS)
    assert not eng.check_oob()
    assert float((got - direct).abs().max()) < 1e-12 * float(direct.abs().max())
    import time
    for name, plan in (("bundle", True), ("direct", False)):
        if not plan:
            eng.clear_forward_plan()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            eng.forward(ot, dt, bench.TMAX, bench.NS)
        torch.cuda.synchronize()
        print("tricubic forward %s: %.3f ms" % (name, (time.perf_counter() - t) / 20 * 1e3))


@pytest.mark.parametrize("interp", ["linear", "cubic"])
def test_planned_tensors_edited_in_place_never_give_garbage(OC, interp):
    """a plan is keyed on device pointers -- ``o_t.copy_(new)`` into planned tensors used to make k_forward_bundle
    interpolate from LDS addresses outside the staged window.  Every planned launch now checks a 64-bit checksum per ray against the
    plan's: the forward recomputes the bundles concerned from the arrays with direct loads (exact; the tricubic one, whose derivative
    fields exist only where the PLANNED rays read them since round 5, straight from the node values: 216 taps per sample), the
    back-projection poisons the edited rays (NaN, never a plausible number), both raise the flag ``plan_stale`` / ``check_plans`` report."""
    import bench
    from oracle import oracle as O
    w = bench.build_workload(0)
    sel = np.arange(bench.NA * bench.NT * bench.ND).reshape(bench.NA, bench.NT, bench.ND)[:, :10].reshape(-1)      # 26 040 rays
    o, d = w["origins"][sel].copy(), w["directions"][sel].copy()
    eng = engine(w["xvec"], w["yvec"], w["zvec"], interp=interp)
    eng.set_log_model(eng.tensor(w["m"]), w["K_ne"] / 1e13)
    ot, dt = eng.tensor(o), eng.tensor(d)
    nb, _, fit = eng.plan_forward(ot, dt, bench.TMAX, bench.NS)
    assert nb > 0 and fit > 0.9
    planned = eng.forward(ot, dt, bench.TMAX, bench.NS).cpu().numpy()
    assert not eng.plan_stale()
    # (1) a handful of rays edited in place: other directions, one moved to another station
    rng = np.random.default_rng(3)
    o2, d2 = o.copy(), d.copy()
    hit = rng.choice(len(o), size=40, replace=False)
    d2[hit, :2] += rng.normal(size=(40, 2)) * 0.02
    o2[hit[0]] = o[(hit[0] + 5000) % len(o)]
    ot.copy_(eng.tensor(o2))
    dt.copy_(eng.tensor(d2))
    got = eng.forward(ot, dt, bench.TMAX, bench.NS).cpu().numpy()
    assert eng.plan_stale() and not eng.plan_stale()                        # raised once, cleared by the read
    eng2 = engine(w["xvec"], w["yvec"], w["zvec"], interp=interp)            # the same rays on a context that never saw a plan
    eng2.set_log_model(eng2.tensor(w["m"]), w["K_ne"] / 1e13)
    fresh = eng2.forward(eng2.tensor(o2), eng2.tensor(d2), bench.TMAX, bench.NS).cpu().numpy()
    assert np.all(np.isfinite(got))
    assert np.max(np.abs(got - fresh)) <= 1e-12 * np.max(np.abs(fresh))
    assert np.max(np.abs(got[hit] - planned[hit])) > 1e-6 * np.max(np.abs(planned))          # (the edit did change those rays)
    keep = np.setdiff1d(np.arange(len(o)), hit)
    assert np.max(np.abs(got[keep] - planned[keep])) <= 1e-12 * np.max(np.abs(planned))
    # (2) a full rewrite: every ray another ray
    ot.copy_(eng.tensor(o[::-1].copy()))
    dt.copy_(eng.tensor(d[::-1].copy()))
    got = eng.forward(ot, dt, bench.TMAX, bench.NS).cpu().numpy()
    assert np.max(np.abs(got - planned[::-1])) <= 1e-12 * np.max(np.abs(planned))
    with pytest.raises(ValueError):
        eng.check_plans()
    # (3) a new plan on the rewritten tensors: planned path again, no flag
    eng.plan_forward(ot, dt, bench.TMAX, bench.NS)
    again = eng.forward(ot, dt, bench.TMAX, bench.NS).cpu().numpy()
    assert not eng.plan_stale()
    assert np.max(np.abs(again - planned[::-1])) <= 1e-12 * np.max(np.abs(planned))
    # (4) the back-projection works from the plan's own ray records: edited rays come out as NaN + flag, never as the old rays' answer
    eng.plan_adjoint(ot, dt, bench.TMAX, bench.NS)
    y = eng.tensor(rng.normal(size=len(o)))
    g0 = eng.adjoint(ot, dt, y, bench.TMAX, bench.NS).cpu().numpy()
    assert np.all(np.isfinite(g0)) and not eng.plan_stale()
    dt[hit[1], 0] += 0.01                                                     # ONE ray, in place
    g1 = eng.adjoint(ot, dt, y, bench.TMAX, bench.NS).cpu().numpy()
    assert eng.plan_stale() and np.isnan(g1).any()
    # ... and the ray restored in place: the hashes match again, but the plan's record stays poisoned -- so the flag must be raised
    # again by every later launch until the rays are planned anew (flag and NaN never come apart)
    dt[hit[1], 0] -= 0.01
    g1b = eng.adjoint(ot, dt, y, bench.TMAX, bench.NS).cpu().numpy()
    assert np.isnan(g1b).any() and eng.plan_stale()
    dt[hit[1], 0] += 0.01
    eng.plan_adjoint(ot, dt, bench.TMAX, bench.NS)
    g2 = eng.adjoint(ot, dt, y, bench.TMAX, bench.NS).cpu().numpy()
    assert np.all(np.isfinite(g2)) and not eng.plan_stale() and np.max(np.abs(g2 - g0)) > 0.0


def test_bundle_forward_on_a_grid_with_a_plane_above_2_to_24_bytes(OC):
    """The window prefetch forms lane offsets with 24-bit multiply-adds while a grid plane (ny nz 8 bytes) is below 2^24 bytes; a
    wider grid (here 12 x 8200 x 258: planes of 16.9 MB) takes the 32-bit arithmetic and must give the direct kernel's numbers."""
    rng = np.random.default_rng(77)
    n = (12, 8200, 258)
    assert n[1] * n[2] * 8 > 1 << 24
    xv, yv, zv = np.linspace(0.0, 11.0, n[0]), np.linspace(0.0, 8199.0, n[1]), np.linspace(0.0, 257.0, n[2])
    R, Ns = 8192, 257
    yc = rng.uniform(100, 8100, 64)                                    # 64 clusters of 128 rays: bundles fill
    o = np.stack([rng.uniform(3, 8, R), np.repeat(yc, R // 64) + rng.normal(size=R) * 1.5, np.full(R, 1.3)], 1)
    d = np.stack([rng.normal(size=R) * 3e-3, rng.normal(size=R) * 0.05, np.ones(R)], 1)
    tmax = 254.0
    eng = engine(xv, yv, zv)
    M = rng.uniform(1, 2, size=n)
    eng.set_values(eng.tensor(M))
    ot, dt = eng.tensor(o), eng.tensor(d)
    direct = eng.forward(ot, dt, tmax, Ns).cpu().numpy()
    assert not eng.check_oob()
    nb, _, fit = eng.plan_forward(ot, dt, tmax, Ns)
    assert nb > 0 and fit > 0.5
    tec = eng.forward(ot, dt, tmax, Ns).cpu().numpy()
    assert not eng.check_oob()
    assert np.max(np.abs(tec - direct)) <= 2e-13 * np.max(np.abs(direct))
    sel = rng.choice(R, 64, replace=False)
    ref = OC.forward_tec_straight(xv, yv, zv, M, o[sel], d[sel], tmax, Ns)
    assert np.max(np.abs(tec[sel] - ref)) <= 1e-12 * np.max(np.abs(ref))

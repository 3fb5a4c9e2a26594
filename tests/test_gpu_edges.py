"""Round-2 boundary cases through the C-ABI: rays grazing a LOW grid face, the A/B environment switches (none may
change results), a failed iono_grid_set leaving the ctx intact, the library's own walk order, Fermat(type='s') and
the parity-first Fermat defaults, both even-N quadrature rules against their fixtures.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest
import torch

from ionotomo_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def OC():
    from oracle import oracle_c
    return oracle_c


def rel(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


def grazing_problem(seed, margin_ulp, Ns, n=40, R=256):
    """Rays descending onto the low x / y faces: every ray ends (to within rounding) at the same (X0, Y0), and the
    grid's low faces sit ``margin_ulp`` ulp below the smallest computed end coordinate.  All rays then pass the
    kernels' end-point validity test (real coordinates), while the grid coordinate the sample loop recomputes
    (f0 + k df, error ~1e-14 cells) comes out NEGATIVE for some of them when 1/(Ns-1) is not a power of two
    (floor() then gave cell -1: a wrong sample in the forward, an out-of-bounds atomic in the adjoint)."""
    rng = np.random.default_rng(seed)
    tmax = 1000.0
    X0, Y0 = 20.0 + rng.uniform(0, 1), 17.0 + rng.uniform(0, 1)
    o = np.zeros((R, 3))
    o[:, 0], o[:, 1] = X0 + rng.uniform(5.0, 45.0, R), Y0 + rng.uniform(5.0, 45.0, R)
    o[:, 2] = rng.uniform(0.0, 0.5, R)
    L = tmax - o[:, 2]
    d = np.stack([(X0 - o[:, 0]) / L, (Y0 - o[:, 1]) / L, np.ones(R)], 1)
    p = d / np.sqrt((d * d).sum(1))[:, None]
    xe, ye = o[:, 0] + p[:, 0] / p[:, 2] * L, o[:, 1] + p[:, 1] / p[:, 2] * L
    x_lo, y_lo = xe.min() - margin_ulp * np.spacing(X0), ye.min() - margin_ulp * np.spacing(Y0)
    xv = np.linspace(x_lo, o[:, 0].max() + 1.0, n)
    yv = np.linspace(y_lo, o[:, 1].max() + 1.0, n + 3)
    zv = np.linspace(0.0, tmax, n + 5)
    M = rng.uniform(1.0, 2.0, size=(n, n + 3, n + 5))
    # numpy emulation of the kernels' grid-coordinate arithmetic at the last sample (no FMA: indicative only)
    neg = 0
    for ax, c in ((xv, 0), (yv, 1)):
        ih = 1.0 / ((ax[-1] - ax[0]) / (len(ax) - 1))
        fe = (o[:, c] - ax[0]) * ih + (Ns - 1) * (p[:, c] / p[:, 2] * (L * (1.0 / (Ns - 1))) * ih)
        neg += int((fe < 0).sum())
    return xv, yv, zv, M, o, d, tmax, neg


@pytest.mark.parametrize("Ns", [200, 131])
def test_rays_grazing_low_faces_forward_and_adjoint(OC, Ns):
    from ionotomo_amd.engine import RayEngine
    eng = RayEngine(0)
    negatives = checked = 0
    for seed in range(16):
        # the kernel validates end points with its own (FMA-contracted) arithmetic, which can differ from numpy's by
        # an ulp: take the tightest margin it accepts
        for margin in (0, 1, 2, 3):
            xv, yv, zv, M, o, d, tmax, neg = grazing_problem(seed, margin, Ns)
            eng.set_grid(xv, yv, zv)
            eng.set_values(eng.tensor(M))
            ot, dt = eng.tensor(o), eng.tensor(d)
            tec = eng.forward(ot, dt, tmax, Ns).cpu().numpy()
            if not eng.check_oob():
                break
        else:
            continue
        checked += 1
        negatives += neg
        ref = OC.forward_tec_straight(xv, yv, zv, M, o, d, tmax, Ns)
        assert rel(tec, ref) < 1e-12, seed
        # adjoint into a buffer with guard zones on both sides: nothing may be written outside the gradient
        w = np.random.default_rng(seed + 100).normal(size=len(o))
        ncell, pad = M.size, 1 << 16
        big = torch.zeros(ncell + 2 * pad, dtype=torch.float64, device=eng.device)
        g = big[pad:pad + ncell].view(M.shape)
        order = eng.locality_order(ot, dt, tmax)
        gref = OC.adjoint_straight(xv, yv, zv, o, d, w, tmax, Ns)
        for ordr in (None, order):
            big.zero_()
            eng.adjoint(ot, dt, eng.tensor(w), tmax, Ns, out=g, order=ordr)
            assert float(big[:pad].abs().max()) == 0.0 and float(big[pad + ncell:].abs().max()) == 0.0, seed
            assert np.max(np.abs(g.cpu().numpy() - gref)) < 1e-11 * np.max(np.abs(gref)), seed
        assert not eng.check_oob()
    assert checked >= 8 and negatives > 0, "no accepted problem had a negative grid coordinate: the test lost its teeth"


@pytest.mark.parametrize("env", [{"IONOTOMO_HYBRID_MIN": v} for v in ("1", "8", "65")] + [{"IONOTOMO_VARIANT": v} for v in ("2", "3", "4")] +
                         [{"IONOTOMO_FORCE_GENERAL": v} for v in ("1", "2")] + [{"IONOTOMO_SEG_LANES": v} for v in ("4", "8")] +
                         [{"IONOTOMO_BLOCKS_PER_CU": "2"}, {"IONOTOMO_ADJ_ABLATE": "12"}, {"IONOTOMO_WALK": "5", "IONOTOMO_ADJ_BUNDLE": "64"}])
def test_ab_switches_never_change_results(env, monkeypatch, OC):
    """Every A/B variable the README documents (round 6: IONOTOMO_WALK and IONOTOMO_ADJ_BUNDLE are gone -- set here, they must be
    ignored; the timing ablations exist only in a -DIONO_ABLATION build) leaves forward and adjoint results, unplanned AND planned,
    equal to the oracle's."""
    from ionotomo_amd.engine import RayEngine
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = syn.make_workload(antennas="lofar", na=62, nd=6, nt=4, n=48)
    eng = RayEngine(0)                                   # the variables are read when the ctx is created
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    M = w["ne"] / 1e13
    eng.set_values(eng.tensor(M))
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    ot, dt = eng.tensor(o), eng.tensor(d)
    Ns = 129
    order = eng.locality_order(ot, dt, w["tmax"])
    ref = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], M, o, d, w["tmax"], Ns)
    for ordr in (None, order):
        assert rel(eng.forward(ot, dt, w["tmax"], Ns, order=ordr).cpu().numpy(), ref) < 1e-12
    y = np.random.default_rng(1).normal(size=len(o))
    gref = OC.adjoint_straight(w["xvec"], w["yvec"], w["zvec"], o, d, y, w["tmax"], Ns)
    for ordr in (None, order):
        g = eng.adjoint(ot, dt, eng.tensor(y), w["tmax"], Ns, order=ordr).cpu().numpy()
        assert np.max(np.abs(g - gref)) < 1e-11 * np.max(np.abs(gref))
    eng.plan_forward(ot, dt, w["tmax"], Ns)
    eng.plan_adjoint(ot, dt, w["tmax"], Ns)
    assert rel(eng.forward(ot, dt, w["tmax"], Ns).cpu().numpy(), ref) < 1e-12
    g = eng.adjoint(ot, dt, eng.tensor(y), w["tmax"], Ns).cpu().numpy()
    assert np.max(np.abs(g - gref)) < 1e-11 * np.max(np.abs(gref))
    assert not eng.check_oob() and not eng.plan_stale()


def test_failed_grid_set_leaves_context_intact(OC):
    w = syn.make_workload("cfg1")
    c = _lib.Context(0)
    M = w["ne"] / 1e13
    c.set_grid(w["xvec"], w["yvec"], w["zvec"], M)
    before = c.forward_tec_straight(w["origins"], w["directions"], w["tmax"], 65)
    bad_y = w["yvec"].copy() * 3.0 + 7.0                   # a different spacing / origin ...
    bad_z = w["zvec"].copy()
    bad_z[5] = bad_z[3]                                    # ... and a z axis that is not increasing
    with pytest.raises(ValueError):
        c.set_grid(w["xvec"] * 2.0, bad_y, bad_z, M)
    assert c.grid_shape == M.shape
    after = c.forward_tec_straight(w["origins"], w["directions"], w["tmax"], 65)
    assert np.array_equal(before, after)
    c.close()


def test_plan_then_small_download_then_plan_on_one_context(OC):
    """the first small iono_dev_download on a ctx must not free the plan builders' pinned staging buffer (a stray
    hipHostFree left h_plan dangling: the next plan of a size that fits the stale capacity copied into freed memory, and
    iono_ctx_destroy freed it twice).  One ctx: forward plan, download, forward plan, forward, destroy."""
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload(antennas="lofar", na=62, nd=8, nt=4, n=64)
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    eng = RayEngine(0)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    eng.set_values(eng.tensor(w["ne"] / 1e13))
    ot, dt = eng.tensor(o), eng.tensor(d)
    eng.plan_forward(ot, dt, w["tmax"], 65)
    first = eng.forward(ot, dt, w["tmax"], 65)
    host = np.empty(len(o))
    eng.ctx.call("iono_dev_download", _lib._V(host.ctypes.data), _lib._V(first.data_ptr()), host.nbytes)     # first small download of this ctx
    assert np.array_equal(host, first.cpu().numpy())
    eng.clear_forward_plan()
    ot2, dt2 = ot.clone(), dt.clone()
    eng.plan_forward(ot2, dt2, w["tmax"], 65)          # same size: fits the capacity recorded for the staging buffer
    again = eng.forward(ot2, dt2, w["tmax"], 65).cpu().numpy()
    assert np.array_equal(again, host)
    ref = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], w["ne"] / 1e13, o, d, w["tmax"], 65)
    assert np.max(np.abs(again - ref)) < 1e-12 * np.max(np.abs(ref))
    eng.ctx.close()


def test_library_walk_order_and_host_adjoint(OC):
    w = syn.make_workload(antennas="lofar", na=62, nd=8, nt=4, n=64)
    c = _lib.Context(0)
    c.set_grid(w["xvec"], w["yvec"], w["zvec"], w["ne"] / 1e13)
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    order = c.walk_order(o, d, w["tmax"])
    assert order.dtype == np.int32 and np.array_equal(np.sort(order), np.arange(len(o)))
    # neighbours in the walk are neighbours in space: foot-point + far-end distance of consecutive rays is far below
    # that of the memory order (where consecutive rays share the antenna but fan out to different directions)
    end = o[:, :2] + d[:, :2] * ((w["tmax"] - o[:, 2]) / d[:, 2])[:, None]
    step = lambda idx: np.mean(np.linalg.norm(np.diff(o[idx, :2], axis=0), axis=1) +
                               np.linalg.norm(np.diff(end[idx], axis=0), axis=1))
    assert step(order) < 0.5 * step(np.arange(len(o)))
    y = np.random.default_rng(2).normal(size=len(o))
    g = c.adjoint_straight(o, d, y, w["tmax"], 65)         # R >= 1024: ordered internally
    gref = OC.adjoint_straight(w["xvec"], w["yvec"], w["zvec"], o, d, y, w["tmax"], 65)
    assert np.max(np.abs(g - gref)) < 1e-11 * np.max(np.abs(gref))
    c.close()


# --------------------------------------------------------------------------- Fermat defaults, type='s'
def test_fermat_defaults_are_the_shipped_behaviour(golden):
    """Fermat(..., straight_line_approx=False) with NO further arguments reproduces the reference's shipped mode
    (grad n = 0, trilinear n): pinned to the reference's own output."""
    import ionotomo_amd as it
    g = golden("fermat_shipped")
    ne = syn.ne_model(g["xvec"], g["yvec"], g["zvec"], seed=int(g["ne_seed"]))
    fer = it.Fermat(it.TriCubic(g["xvec"], g["yvec"], g["zvec"], ne), float(g["frequency"]), 'z', False)
    assert fer.bend is False and fer.kind == "linear"
    rays = fer.integrate_rays(g["origins"], g["directions"], float(g["tmax"]), 65)
    ref = g["rays"]
    assert np.max(np.abs(rays[..., :3, :] - ref[..., :3, :])) < 1e-9
    assert np.max(np.abs(rays[..., 3, :] - ref[..., 3, :])) < 2e-6 * np.max(ref[..., 3, :])


def test_fermat_type_s_golden(golden, O):
    import ionotomo_amd as it
    g = golden("fermat_type_s")
    w = syn.make_workload("cfg1")
    smax, N = float(g["smax"]), int(g["N"])
    tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["ne"])
    st = it.Fermat(tci, float(g["frequency"]), 's', True).integrate_rays(g["origins"], g["directions"], smax, N)
    assert np.max(np.abs(st - g["straight"])) < 1e-9 * smax
    assert np.max(np.abs(st - O.straight_rays_s(g["origins"], g["directions"], smax, N))) < 1e-12 * smax
    sh = it.Fermat(tci, float(g["frequency"]), 's', False, substeps=8).integrate_rays(g["origins"], g["directions"], smax, N)
    assert np.max(np.abs(sh - g["shipped"])) < 2e-6 * smax
    field = O.n_field_trilinear(w["xvec"], w["yvec"], w["zvec"], O.ne_to_n(w["ne"], float(g["frequency"])))
    mine = O.fermat_trace(g["origins"], g["directions"], smax, N, field, bend=False, substeps=8, type='s')
    assert np.max(np.abs(sh - mine)) < 1e-9
    x, y, z, s = it.Fermat(tci, float(g["frequency"]), 's', True).integrate_ray(g["origins"][0, 0], g["directions"][0, 0], smax, N)
    assert np.allclose(s, np.linspace(0, smax, N)) and x.shape == (N,)
    # bending in arc length: every lane mapping of the tracer against the oracle's RK4
    from test_oracle_golden import smooth_bending_case
    xv, yv, zv, nM, o, d, tmax = smooth_bending_case()
    ne = (1.0 - nM ** 2) * (30e6 ** 2 / 8.980 ** 2)
    c = _lib.Context(0)
    c.set_grid(xv, yv, zv, ne)
    for kind, fld in (("linear", O.n_field_trilinear(xv, yv, zv, nM)), ("cubic", O.n_field_tricubic(xv, yv, zv, nM))):
        ref = O.fermat_trace(o, d, 0.8 * tmax, 17, fld, bend=True, substeps=4, type='s')
        got = c.trace_fermat(o, d, 0.8 * tmax, 17, 30e6, bend=True, kind=kind, substeps=4, type='s').reshape(ref.shape)
        assert np.max(np.abs(got - ref)) < 1e-8
    c.close()
    with pytest.raises(ValueError):
        it.Fermat(tci, 120e6, 'q', True)


# --------------------------------------------------------------------------- even-N quadrature, both rules pinned
def test_even_N_both_rules_against_their_fixtures(golden):
    """N = nz is the reference default and is even.  The facade default quad='avg' is the reference-era
    simps(even='avg') (fixture forward_tec_even_avg: reference module + that rule); quad='scipy' is what the
    reference computes with the scipy installed here (fixture forward_tec).  Both go through the same kernels."""
    import ionotomo_amd as it
    g_avg, g_sp, c = golden("forward_tec_even_avg"), golden("forward_tec"), golden("cast_ray")
    w = syn.make_workload("cfg1")
    m_tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["m"])
    K_ne, i0 = float(g_avg["K_ne"]), int(g_avg["i0"])
    scale = np.max(np.abs(g_sp["tec64"]))
    d_avg = it.forward_equation(c["rays64"], K_ne, m_tci, i0)                     # default rule
    assert np.max(np.abs(d_avg - g_avg["dtec64"])) < 1e-12 * scale
    d_sp = it.forward_equation(c["rays64"], K_ne, m_tci, i0, quad="scipy")
    assert np.max(np.abs(d_sp - g_sp["dtec64"])) < 1e-12 * scale
    assert np.max(np.abs(d_avg - d_sp)) > 1e-9 * scale                            # the rules really differ
    ctx = _lib.default_context()
    ctx.set_grid(w["xvec"], w["yvec"], w["zvec"], None)
    ctx.set_values_exp(w["m"], K_ne / 1e13)
    for rule, gg in (("avg", g_avg), ("scipy", g_sp)):
        t_rays = ctx.forward_tec_rays(c["rays64"], rule=rule)
        t_str = ctx.forward_tec_straight(w["origins"], w["directions"], w["tmax"], 64, rule=rule).reshape(8, 1, 8)
        assert rel(t_rays, gg["tec64"]) < 1e-12 and rel(t_str, gg["tec64"]) < 1e-12
    # the default rule against the UNMODIFIED reference on the scipy it was written for (scipy 1.7.1 ``simps``; round 3:
    # oracle/make_golden_conda.py) -- even N = 64 and 32, odd N = 65, through the facade and both kernels
    g_un = golden("forward_tec_even_simps_unmodified")
    for N in (64, 32, 65):
        rays = g_un["rays%d" % N]
        d = it.forward_equation(rays, K_ne, m_tci, i0)
        assert np.max(np.abs(d - g_un["dtec%d" % N])) < 1e-12 * scale, N
        assert rel(ctx.forward_tec_rays(rays, rule="avg"), g_un["tec%d" % N]) < 1e-12, N
        t_str = ctx.forward_tec_straight(w["origins"], w["directions"], w["tmax"], N, rule="avg").reshape(8, 1, 8)
        assert rel(t_str, g_un["tec%d" % N]) < 1e-9, N            # (in-kernel rays vs the reference's LSODA rays: 1e-9 km)

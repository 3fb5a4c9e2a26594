"""HIP path vs the CPU oracle, through the C-ABI (ctypes).  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest

from ionotomo_amd import _lib, synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def rel(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


TEC_RTOL = 1e-12      # float64 grid: same arithmetic up to summation order / FMA contraction
TEC_RTOL_F32 = 2e-7   # float32 grid storage: one rounding of each node value (north_star asks <= 1e-6)


# --------------------------------------------------------------------------- TriCubic.interp
def test_interp_matches_reference_golden(ctx, golden):
    g = golden("tci_interp")
    ctx.set_grid(g["xvec"], g["yvec"], g["zvec"], g["M"])
    val = ctx.interp(g["px"], g["py"], g["pz"])
    assert np.max(np.abs(val - g["val"])) < 1e-14
    ex = g["ex"]
    exval = ctx.interp(ex[:, 0], ex[:, 1], ex[:, 2], extrapolate=True)
    assert np.allclose(exval, g["exval"], rtol=1e-12, atol=1e-12)
    for p in ex:                                     # out of bounds -> ValueError, like scipy
        with pytest.raises(ValueError):
            ctx.interp(p[:1], p[1:2], p[2:3])
    with pytest.raises(ValueError):
        ctx.interp(np.array([np.nan]), np.array([g["yvec"][2]]), np.array([g["zvec"][2]]))
    assert ctx.interp(np.zeros((0,)), np.zeros((0,)), np.zeros((0,))).shape == (0,)      # empty input


def test_interp_shapes_and_f32_storage(ctx, golden):
    g = golden("tci_interp")
    ctx.set_grid(g["xvec"], g["yvec"], g["zvec"], g["M"], storage="f32")
    n = 4096
    val = ctx.interp(g["px"][:n].reshape(64, 64), g["py"][:n].reshape(64, 64), g["pz"][:n].reshape(64, 64))
    assert val.shape == (64, 64)
    assert np.max(np.abs(val.ravel() - g["val"][:n])) < 4e-7 * np.max(np.abs(g["M"]))
    assert np.allclose(ctx.get_values(), g["M"].astype(np.float32).astype(np.float64))


def test_nonfinite_grid_values_raise_assertion(ctx, golden):
    g = golden("tci_interp")
    M = g["M"].copy()
    M[3, 4, 5] = np.nan
    with pytest.raises(AssertionError):
        ctx.set_grid(g["xvec"], g["yvec"], g["zvec"], M)
    ctx.set_grid(g["xvec"], g["yvec"], g["zvec"], g["M"])
    with pytest.raises(AssertionError):
        ctx.set_values_exp(np.full(g["M"].shape, 1e6), 1.0)      # exp overflows to inf


def test_tricubic_matches_oracle(ctx, O):
    rng = np.random.default_rng(2)
    xv = np.cumsum(rng.uniform(0.5, 1.5, 14))
    yv = np.cumsum(rng.uniform(0.5, 1.5, 13))
    zv = np.cumsum(rng.uniform(0.5, 1.5, 15))
    M = rng.normal(size=(14, 13, 15))
    ctx.set_grid(xv, yv, zv, M)
    p = rng.uniform(size=(3, 2000))
    x = xv[2] + p[0] * (xv[-3] - xv[2])
    y = yv[2] + p[1] * (yv[-3] - yv[2])
    z = zv[2] + p[2] * (zv[-3] - zv[2])
    val = ctx.interp(x, y, z, kind="cubic")
    assert np.max(np.abs(val - O.tricubic(xv, yv, zv, M, x, y, z))) < 1e-12
    with pytest.raises(ValueError):                 # stencil would leave the grid
        ctx.interp(xv[:1] + 0.1, yv[5:6], zv[5:6], kind="cubic")


def test_tricubic_matches_notebook_coefficients_on_unit_grid(ctx, golden, O):
    g = golden("lm_tricubic")
    ux = np.linspace(0, 8, 9)
    M = np.ascontiguousarray(g["M"][:, :9, :9])
    ctx.set_grid(ux, ux, ux, M)
    for (i, j, k), A in zip(g["ucells"], g["ucoeffs"]):
        for (u, v, w) in ((0.3, 0.6, 0.9), (0.5, 0.5, 0.5), (0.01, 0.99, 0.2)):
            mine = ctx.interp(np.array([ux[i] + u]), np.array([ux[j] + v]), np.array([ux[k] + w]), kind="cubic")[0]
            assert abs(mine - O.lm_polynomial(A, u, v, w)) < 1e-11


# --------------------------------------------------------------------------- ray geometry
def test_trace_straight_matches_cast_ray_golden(ctx, golden, O):
    g = golden("cast_ray")
    for N in (64, 65):
        rays = ctx.trace_straight(g["origins"], g["directions"], float(g["tmax"]), N).reshape(8, 1, 8, 4, N)
        assert np.max(np.abs(rays - g["rays%d" % N])) < 1e-9           # reference: LSODA tolerance
        assert np.max(np.abs(rays - O.straight_rays(g["origins"], g["directions"], float(g["tmax"]), N))) < 1e-12


# --------------------------------------------------------------------------- forward dTEC
def test_forward_equation_golden_cfg1(ctx, golden):
    g, c = golden("forward_tec"), golden("cast_ray")
    w = syn.make_workload("cfg1")
    ctx.set_grid(w["xvec"], w["yvec"], w["zvec"], None)
    ctx.set_values_exp(w["m"], float(g["K_ne"]) / 1e13)
    tec = ctx.forward_tec_rays(c["rays65"])
    assert rel(tec, g["tec65"]) < TEC_RTOL
    tec64 = ctx.forward_tec_rays(c["rays64"], rule="scipy")            # even N: fixture is scipy-1.15 semantics
    assert rel(tec64, g["tec64"]) < TEC_RTOL
    # straight-ray kernel (samples generated in-kernel) == explicit-sample kernel
    tecs = ctx.forward_tec_straight(w["origins"], w["directions"], w["tmax"], 65).reshape(8, 1, 8)
    assert rel(tecs, g["tec65"]) < TEC_RTOL
    tecs64 = ctx.forward_tec_straight(w["origins"], w["directions"], w["tmax"], 64, rule="scipy").reshape(8, 1, 8)
    assert rel(tecs64, g["tec64"]) < TEC_RTOL


def test_forward_equation_facade_golden(golden):
    import ionotomo_amd as it
    g, c = golden("forward_tec"), golden("cast_ray")
    w = syn.make_workload("cfg1")
    m_tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["m"])
    dtec = it.forward_equation(c["rays65"], float(g["K_ne"]), m_tci, int(g["i0"]))
    assert dtec.shape == (8, 1, 8) and not np.any(np.isnan(dtec))
    assert np.max(np.abs(dtec - g["dtec65"])) < TEC_RTOL * np.max(np.abs(g["tec65"]))
    assert np.all(dtec[int(g["i0"])] == 0)
    assert np.all(it.forward_equation_dask(c["rays65"], float(g["K_ne"]), m_tci, int(g["i0"])) == dtec)   # tests/test_forward_equation.py:27


@pytest.mark.parametrize("rule", ["avg", "scipy", "trapz"])
@pytest.mark.parametrize("Ns", [2, 3, 4, 63, 64, 65, 129, 200])
def test_quadrature_rules_all_sample_counts(ctx, O, rule, Ns):
    w = syn.make_workload(antennas="example", na=3, nd=5, nt=2, n=20)
    ctx.set_grid(w["xvec"], w["yvec"], w["zvec"], w["ne"] / 1e13)
    rays = O.straight_rays(w["origins"], w["directions"], w["tmax"], Ns)
    code = _lib.quad_rule(rule)
    ref = O.forward_tec(rays, w["xvec"], w["yvec"], w["zvec"], w["ne"] / 1e13, code)
    assert rel(ctx.forward_tec_rays(rays, rule=rule), ref) < TEC_RTOL
    assert rel(ctx.forward_tec_straight(w["origins"], w["directions"], w["tmax"], Ns, rule=rule).reshape(ref.shape), ref) < TEC_RTOL


def test_forward_nonuniform_grid_and_nonuniform_samples(ctx, O):
    rng = np.random.default_rng(8)
    xv = np.cumsum(rng.uniform(0.5, 1.5, 30)) - 15
    yv = np.cumsum(rng.uniform(0.5, 1.5, 28)) - 14
    zv = np.cumsum(rng.uniform(0.5, 1.5, 40))
    M = rng.uniform(1, 2, size=(30, 28, 40))
    ctx.set_grid(xv, yv, zv, M)
    R, Ns = 37, 51
    t = np.sort(rng.uniform(0, 1, size=(R, Ns)), axis=-1)
    a = np.stack([rng.uniform(xv[2], xv[-3], R), rng.uniform(yv[2], yv[-3], R), rng.uniform(zv[0], zv[5], R)], -1)
    b = np.stack([rng.uniform(xv[2], xv[-3], R), rng.uniform(yv[2], yv[-3], R), rng.uniform(zv[-6], zv[-1], R)], -1)
    pts = a[:, :, None] + (b - a)[:, :, None] * t[:, None, :]
    s = np.linalg.norm(b - a, axis=-1)[:, None] * t
    rays = np.concatenate([pts, s[:, None, :]], axis=1)
    for rule, code in (("avg", 0), ("scipy", 1)):
        assert rel(ctx.forward_tec_rays(rays, rule=rule), O.forward_tec(rays, xv, yv, zv, M, code)) < TEC_RTOL
        assert rel(ctx.forward_tec_rays(rays[:, :, :50], rule=rule), O.forward_tec(rays[:, :, :50], xv, yv, zv, M, code)) < TEC_RTOL
    assert rel(ctx.forward_tec_rays(rays, kind="cubic"), O.forward_tec(rays, xv, yv, zv, M, kind=O.INTERP_TRICUBIC)) < 1e-11 \
        if np.all((rays[:, 2] >= zv[2]) & (rays[:, 2] <= zv[-3])) else True


def test_forward_out_of_bounds_raises_and_empty_is_fine(ctx, O):
    w = syn.make_workload("cfg1")
    ctx.set_grid(w["xvec"], w["yvec"], w["zvec"], w["ne"])
    with pytest.raises(ValueError):
        ctx.forward_tec_straight(w["origins"], w["directions"], w["zvec"][-1] + 50.0, 65)     # rays leave the top
    assert not ctx.check_oob()                                                                  # flag was consumed
    out = ctx.forward_tec_straight(np.zeros((0, 3)), np.zeros((0, 3)), 1000.0, 65)
    assert out.shape == (0,)
    with pytest.raises(ValueError):
        ctx.forward_tec_straight(w["origins"], w["directions"][:4], 1000.0, 65)               # ragged input


def test_forward_cfg2_f64_and_f32_storage(ctx, O):
    """config 2: 62 LOFAR-HBA stations x 42 directions, 128^3, Ns = 129."""
    w = syn.make_workload("cfg2")
    ne = O.ne_from_log_model(w["m"], w["K_ne"])
    rays = O.straight_rays(w["origins"], w["directions"], w["tmax"], w["Ns"])
    ref = O.forward_tec(rays, w["xvec"], w["yvec"], w["zvec"], ne)
    for storage, tol in (("f64", TEC_RTOL), ("f32", TEC_RTOL_F32)):
        ctx.set_grid(w["xvec"], w["yvec"], w["zvec"], None, storage=storage)
        ctx.set_values_exp(w["m"], w["K_ne"] / 1e13)
        tec = ctx.forward_tec_straight(w["origins"], w["directions"], w["tmax"], w["Ns"]).reshape(ref.shape)
        assert np.max(np.abs(tec - ref) / np.abs(ref)) < tol
    # tricubic interpolant of the same field
    ctx.set_grid(w["xvec"], w["yvec"], w["zvec"], ne)
    sub = (slice(0, 62, 7), slice(None), slice(0, 42, 5))
    tec3 = ctx.forward_tec_straight(w["origins"][sub], w["directions"][sub], w["tmax"], w["Ns"], kind="cubic")
    ref3 = O.forward_tec(rays[sub], w["xvec"], w["yvec"], w["zvec"], ne, kind=O.INTERP_TRICUBIC)
    assert rel(tec3.reshape(ref3.shape), ref3) < 1e-11


def test_phase_forward_matches_oracle(ctx, golden, O):
    import ionotomo_amd as it
    from ionotomo_amd.inversion import iterative_newton as itn
    g, c = golden("phase_forward"), golden("cast_ray")
    w = syn.make_workload("cfg1")
    mu = np.log(w["ne"] / 1e11)
    tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["ne"])
    out = itn.forward_equation((mu, g["clock"], g["const"]), tci, c["rays65"], g["freqs"], K=float(g["K"]), i0=int(g["i0"]))
    ref = O.forward_phase(mu, g["clock"], g["const"], w["xvec"], w["yvec"], w["zvec"], c["rays65"], g["freqs"],
                          K=float(g["K"]), i0=int(g["i0"]))
    assert out.shape == (8, 1, 8, 2)
    assert np.max(np.abs(out - ref)) < 1e-11 * np.max(np.abs(ref))
    assert np.allclose(tci.M, np.exp(mu) * 1e11)                      # reference side effect kept
    assert abs(itn.neg_log_like(g["g"], g["dobs"], g["CdCt"]) - float(g["S"])) < 1e-12 * float(g["S"])


def test_phase_forward_reference_compat_matches_the_golden_directly(ctx, golden):
    """The PRODUCT against the reference's actual output (tests/golden/phase_forward.npz: iterative_newton.forward_equation run
    in the build container, transposed-reshape slip of tri_cubic.py:70 included) -- no oracle in between."""
    import ionotomo_amd as it
    from ionotomo_amd.inversion import iterative_newton as itn
    g, c = golden("phase_forward"), golden("cast_ray")
    w = syn.make_workload("cfg1")
    mu = np.log(w["ne"] / 1e11)
    tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["ne"])
    out = itn.forward_equation((mu, g["clock"], g["const"]), tci, c["rays65"], g["freqs"], K=float(g["K"]), i0=int(g["i0"]),
                               compat="reference")
    assert np.max(np.abs(out - g["g"])) < 1e-12 * np.max(np.abs(g["g"]))
    S = itn.neg_log_like(out, g["dobs"], g["CdCt"])
    assert abs(S - float(g["S"])) < 1e-10 * abs(float(g["S"]))
    with pytest.raises(ValueError):
        itn.forward_equation((mu, g["clock"], g["const"]), tci, c["rays65"], g["freqs"], compat="scipy")


# --------------------------------------------------------------------------- adjoint
def test_adjoint_matches_oracle_and_dot_product(ctx, O):
    w = syn.make_workload("cfg1")
    ctx.set_grid(w["xvec"], w["yvec"], w["zvec"], w["ne"])
    rng = np.random.default_rng(5)
    for Ns in (33, 64):
        rays = O.straight_rays(w["origins"], w["directions"], w["tmax"], Ns)
        y = rng.normal(size=rays.shape[:3])
        gref = O.adjoint_tec(rays, w["xvec"], w["yvec"], w["zvec"], y)
        for grad in (ctx.adjoint_straight(w["origins"], w["directions"], y, w["tmax"], Ns), ctx.adjoint_rays(rays, y)):
            assert np.max(np.abs(grad - gref)) < 1e-11 * np.max(np.abs(gref))
        x = rng.normal(size=w["ne"].shape)
        ctx.set_values(x)
        Gx = ctx.forward_tec_straight(w["origins"], w["directions"], w["tmax"], Ns).reshape(y.shape)
        Gty = ctx.adjoint_straight(w["origins"], w["directions"], y, w["tmax"], Ns)
        assert abs(np.sum(Gx * y) - np.sum(x * Gty)) < 1e-10 * np.linalg.norm(Gx) * np.linalg.norm(y)
        ctx.set_values(w["ne"])


def test_compute_gradient_facade_finite_difference(O):
    """The check the reference's tests/test_inversion.py:71-87 intends (disabled there)."""
    import ionotomo_amd as it
    w = syn.make_workload(antennas="example", na=4, nd=3, nt=1, n=12)
    rays = O.straight_rays(w["origins"], w["directions"], w["tmax"], 13)
    rng = np.random.default_rng(1)
    m_tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["m"])
    K, i0 = w["K_ne"], 1
    g0 = it.forward_equation(rays, K, m_tci, i0)
    dobs = g0 + rng.normal(size=g0.shape) * 0.01
    CdCt = np.full(g0.shape, 1e-4)
    grad = it.compute_gradient(rays, g0.copy(), dobs, i0, K, m_tci, None, CdCt, None, None, None)
    ref = O.gradient_log_model(rays, w["xvec"], w["yvec"], w["zvec"], w["m"], K, i0, g0, dobs, CdCt)
    assert np.max(np.abs(grad - ref)) < 1e-10 * np.max(np.abs(ref))

    def S(mm):
        g = it.forward_equation(rays, K, it.TriCubic(w["xvec"], w["yvec"], w["zvec"], mm), i0)
        return 0.5 * np.sum((g - dobs) ** 2 / (CdCt + 1e-15))
    for f in np.argsort(np.abs(grad).ravel())[-3:]:
        e = np.zeros(w["m"].size)
        e[f] = 1e-5
        fd = (S(w["m"] + e.reshape(w["m"].shape)) - S(w["m"] - e.reshape(w["m"].shape))) / 2e-5
        assert abs(fd - grad.ravel()[f]) < 1e-5 * abs(grad.ravel()[f]) + 1e-9


def test_rayop_matmul_and_adjoint(O):
    import ionotomo_amd as it
    w = syn.make_workload("cfg1")
    rays = O.straight_rays(w["origins"], w["directions"], w["tmax"], 33)
    M = w["ne"] / 1e13
    op = it.TECForwardEquation(2, (w["xvec"], w["yvec"], w["zvec"]), M, rays[..., :3, :])
    Ax = op.matmul(np.ones_like(M))
    s = np.concatenate([np.zeros(rays.shape[:3] + (1,)), np.cumsum(np.linalg.norm(np.diff(rays[..., :3, :], axis=-1), axis=-2), -1)], -1)
    r4 = np.concatenate([rays[..., :3, :], s[..., None, :]], axis=-2)
    tec = O.forward_tec(r4, w["xvec"], w["yvec"], w["zvec"], M)
    assert np.max(np.abs(Ax - (tec - tec[2:3]))) < 1e-12 * np.max(np.abs(tec))
    rng = np.random.default_rng(3)
    x, y = rng.normal(size=M.shape), rng.normal(size=Ax.shape)
    assert abs(np.sum(op.matmul(x) * y) - np.sum(x * op.matmul(y, adjoint=True))) < 1e-10 * np.linalg.norm(op.matmul(x)) * np.linalg.norm(y)


# --------------------------------------------------------------------------- Fermat
def test_fermat_shipped_curved_mode_golden(ctx, golden, O):
    g = golden("fermat_shipped")
    ne = syn.ne_model(g["xvec"], g["yvec"], g["zvec"], seed=int(g["ne_seed"]))
    ctx.set_grid(g["xvec"], g["yvec"], g["zvec"], ne)
    rays = ctx.trace_fermat(g["origins"], g["directions"], float(g["tmax"]), 65, float(g["frequency"]), bend=False,
                            kind="linear", substeps=8).reshape(g["rays"].shape)
    ref = g["rays"]
    assert np.max(np.abs(rays[..., :3, :] - ref[..., :3, :])) < 1e-9
    assert np.max(np.abs(rays[..., 3, :] - ref[..., 3, :])) < 2e-6 * np.max(ref[..., 3, :])
    field = O.n_field_trilinear(g["xvec"], g["yvec"], g["zvec"], O.ne_to_n(ne, float(g["frequency"])))
    mine = O.fermat_trace(g["origins"], g["directions"], float(g["tmax"]), 65, field, bend=False, substeps=8)
    assert np.max(np.abs(rays - mine)) < 1e-9


def test_fermat_bending_matches_oracle(ctx, O, monkeypatch):
    from test_oracle_golden import smooth_bending_case
    from ionotomo_amd import _lib
    xv, yv, zv, nM, o, d, tmax = smooth_bending_case()
    ne = (1.0 - nM ** 2) * (30e6 ** 2 / 8.980 ** 2)
    ctx.set_grid(xv, yv, zv, ne)
    # the tracer picks its lane mapping by batch size: 4 (trilinear) / 8 (tricubic) lanes per ray with cell-cached
    # stencils, fewer rays per wave for small batches, lanes = rays for very large ones.  Force each through the
    # threshold env vars (read when a context is created) so all of them are checked on the same rays.
    contexts = [ctx]
    for env in ({"IONOTOMO_FERMAT_COOP_MAX": "0", "IONOTOMO_FERMAT_LIN4_MAX": "0"},
                {"IONOTOMO_FERMAT_LIN4_RPW": "16", "IONOTOMO_FERMAT_COOP_RPW": "3"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        contexts.append(_lib.Context())
        for k_ in env:
            monkeypatch.delenv(k_)
        contexts[-1].set_grid(xv, yv, zv, ne)
    for kind, field in (("cubic", O.n_field_tricubic(xv, yv, zv, O.ne_to_n(ne, 30e6))),
                        ("linear", O.n_field_trilinear(xv, yv, zv, O.ne_to_n(ne, 30e6)))):
        ref = O.fermat_trace(o, d, tmax, 17, field, bend=True, substeps=4)
        for c in contexts:
            rays = c.trace_fermat(o, d, tmax, 17, 30e6, bend=True, kind=kind, substeps=4).reshape(o.shape[:-1] + (4, 17))
            assert np.max(np.abs(rays - ref)) < 1e-8
    straight = O.straight_rays(o, d, tmax, 17)
    assert np.max(np.abs(rays[..., 0, :] - straight[..., 0, :])) > 1.0


def test_calc_rays_facade_shapes(golden):
    import ionotomo_amd as it
    g = golden("cast_ray")
    w = syn.make_workload("cfg1")
    tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["ne"])
    rays = it.calc_rays(w["origins"][:, 0, 0, :], w["directions"][0], [0.0], None, None, None, tci, 120e6, True, 1000.0)
    assert rays.shape == (8, 1, 8, 4, tci.nz)
    rays65 = it.calc_rays(w["origins"][:, 0, 0, :], w["directions"][0], [0.0], None, None, None, tci, 120e6, True, 1000.0, 65)
    assert np.max(np.abs(rays65 - g["rays65"])) < 1e-9
    x, y, z, s = it.Fermat(tci).integrate_ray(w["origins"][1, 0, 2], w["directions"][1, 0, 2], 1000.0, N=65)
    assert np.max(np.abs(x - g["rays65"][1, 0, 2, 0])) < 1e-9 and np.max(np.abs(s - g["rays65"][1, 0, 2, 3])) < 1e-9


# --------------------------------------------------------------------------- C_m smoothing
def test_covariance_smooth_matches_reference_golden(golden, O):
    from ionotomo_amd.ionosphere.covariance import Covariance
    g = golden("covariance_smooth")
    for tag in ("a", "b"):
        dx, dy, dz = g["d_" + tag]
        C = Covariance(dx=dx, dy=dy, dz=dz)
        assert C.c_stencil.shape == g["stencil_" + tag].shape
        assert np.max(np.abs(C.c_stencil - g["stencil_" + tag])) < 1e-15
        out = C.smooth(g["phi_" + tag])
        assert np.max(np.abs(out - g["out_" + tag])) < 1e-12 * np.max(np.abs(g["out_" + tag]))
    # stencil wider than the array (every tap clamps), 2-node-thick axis, z longer than one 64-lane segment
    rng = np.random.default_rng(0)
    phi = rng.normal(size=(3, 2, 70))
    C = Covariance(dx=1.0, dy=1.0, dz=1.0)
    assert np.max(np.abs(C.smooth(phi) - O.smooth(phi, 1.0, 1.0, 1.0))) < 1e-11 * np.max(np.abs(phi)) * C.c_stencil.sum()


def test_calc_rays_from_sky_coordinates():
    import ionotomo_amd as it
    from ionotomo_amd.astro import frames
    ra_ = it.RadioArray(array_file=it.RadioArray.lofar_array)
    lon, lat, _ = frames.geodetic_from_itrs(ra_.get_center())
    t0 = 1.7e9
    phase = (frames.gmst_rad(t0) + lon, lat)
    pat = np.stack([phase[0] + np.array([0.0, 0.01, -0.02]), phase[1] + np.array([0.0, 0.015, 0.01])], -1)
    o, d = frames.model_frame_bundle_from_sky(ra_.get_antenna_locs()[:6], pat, [t0], ra_.get_center(), phase)
    xv, yv, zv = frames.determine_inversion_domain(20.0, o[:, 0, 0, :], d[0, 0], 1000.0, padding=3)
    tci = it.TriCubic(xv, yv, zv, np.ones((len(xv), len(yv), len(zv))))
    rays = it.calc_rays(ra_.get_antenna_locs()[:6], pat, [t0], ra_.get_center(), t0, phase, tci, 120e6, True, 1000.0, 33)
    assert rays.shape == (6, 1, 3, 4, 33)
    assert np.allclose(rays[:, 0, :, 2, -1], 1000.0) and np.allclose(rays[:, 0, :, :3, 0], o[:, 0, :, :])
    tec = it.do_forward_equation(rays[:, 0], tci)                   # unit field: TEC = path length
    assert np.allclose(tec, rays[:, 0, :, 3, -1], rtol=1e-12)


def test_calc_rays_on_reference_typed_objects_equals_the_array_call():
    """The reference's callers hand calc_rays astropy objects (inversion/inversion_pipeline.py:195-197,
    astro/simulate_observables.py:50-62).  Stand-ins exposing exactly the attributes the reference touches
    (tests/astropy_standins.py) must give the plain-array call's rays bit for bit, whatever unit the positions are held in is
    converted, fixtime / times as Time objects, N=None -> ne_tci.nz (geometry/calc_rays.py:111-112)."""
    import sys
    sys.path.insert(0, __import__("os").path.dirname(__file__))
    from astropy_standins import ICRSCoord, ITRSCoord, Time
    import ionotomo_amd as it
    from ionotomo_amd.astro import frames
    ra_ = it.RadioArray(array_file=it.RadioArray.lofar_array)
    ants, centre = ra_.get_antenna_locs()[:7], ra_.get_center()
    lon, lat, _ = frames.geodetic_from_itrs(centre)
    times = 1.49e9 + 8.0 * np.arange(3)
    phase = np.array([(frames.gmst_rad(times[1]) + lon) % (2 * np.pi), lat])
    pat = phase + np.deg2rad(np.random.default_rng(8).uniform(-2, 2, size=(5, 2)))
    o, d = frames.model_frame_bundle_from_sky(ants, pat, times, centre, phase)
    xv, yv, zv = frames.determine_inversion_domain(25.0, o[:, 0, 0, :], d[0].reshape(-1, 3), 1000.0, padding=3)
    tci = it.TriCubic(xv, yv, zv, np.ones((len(xv), len(yv), len(zv))))
    plain = it.calc_rays(ants, pat, times, centre, times[1], phase, tci, 120e6, True, 1000.0, None)
    assert plain.shape == (7, 3, 5, 4, tci.nz)
    typed = it.calc_rays(ITRSCoord(ants), ICRSCoord(pat[:, 0], pat[:, 1]), Time(times), ITRSCoord(centre), Time(times)[1],
                         ICRSCoord(phase[0], phase[1]), tci, 120e6, True, 1000.0, None)
    assert np.array_equal(plain, typed)
    # one timestep as the pipeline's graph slices it (times[time_idx:time_idx+1], fixtime = times[time_idx]: :157-158), positions in km
    one = it.calc_rays(ITRSCoord(ants / 1e3, "km"), ICRSCoord(pat[:, 0], pat[:, 1]), Time(times)[1:2], ITRSCoord(centre / 1e3, "km"),
                       Time(times)[1], ICRSCoord(phase[0], phase[1]), tci, 120e6, True, 1000.0, None)
    assert one.shape == (7, 1, 5, 4, tci.nz)
    assert np.max(np.abs(one[:, 0] - plain[:, 1])) <= 1e-9                 # (km -> m -> km: not bit-identical, 1e-12 relative)
    # a Time-like that only offers .gps (leap seconds handled: astro/coords.py)
    gps = it.calc_rays(ITRSCoord(ants), ICRSCoord(pat[:, 0], pat[:, 1]), Time(times, only="gps"), ITRSCoord(centre), None,
                       ICRSCoord(phase[0], phase[1]), tci, 120e6, True, 1000.0, None)
    assert np.array_equal(plain, gps)


def test_shipped_chord_gradient_from_the_product(ctx, golden, O):
    """the reference's own gradient discretisation (chord lengths, inversion/gradient.py:15-20) is
    available from the product for comparison -- pinned to the reference's ``do_gradient`` output and to the oracle's
    restatement on a larger straight-ray case."""
    g = golden("ray_dirac")
    ctx.set_grid(g["xvec"], g["yvec"], g["zvec"], g["M"])
    grad = ctx.gradient_chords(g["rays"], g["dd"])
    assert np.max(np.abs(grad - g["grad"])) < 1e-12 * np.max(np.abs(g["grad"]))
    rng = np.random.default_rng(0)
    xv, yv, zv = np.linspace(-20, 20, 21), np.linspace(-15, 25, 17), np.linspace(0, 60, 25)
    M = rng.uniform(1, 2, size=(21, 17, 25))
    o = np.stack([rng.uniform(-8, 8, (3, 4)), rng.uniform(-8, 8, (3, 4)), np.full((3, 4), 1.0)], -1)
    d = np.stack([rng.uniform(-0.2, 0.2, (3, 4)), rng.uniform(-0.2, 0.2, (3, 4)), np.ones((3, 4))], -1)
    d[0, 0, :2] = 0.0                                            # a vertical ray: 0 * inf -> NaN -> 0 in the slab method
    rays = O.straight_rays(o, d, 58.0, 30)
    dd = rng.normal(size=(3, 4))
    ctx.set_grid(xv, yv, zv, M)
    ref = O.gradient_chords(rays, xv, yv, zv, M, dd)
    got = ctx.gradient_chords(rays, dd)
    assert np.max(np.abs(got - ref)) < 1e-12 * np.max(np.abs(ref))
    import ionotomo_amd as it
    m_tci = it.TriCubic(xv, yv, zv, np.log(M))
    gch = it.compute_gradient(rays[:, None], np.zeros((3, 1, 4)), -dd[:, None, :] * 1.0, 0, 1e13, m_tci, None, np.ones((3, 1, 4)) - 1e-15,
                              None, None, None, method="chords")
    assert np.max(np.abs(gch - ref)) < 1e-10 * np.max(np.abs(ref))

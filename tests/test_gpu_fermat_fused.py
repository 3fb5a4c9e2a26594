"""Fused curved-ray kernels -- iono_forward_tec_fermat_dev / iono_adjoint_fermat_dev (k_fermat_tec: RK4 stepper + streaming
non-uniform Simpson, rays[R,4,Ns] never materialised) -- against the two-step path they replace (trace_fermat -> forward_rays /
adjoint of explicit rays), the oracle's RK4 + quadrature, every quadrature rule and sample-count parity, both independent
variables, both interpolants, config 3 at its stated size, and 620,000 bending rays through 256^3 in one launch (config 4's ray
count: the explicit rays would be 5.1 GB).  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest
import torch

from ionotomo_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def make_engine(w, **kw):
    from ionotomo_amd.engine import RayEngine
    eng = RayEngine(0, **kw)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    return eng


@pytest.mark.parametrize("quad", ["avg", "scipy", "trapz"])
@pytest.mark.parametrize("Ns", [2, 3, 4, 9, 16, 33])
def test_streaming_quadrature_equals_the_explicit_sample_kernels(quad, Ns):
    """Every rule and parity of N: fused == trace + explicit-sample integral (same samples bit for bit, weights to rounding)."""
    w = syn.make_workload(antennas="example", na=5, nd=4, nt=2, n=20, margin_cells=8)
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    for kind, typ in (("linear", "z"), ("cubic", "z"), ("linear", "s")):
        eng = make_engine(w, quad=quad, interp=kind)
        eng.set_values(eng.tensor(w["ne"]))
        ot, dt = eng.tensor(o), eng.tensor(d)
        tmax = w["tmax"] if typ == "z" else 0.8 * w["tmax"]
        rays = eng.trace_fermat(ot, dt, tmax, Ns, 60e6, bend=True, kind=kind, substeps=3, type=typ)
        two_step = eng.forward_rays(rays).cpu().numpy()
        fused = eng.forward_fermat(ot, dt, tmax, Ns, 60e6, bend=True, kind=kind, substeps=3, type=typ, fused=True).cpu().numpy()
        assert not eng.check_oob()
        assert np.max(np.abs(fused - two_step)) <= 1e-12 * np.max(np.abs(two_step)), (kind, typ)
        # transpose: re-trace + scatter == scatter along the stored rays, and <G x, y> = <x, G^T y>
        y = np.random.default_rng(Ns).normal(size=len(o))
        yt = eng.tensor(y)
        g_f = eng.adjoint_fermat(ot, dt, yt, tmax, Ns, 60e6, bend=True, kind=kind, substeps=3, type=typ, fused=True)
        lhs, rhs = float((eng.tensor(fused) * yt).sum()), float((g_f * eng.tensor(w["ne"])).sum())
        assert abs(lhs - rhs) <= 1e-11 * np.linalg.norm(fused) * np.linalg.norm(y), (kind, typ)
        assert not eng.check_oob()


def test_fused_adjoint_equals_the_explicit_ray_adjoint():
    from ionotomo_amd import _lib
    w = syn.make_workload(antennas="example", na=6, nd=5, nt=1, n=24, margin_cells=8)
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    rng = np.random.default_rng(1)
    y = rng.normal(size=len(o))
    for kind in ("linear", "cubic"):
        eng = make_engine(w, interp=kind)
        eng.set_values(eng.tensor(w["ne"]))
        ot, dt, yt = eng.tensor(o), eng.tensor(d), eng.tensor(y)
        rays = eng.trace_fermat(ot, dt, w["tmax"], 21, 60e6, bend=True, kind=kind, substeps=2)
        ref = torch.zeros(eng.shape, dtype=torch.float64, device=eng.device)
        eng.ctx.call("iono_adjoint_rays_dev", _lib._V(rays.data_ptr()), _lib._V(yt.data_ptr()), len(o), 21, eng.kind, eng.rule,
                     _lib._V(ref.data_ptr()), _lib.F64)
        got = eng.adjoint_fermat(ot, dt, yt, w["tmax"], 21, 60e6, bend=True, kind=kind, substeps=2, ne_scale=1.0, fused=True)
        assert float((got - ref).abs().max()) <= 1e-11 * float(ref.abs().max()), kind
        # ne_scale scales both directions
        got2 = eng.adjoint_fermat(ot, dt, yt, w["tmax"], 21, 60e6, bend=True, kind=kind, substeps=2, ne_scale=1e-13, fused=True)
        assert float((got2 * 1e13 - ref).abs().max()) <= 1e-11 * float(ref.abs().max())
        assert not eng.check_oob()


@pytest.mark.parametrize("kind", ["linear", "cubic"])
def test_config3_fused_equals_trace_then_integrate_and_the_oracle(kind, O):
    w = syn.make_workload("cfg2", margin_cells=16)
    eng = make_engine(w)
    eng.set_values(eng.tensor(w["ne"]))
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    ot, dt = eng.tensor(o), eng.tensor(d)
    Ns, freq = w["Ns"], 120e6
    rays_t = eng.trace_fermat(ot, dt, w["tmax"], Ns, freq, bend=True, kind=kind, substeps=4)
    for tk, ok in (("linear", O.INTERP_TRILINEAR), ("cubic", O.INTERP_TRICUBIC)):
        two_step = eng.forward_rays(rays_t, kind=tk).cpu().numpy()
        fused = eng.forward_fermat(ot, dt, w["tmax"], Ns, freq, bend=True, kind=kind, substeps=4, ne_kind=tk, fused=True).cpu().numpy()
        auto = eng.forward_fermat(ot, dt, w["tmax"], Ns, freq, bend=True, kind=kind, substeps=4, ne_kind=tk, ne_scale=0.5).cpu().numpy()
        assert np.max(np.abs(auto - 0.5 * fused) / np.abs(fused)) < 1e-12       # (a tricubic index: trace + integrate; else the fused kernel)
        assert np.max(np.abs(fused - two_step) / np.abs(two_step)) < 1e-12
        idx = np.sort(np.random.default_rng(0).choice(len(o), 104, replace=False))
        nM = O.ne_to_n(w["ne"], freq)
        field = (O.n_field_trilinear if kind == "linear" else O.n_field_tricubic)(w["xvec"], w["yvec"], w["zvec"], nM)
        ref_rays = O.fermat_trace(o[idx], d[idx], w["tmax"], Ns, field, bend=True, substeps=4)
        tref = O.forward_tec(ref_rays, w["xvec"], w["yvec"], w["zvec"], w["ne"], kind=ok)
        assert np.max(np.abs(fused[idx] - tref) / np.abs(tref)) < 1e-10
    assert not eng.check_oob()


def test_620k_bending_rays_through_256_cubed_in_one_launch():
    """Config 4's ray count with the bending tracer: one launch, no 5.1 GB ray buffer; sampled against trace + integrate; the
    transpose by the dot-product test on the full batch and entry by entry on a subset."""
    w = syn.make_workload("cfg4", margin_cells=16)
    eng = make_engine(w)
    eng.set_values(eng.tensor(w["ne"]))
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    ot, dt = eng.tensor(o), eng.tensor(d)
    R, Ns, freq = len(o), w["Ns"], 150e6
    assert R == 620000
    free0 = torch.cuda.mem_get_info()[0]
    tec = eng.forward_fermat(ot, dt, w["tmax"], Ns, freq, bend=True, kind="linear", substeps=2, ne_scale=1e-13)
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < (1 << 30)                     # nothing like R x 4 x Ns x 8 B = 5.1 GB was allocated
    assert not eng.check_oob()
    tec = tec.cpu().numpy()
    assert np.all(np.isfinite(tec)) and np.all(tec > 0)
    idx = np.sort(np.random.default_rng(2).choice(R, 4096, replace=False))
    oi, di = eng.tensor(o[idx]), eng.tensor(d[idx])
    rays = eng.trace_fermat(oi, di, w["tmax"], Ns, freq, bend=True, kind="linear", substeps=2)
    two = eng.forward_rays(rays).cpu().numpy() * 1e-13
    assert np.max(np.abs(tec[idx] - two) / two) < 1e-12
    straight = eng.forward(oi, di, w["tmax"], Ns).cpu().numpy() * 1e-13
    assert np.max(np.abs(two - straight) / straight) > 1e-6                       # the rays really bend
    # the transpose on the FULL batch (re-trace + scatter through each wave's LDS window): <G x, y> = <x, G^T y>
    y = eng.tensor(np.random.default_rng(3).normal(size=R))
    g = eng.adjoint_fermat(ot, dt, y, w["tmax"], Ns, freq, bend=True, kind="linear", substeps=2, ne_scale=1e-13)
    lhs, rhs = float((eng.tensor(tec) * y).sum()), float((g * eng.tensor(w["ne"])).sum())
    assert abs(lhs - rhs) < 1e-10 * np.linalg.norm(tec) * float(y.norm())
    # ... and on a subset, entry by entry, against the explicit-ray transpose (plain atomics per corner)
    from ionotomo_amd import _lib
    ys = eng.tensor(np.random.default_rng(4).normal(size=len(idx)))
    ref = torch.zeros(eng.shape, dtype=torch.float64, device=eng.device)
    eng.ctx.call("iono_adjoint_rays_dev", _lib._V(rays.data_ptr()), _lib._V(ys.data_ptr()), len(idx), Ns, eng.kind, eng.rule,
                 _lib._V(ref.data_ptr()), _lib.F64)
    got = eng.adjoint_fermat(oi, di, ys, w["tmax"], Ns, freq, bend=True, kind="linear", substeps=2)
    assert float((got - ref).abs().max()) <= 1e-11 * float(ref.abs().max())
    assert not eng.check_oob()


@pytest.mark.parametrize("rep", [1, 80])
@pytest.mark.parametrize("aligned", [False, True])
def test_ideal_grid_right_hand_side_equals_the_general_one(aligned, rep, O, monkeypatch):
    """On an ideal-uniform grid the tracer's right-hand side skips the axis tables and the divisions (trilinear_grad_ideal) except
    within 1e-9 of a cell face, where the general form decides the cell as scipy does.  Traced rays and fused TEC equal the
    general form's (IONOTOMO_FORCE_GENERAL=2) to rounding and the oracle's RK4 -- also when EVERY sample sits on a z face (``aligned``:
    samples one cell apart starting on a level, the case where the face rule decides which cell's gradient bends the ray).
    ``rep`` = 1: 60 rays, the small-batch tracer that keeps the cell's polynomial in registers (k_trace_fermat_poly) against the
    4-lanes-per-ray kernel that serves the other grids (the same switch); 80: 4 800 rays, the lanes = rays tracer."""
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload(antennas="example", na=6, nd=5, nt=2, n=33, margin_cells=10)
    xv, yv, zv = w["xvec"], w["yvec"], w["zvec"]
    o, d = w["origins"].reshape(-1, 3).copy(), w["directions"].reshape(-1, 3)
    Ns, tmax = 21, float(zv[24])
    if aligned:
        o[:, 2] = zv[4]                                   # z_k = linspace(zv[4], zv[24], 21): every sample on a level of the grid
    res = {}
    other = "2"                                           # IONOTOMO_FORCE_GENERAL=2: no ideal-uniform kernels (general right-hand side; 4 lanes per ray)
    for variant in ("0", other):
        monkeypatch.setenv("IONOTOMO_FORCE_GENERAL", variant)
        eng = RayEngine(0)
        monkeypatch.delenv("IONOTOMO_FORCE_GENERAL")
        eng.set_grid(xv, yv, zv)
        eng.set_values(eng.tensor(w["ne"]))
        ot, dt = eng.tensor(np.tile(o, (rep, 1))), eng.tensor(np.tile(d, (rep, 1)))
        rays = eng.trace_fermat(ot, dt, tmax, Ns, 100e6, bend=True, kind="linear", substeps=2)
        tec = eng.forward_fermat(ot, dt, tmax, Ns, 100e6, bend=True, kind="linear", substeps=2)
        assert not eng.check_oob()
        res[variant] = (rays.cpu().numpy()[: len(o)], tec.cpu().numpy()[: len(o)])
    (r0, t0), (r1, t1) = res["0"], res[other]
    assert np.max(np.abs(r0 - r1)) < 1e-10 and np.max(np.abs(t0 - t1)) < 1e-11 * np.max(np.abs(t1))
    field = O.n_field_trilinear(xv, yv, zv, O.ne_to_n(w["ne"], 100e6))
    ref = O.fermat_trace(o, d, tmax, Ns, field, bend=True, substeps=2)
    assert np.max(np.abs(r0 - ref)) < 1e-9


def test_tricubic_index_is_fused_by_default_on_ideal_axes(monkeypatch):
    """Round 4: curved rays through a TRICUBIC refractive index no longer need rays[R,4,Ns].  ``fused=None`` on
    ideal-uniform axes runs k_fermat_tec_lm -- the 8-lanes-per-ray stepper of the record tracer feeding the streaming quadrature --
    and equals trace + integrate along the stored rays to 1e-11 for both integrand interpolants, both independent variables, every
    quadrature rule and even / odd sample counts; the lanes = rays kernel (what grids without ideal-uniform axes get: IONOTOMO_FORCE_GENERAL=2) agrees too.  The TRANSPOSE keeps the
    two-step route by default (its fused form steps lanes = rays with 216 taps)."""
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload(antennas="example", na=6, nd=5, nt=2, n=24, margin_cells=8)
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    yt = None
    for interp in ("cubic", "linear"):                       # the ENGINE's interpolant = the integrand's (ne_kind default)
        eng = make_engine(w, interp=interp)
        eng.set_values(eng.tensor(w["ne"]))
        ot, dt = eng.tensor(o), eng.tensor(d)
        assert eng.fermat_lm_ok("cubic", interp, len(o))               # the library's own predicate (iono_fermat_lm_ok)
        assert not eng._two_step_fermat(len(o), 21, "cubic", None) and eng._two_step_fermat(len(o), 21, "cubic", None, adjoint=True)
        assert not eng._two_step_fermat(len(o), 21, "linear", None)
        for typ in ("z", "s"):
            tmax = w["tmax"] if typ == "z" else 0.8 * w["tmax"]
            for Ns in (21, 22):
                a = eng.forward_fermat(ot, dt, tmax, Ns, 60e6, kind="cubic", substeps=2, ne_scale=1e-13, type=typ)              # fused (default)
                b = eng.forward_fermat(ot, dt, tmax, Ns, 60e6, kind="cubic", substeps=2, ne_scale=1e-13, type=typ, fused=False)  # trace + integrate
                assert float((a - b).abs().max()) < 1e-11 * float(b.abs().max()), (interp, typ, Ns)
        assert not eng.check_oob()
        if yt is None:
            yt = eng.tensor(np.random.default_rng(4).normal(size=len(o)))
        ga = eng.adjoint_fermat(ot, dt, yt, w["tmax"], 21, 60e6, kind="cubic", substeps=2, ne_scale=1e-13)
        gb = eng.adjoint_fermat(ot, dt, yt, w["tmax"], 21, 60e6, kind="cubic", substeps=2, ne_scale=1e-13, fused=True)
        assert float((ga - gb).abs().max()) < 1e-12 * float(gb.abs().max())
    # the lanes = rays kernel on the same problem
    monkeypatch.setenv("IONOTOMO_FORCE_GENERAL", "2")
    e17 = RayEngine(0, interp="cubic")
    e17.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e17.set_values(e17.tensor(w["ne"]))
    c17 = e17.forward_fermat(e17.tensor(o), e17.tensor(d), w["tmax"], 21, 60e6, kind="cubic", substeps=2, ne_scale=1e-13, fused=True)
    # ... which the DEFAULT route never lands on silently: where the library would not run k_fermat_tec_lm, fused=None traces + integrates
    assert not e17.fermat_lm_ok("cubic", "cubic", len(o)) and e17._two_step_fermat(len(o), 21, "cubic", None)
    monkeypatch.delenv("IONOTOMO_FORCE_GENERAL")
    eng = RayEngine(0, interp="cubic")
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    eng.set_values(eng.tensor(w["ne"]))
    a = eng.forward_fermat(eng.tensor(o), eng.tensor(d), w["tmax"], 21, 60e6, kind="cubic", substeps=2, ne_scale=1e-13)
    assert float((a - c17).abs().max()) < 1e-11 * float(c17.abs().max())
    # rays that leave the tricubic domain: skipped samples + flag, as on the two-step route
    d2 = d.copy()
    d2[3, 0] += 0.8
    t2 = eng.forward_fermat(eng.tensor(o), eng.tensor(d2), w["tmax"], 21, 60e6, kind="cubic", substeps=2)
    assert eng.check_oob()
    t3 = eng.forward_fermat(eng.tensor(o), eng.tensor(d2), w["tmax"], 21, 60e6, kind="cubic", substeps=2, fused=False)
    assert eng.check_oob()
    keep = np.arange(len(o)) != 3
    assert float((t2 - t3).abs()[torch.from_numpy(keep).to(t2.device)].max()) < 1e-11 * float(t3.abs().max())


@pytest.mark.parametrize("typ", ["z", "s"])
def test_tricubic_tracer_on_node_records_equals_the_216_tap_kernel(typ, O, monkeypatch):
    """On ideal-uniform grids the tricubic tracer reads the Lekien-Marsden records of the refractive index, one node per lane
    (k_trace_fermat_lm), instead of a 6 x 6 plane of taps per lane (k_trace_fermat_coop: the other grids, IONOTOMO_FORCE_GENERAL=2): the same interpolant,
    so the rays agree to rounding -- both independent variables, a changed model (the records are rebuilt), a second frequency --
    and with the oracle's RK4 on its 216-tap form."""
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload(antennas="example", na=6, nd=5, nt=2, n=33, margin_cells=10)
    xv, yv, zv = w["xvec"], w["yvec"], w["zvec"]
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    Ns = 21
    tmax = float(zv[24]) if typ == "z" else 0.7 * float(zv[24])
    res = {}
    for variant in ("0", "17"):
        monkeypatch.setenv("IONOTOMO_FORCE_GENERAL", "2" if variant == "17" else "0")
        eng = RayEngine(0, interp="cubic")
        monkeypatch.delenv("IONOTOMO_FORCE_GENERAL")
        eng.set_grid(xv, yv, zv)
        out = []
        for ne, freq in ((w["ne"], 100e6), (w["ne"] * 1.3, 100e6), (w["ne"] * 1.3, 140e6)):
            eng.set_values(eng.tensor(ne))
            out.append(eng.trace_fermat(eng.tensor(o), eng.tensor(d), tmax, Ns, freq, bend=True, kind="cubic", substeps=2, type=typ).cpu().numpy())
        assert not eng.check_oob()
        res[variant] = out
    for a, b in zip(res["0"], res["17"]):
        assert np.max(np.abs(a - b)) < 1e-10
    assert np.max(np.abs(res["0"][0] - res["0"][1])) > 1e-6          # the model change was seen
    field = O.n_field_tricubic(xv, yv, zv, O.ne_to_n(w["ne"], 100e6))
    ref = O.fermat_trace(o, d, tmax, Ns, field, bend=True, substeps=2, type=typ)
    assert np.max(np.abs(res["0"][0] - ref)) < 1e-9


@pytest.mark.parametrize("how", ["forced", "by_batch_size"])
def test_two_lanes_per_ray_of_the_record_tracer(how, O, monkeypatch):
    """Round 5: from 32 768 rays on the record tracer and the fused curved-ray TEC kernel give a ray TWO lanes (four nodes of the cell
    each: fermat_rhs_lmn) instead of eight -- the launch is bound by vector-instruction issue and most of what eight lanes issue is the
    same work eight times (620 000 rays: 58 -> 24 ms, profiles/r05_ab_fermat_lanes.json).  Same right-hand side to rounding: traced
    rays against the 8-lane form and the oracle's RK4, fused TEC (both integrands, both independent variables, even / odd sample
    counts, shipped behaviour bend=False too) against trace + integrate, rays that leave the tricubic domain flagged alike."""
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload(antennas="example", na=7, nd=5, nt=3, n=33, margin_cells=10)
    xv, yv, zv = w["xvec"], w["yvec"], w["zvec"]
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)          # 105 rays: not a multiple of 32 (idle groups shadow the last ray)

    def make(lanes, interp):
        monkeypatch.delenv("IONOTOMO_FERMAT_LM_LANES", raising=False)
        monkeypatch.delenv("IONOTOMO_FERMAT_LM_FEW_MIN", raising=False)
        if lanes == 2 and how == "forced":
            monkeypatch.setenv("IONOTOMO_FERMAT_LM_LANES", "2")
        elif lanes == 2:
            monkeypatch.setenv("IONOTOMO_FERMAT_LM_FEW_MIN", "64")              # "a batch that fills the chip" starts at 64 rays here
        else:
            monkeypatch.setenv("IONOTOMO_FERMAT_LM_LANES", "8")
        e = RayEngine(0, interp=interp)
        e.set_grid(xv, yv, zv)
        e.set_values(e.tensor(w["ne"]))
        return e
    for interp in ("linear", "cubic"):
        e8, e2 = make(8, interp), make(2, interp)
        o8, d8, o2, d2 = e8.tensor(o), e8.tensor(d), e2.tensor(o), e2.tensor(d)
        for typ in ("z", "s"):
            tmax = float(zv[24]) if typ == "z" else 0.7 * float(zv[24])
            for bend in (True, False):
                r8 = e8.trace_fermat(o8, d8, tmax, 21, 100e6, bend=bend, kind="cubic", substeps=2, type=typ)
                r2 = e2.trace_fermat(o2, d2, tmax, 21, 100e6, bend=bend, kind="cubic", substeps=2, type=typ)
                assert float((r8 - r2).abs().max()) < 1e-10
                for Ns in (21, 22):
                    a = e2.forward_fermat(o2, d2, tmax, Ns, 100e6, bend=bend, kind="cubic", substeps=2, type=typ, ne_scale=1e-13, fused=True)
                    b = e8.forward_fermat(o8, d8, tmax, Ns, 100e6, bend=bend, kind="cubic", substeps=2, type=typ, ne_scale=1e-13, fused=False)
                    assert float((a - b).abs().max()) < 1e-11 * float(b.abs().max()), (interp, typ, bend, Ns)
            assert not e8.check_oob() and not e2.check_oob()
        if interp == "cubic":
            field = O.n_field_tricubic(xv, yv, zv, O.ne_to_n(w["ne"], 100e6))
            ref = O.fermat_trace(o, d, float(zv[24]), 21, field, bend=True, substeps=2, type="z")
            got = e2.trace_fermat(o2, d2, float(zv[24]), 21, 100e6, bend=True, kind="cubic", substeps=2, type="z").cpu().numpy()
            assert np.max(np.abs(got - ref)) < 1e-9
        # rays that leave the domain: skipped samples + flag, the same numbers either way
        far = o.copy()
        far[::7, 0] = xv[-1] - 0.5 * (xv[1] - xv[0])
        a = e2.forward_fermat(e2.tensor(far), d2, float(zv[24]), 21, 100e6, kind="cubic", substeps=2, ne_scale=1e-13, fused=True)
        b = e8.forward_fermat(e8.tensor(far), d8, float(zv[24]), 21, 100e6, kind="cubic", substeps=2, ne_scale=1e-13, fused=True)
        assert e2.check_oob() and e8.check_oob()
        assert float((a - b).abs().max()) < 1e-11 * float(b.abs().max())


@pytest.mark.parametrize("interp", ["linear", "cubic"])
def test_transpose_on_the_record_stepper(interp, monkeypatch):
    """Round 5: large batches of bending rays through a tricubic index are BACK-PROJECTED on the record stepper too (k_fermat_tec_lm<ADJ>:
    two lanes per ray, the wave's LDS scatter window for a trilinear integrand, per-lane 3 x 3 x 3 blocks for a tricubic one) instead of
    through a ray tensor of 32 R Ns bytes.  Same numbers as trace + explicit-sample transpose, <A x, w> = <x, A^T w> against its own
    forward, zero weights and rays leaving the domain contribute nothing, and small batches keep the two-step default."""
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload(antennas="example", na=7, nd=5, nt=3, n=33, margin_cells=10)
    xv, yv, zv = w["xvec"], w["yvec"], w["zvec"]
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)          # 105 rays: the last wave has idle lane groups
    small = RayEngine(0, interp=interp)
    small.set_grid(xv, yv, zv)
    assert small.fermat_lm_ok("cubic", interp, len(o)) and not small.fermat_lm_ok("cubic", interp, len(o), transpose=True)
    assert small.fermat_lm_ok("cubic", interp, 40000, transpose=True) and not small.fermat_lm_ok("cubic", interp, 40000, transpose=True, bend=False)
    assert small._two_step_fermat(len(o), 21, "cubic", None, adjoint=True) and not small._two_step_fermat(40000, 21, "cubic", None, adjoint=True)
    monkeypatch.setenv("IONOTOMO_FERMAT_LM_FEW_MIN", "64")
    eng = RayEngine(0, interp=interp)
    eng.set_grid(xv, yv, zv)
    eng.set_values(eng.tensor(w["ne"]))
    assert eng.fermat_lm_ok("cubic", interp, len(o), transpose=True) and not eng._two_step_fermat(len(o), 21, "cubic", None, adjoint=True)
    rng = np.random.default_rng(11)
    y = rng.normal(size=len(o))
    y[::9] = 0.0
    ot, dt, yt = eng.tensor(o), eng.tensor(d), eng.tensor(y)
    for typ in ("z", "s"):
        tmax = float(zv[24]) if typ == "z" else 0.7 * float(zv[24])
        for Ns in (21, 22):
            ga = eng.adjoint_fermat(ot, dt, yt, tmax, Ns, 100e6, kind="cubic", substeps=2, type=typ, ne_scale=1e-13)                # the record stepper
            gb = eng.adjoint_fermat(ot, dt, yt, tmax, Ns, 100e6, kind="cubic", substeps=2, type=typ, ne_scale=1e-13, fused=False)   # trace + scatter
            assert float((ga - gb).abs().max()) < 1e-11 * float(gb.abs().max()), (typ, Ns)
            # <A x, w> = <x, A^T w> with the fused forward on the same stepper
            x = eng.tensor(rng.uniform(0.5, 1.5, size=w["ne"].shape) * 1e11)
            eng.set_values(x)
            ax = eng.forward_fermat(ot, dt, tmax, Ns, 100e6, kind="cubic", substeps=2, type=typ, ne_scale=1e-13)
            atw = eng.adjoint_fermat(ot, dt, yt, tmax, Ns, 100e6, kind="cubic", substeps=2, type=typ, ne_scale=1e-13)
            lhs, rhs = float((ax * yt).sum()), float((atw * x).sum())
            assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), float((ax.abs() * yt.abs()).sum())), (typ, Ns)
            eng.set_values(eng.tensor(w["ne"]))
    assert not eng.check_oob()
    # accumulates into `out`; a ray that leaves the (tricubic) domain is flagged and its outside samples add nothing, like the two-step route
    far = o.copy()
    far[::7, 0] = xv[-1] - 0.5 * (xv[1] - xv[0])
    ft = eng.tensor(far)
    base = eng.tensor(rng.normal(size=w["ne"].shape))
    ga = eng.adjoint_fermat(ft, dt, yt, float(zv[24]), 21, 100e6, kind="cubic", substeps=2, out=base.clone())
    assert eng.check_oob()
    gb = eng.adjoint_fermat(ft, dt, yt, float(zv[24]), 21, 100e6, kind="cubic", substeps=2, fused=False, out=base.clone())
    assert eng.check_oob()
    assert float((gb - base).abs().max()) > 1.0 and float((ga - gb).abs().max()) < 1e-11 * float((gb - base).abs().max())
    # a grid too small for the scatter window (fewer than 14 x 14 x 4 nodes): plain atomics, same numbers
    ws = syn.make_workload(antennas="example", na=5, nd=4, nt=2, n=11, margin_cells=3)
    monkeypatch.setenv("IONOTOMO_FERMAT_LM_FEW_MIN", "16")
    es = RayEngine(0, interp=interp)
    es.set_grid(ws["xvec"], ws["yvec"], ws["zvec"])
    es.set_values(es.tensor(ws["ne"]))
    os_, ds_ = es.tensor(ws["origins"].reshape(-1, 3)), es.tensor(ws["directions"].reshape(-1, 3))
    ys = es.tensor(rng.normal(size=os_.shape[0]))
    tm = float(ws["zvec"][7])
    assert es.fermat_lm_ok("cubic", interp, os_.shape[0], transpose=True)
    ga = es.adjoint_fermat(os_, ds_, ys, tm, 15, 100e6, kind="cubic", substeps=2, ne_scale=1e-13)
    gb = es.adjoint_fermat(os_, ds_, ys, tm, 15, 100e6, kind="cubic", substeps=2, ne_scale=1e-13, fused=False)
    assert float((ga - gb).abs().max()) < 1e-11 * float(gb.abs().max())

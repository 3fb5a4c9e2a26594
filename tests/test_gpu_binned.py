"""Node-stationary (box-binned) back-projection -- iono_adjoint_plan_dev / k_adjoint_binned -- against the C oracle and
the ray-stationary kernels: random geometries incl. steep rays that leave their box image, sample counts that are not a
multiple of the segment length, grids smaller than a box, invalid rays, the fused modes, float32 accumulation, the
tricubic channels, and the bench shape.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest
import torch

from ionotomo_amd import parallel, synthetic as syn

pytestmark = pytest.mark.gpu
SOAK = int(__import__("os").environ.get("IONO_SOAK", "1"))      # IONO_SOAK=20: twenty times the seeds (a soak run on the GPU box)


@pytest.fixture(scope="module")
def OC():
    from oracle import oracle_c
    return oracle_c


def rel(a, b):
    return np.max(np.abs(a - b)) / np.max(np.abs(b))


def engine(xv, yv, zv, **kw):
    from ionotomo_amd.engine import RayEngine
    eng = RayEngine(0, **kw)
    eng.set_grid(xv, yv, zv)
    return eng


@pytest.mark.parametrize("seed", range(SOAK * 8))
def test_binned_adjoint_random_geometries(seed, OC):
    rng = np.random.default_rng(seed)
    n = [int(v) for v in rng.integers(6, 70, 3)]
    xv, yv, zv = (np.linspace(0.0, float(rng.uniform(20, 200)), m) for m in n)
    R = int(rng.integers(1, 700))
    Ns = int(rng.choice([2, 9, 16, 17, 33, 64, 65, 100, 257]))
    steep = float(rng.choice([0.02, 0.3, 1.5]))            # 1.5: tens of cells of lateral drift per segment
    zlo, zhi = zv[0] + rng.uniform(0, 0.3) * (zv[-1] - zv[0]), zv[-1] - rng.uniform(0, 0.2) * (zv[-1] - zv[0])
    o = np.stack([rng.uniform(xv[0], xv[-1], R), rng.uniform(yv[0], yv[-1], R), np.full(R, zlo)], 1)
    d = np.stack([rng.normal(size=R) * steep, rng.normal(size=R) * steep, np.ones(R)], 1)
    end = o + d * ((zhi - zlo) / d[:, 2])[:, None]
    inside = (end[:, 0] >= xv[0]) & (end[:, 0] <= xv[-1]) & (end[:, 1] >= yv[0]) & (end[:, 1] <= yv[-1])
    if inside.sum() == 0:
        d[:, :2] = 0.0
        inside[:] = True
    eng = engine(xv, yv, zv)
    eng.set_values(eng.tensor(rng.uniform(1, 2, size=n)))
    ot, dt = eng.tensor(o), eng.tensor(d)
    w = rng.normal(size=R)
    segs, units, outside = eng.plan_adjoint(ot, dt, zhi, Ns)
    assert segs > 0 and units > 0 and outside == 0.0      # (round 5: a steep ray is cut into segments that fit their box image: nothing goes past it)
    g = eng.adjoint(ot, dt, eng.tensor(w), zhi, Ns).cpu().numpy()
    assert eng.check_oob() == (not inside.all())                               # rays leaving the grid are skipped and flagged
    gref = OC.adjoint_straight(xv, yv, zv, o[inside], d[inside], w[inside], zhi, Ns)
    assert np.max(np.abs(g - gref)) < 1e-11 * np.max(np.abs(gref)), (n, R, Ns, steep, outside)
    g32 = eng.adjoint(ot, dt, eng.tensor(w), zhi, Ns, accum=torch.float32).cpu().numpy()
    eng.check_oob()
    assert np.max(np.abs(g32 - gref)) < 2e-5 * np.max(np.abs(gref))
    eng.clear_adjoint_plan()
    g2 = eng.adjoint(ot, dt, eng.tensor(w), zhi, Ns).cpu().numpy()             # ray-stationary kernel, same answer
    eng.check_oob()
    assert np.max(np.abs(g2 - gref)) < 1e-11 * np.max(np.abs(gref))


def test_binned_adjoint_bench_shape_and_fused_modes(OC):
    import bench
    w = bench.build_workload(0)
    eng = engine(w["xvec"], w["yvec"], w["zvec"])
    x = np.exp(w["m"])
    eng.set_values(eng.tensor(x))
    na, P = bench.NA, bench.NT * bench.ND
    rng = np.random.default_rng(0)
    prob = parallel.ShardedRays(eng, w["origins"].reshape(na, P, 3), w["directions"].reshape(na, P, 3), bench.TMAX, bench.NS,
                                dobs=rng.normal(size=(na, P)) * 0.1, cdct=rng.uniform(0.5, 2.0, size=(na, P)), i0=3, tune=False,
                                plan=False)
    y = eng.tensor(rng.normal(size=na * P))
    ref = eng.adjoint(prob.origins, prob.dirs, y, bench.TMAX, bench.NS, order=prob.order)          # ray-stationary
    tec = prob.forward_tec()
    ref1 = prob.gradient_from_tec(tec)
    sc = eng.tensor(rng.uniform(0.5, 2.0, size=na * P))
    ref2 = eng.adjoint_differential(prob.origins, prob.dirs, y, sc, na, 3, bench.TMAX, bench.NS, order=prob.order)
    segs, units, outside = eng.plan_adjoint(prob.origins, prob.dirs, bench.TMAX, bench.NS)
    assert segs >= na * P * (bench.NS // 16) and outside < 0.05
    got = eng.adjoint(prob.origins, prob.dirs, y, bench.TMAX, bench.NS)
    assert float((got - ref).abs().max()) < 1e-11 * float(ref.abs().max())
    lhs, rhs = float(torch.dot(tec, y)), float((got * eng.tensor(x)).sum())
    assert abs(lhs - rhs) < 1e-10 * float(tec.norm()) * float(y.norm())
    got1 = prob.gradient_from_tec(tec)
    assert float((got1 - ref1).abs().max()) < 1e-11 * float(ref1.abs().max())
    got2 = eng.adjoint_differential(prob.origins, prob.dirs, y, sc, na, 3, bench.TMAX, bench.NS)
    assert float((got2 - ref2).abs().max()) < 1e-11 * float(ref2.abs().max())
    # a 500-ray sample against the C oracle
    idx = np.sort(rng.choice(na * P, 500, replace=False))
    ys = torch.zeros_like(y)
    ys[idx] = y[idx]
    gs = eng.adjoint(prob.origins, prob.dirs, ys, bench.TMAX, bench.NS).cpu().numpy()
    gref = OC.adjoint_straight(w["xvec"], w["yvec"], w["zvec"], w["origins"][idx], w["directions"][idx], y.cpu().numpy()[idx],
                               bench.TMAX, bench.NS)
    assert np.max(np.abs(gs - gref)) < 1e-11 * np.max(np.abs(gref))
    assert not eng.check_oob()


@pytest.mark.parametrize("lanes", [4, 8, 16])
@pytest.mark.parametrize("Ns", [17, 64, 257])
def test_binned_segment_widths(lanes, Ns, OC, monkeypatch):
    """The plan packs a segment into 4, 8 or 16 lanes (IONOTOMO_SEG_LANES forces one; without it the width with the fewest empty
    lanes is chosen per geometry): the trilinear, tricubic and phase back-projections are the same sums at every width --
    also where a DPP row mixes segments of different rays (the z-neighbour merge is decided on the node, not on the ray)."""
    monkeypatch.setenv("IONOTOMO_SEG_LANES", str(lanes))
    w = syn.make_workload(antennas="lofar", na=20, nd=5, nt=3, n=72)
    xv, yv, zv = w["xvec"], w["yvec"], w["zvec"]
    M = w["ne"] / 1e13
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    rng = np.random.default_rng(lanes * 1000 + Ns)
    y = rng.normal(size=len(o))
    eng = engine(xv, yv, zv)
    eng.set_values(eng.tensor(M))
    ot, dt = eng.tensor(o), eng.tensor(d)
    segs, units, _ = eng.plan_adjoint(ot, dt, w["tmax"], Ns)
    assert eng.plan_segment_lanes() == lanes and segs >= len(o) * ((Ns + lanes - 1) // lanes)
    g = eng.adjoint(ot, dt, eng.tensor(y), w["tmax"], Ns).cpu().numpy()
    gref = OC.adjoint_straight(xv, yv, zv, o, d, y, w["tmax"], Ns)
    assert np.max(np.abs(g - gref)) < 1e-11 * np.max(np.abs(gref))
    # phase transpose: planned against the ray-stationary kernel
    freqs = np.array([120e6, 150e6, 180e6])
    eng.set_values(eng.tensor(w["ne"]))
    yp = eng.tensor(rng.normal(size=(20, 15, 3)))
    gp = eng.adjoint_phase(ot, dt, yp, 20, w["tmax"], Ns, freqs, 2)
    eng.clear_adjoint_plan()
    assert eng.plan_segment_lanes() == 0
    gq = eng.adjoint_phase(ot, dt, yp, 20, w["tmax"], Ns, freqs, 2)
    assert float((gp - gq).abs().max()) < 1e-11 * float(gq.abs().max())
    # tricubic channels
    engc = engine(xv, yv, zv, interp="cubic")
    engc.set_values(engc.tensor(M))
    oc, dc = engc.tensor(o), engc.tensor(d)
    gu = engc.adjoint(oc, dc, engc.tensor(y), w["tmax"], Ns)
    engc.plan_adjoint(oc, dc, w["tmax"], Ns)
    assert engc.plan_segment_lanes() == lanes
    gc = engc.adjoint(oc, dc, engc.tensor(y), w["tmax"], Ns)
    assert float((gc - gu).abs().max()) < 1e-11 * float(gu.abs().max())
    assert not eng.check_oob() and not engc.check_oob()


def test_binned_segment_width_is_chosen_per_geometry():
    """65 samples through 256 cells leave 3.8 samples in a 15-cell layer: 4 lanes; one sample per cell (the bench shape) and 257
    samples through 48 cells fill 16."""
    import bench
    w = bench.build_workload(0)
    eng = engine(w["xvec"], w["yvec"], w["zvec"])
    na, P = bench.NA, bench.NT * bench.ND
    ot, dt = eng.tensor(w["origins"].reshape(-1, 3)[: 40 * P]), eng.tensor(w["directions"].reshape(-1, 3)[: 40 * P])
    eng.plan_adjoint(ot, dt, bench.TMAX, 65)
    assert eng.plan_segment_lanes() == 4
    eng.plan_adjoint(ot, dt, bench.TMAX, 129)
    assert eng.plan_segment_lanes() == 8
    eng.plan_adjoint(ot, dt, bench.TMAX, bench.NS)
    assert eng.plan_segment_lanes() == 16
    v = syn.make_workload(antennas="lofar", na=62, nd=6, nt=4, n=48)
    eng2 = engine(v["xvec"], v["yvec"], v["zvec"])
    o2, d2 = eng2.tensor(v["origins"].reshape(-1, 3)), eng2.tensor(v["directions"].reshape(-1, 3))
    eng2.plan_adjoint(o2, d2, v["tmax"], 257)
    assert eng2.plan_segment_lanes() == 16


def test_binned_tricubic_channels():
    from oracle import oracle as O
    w = syn.make_workload("cfg2")
    eng = engine(w["xvec"], w["yvec"], w["zvec"], interp="cubic")
    eng.set_values(eng.tensor(w["ne"] / 1e13))
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    idx = np.sort(np.random.default_rng(2).choice(len(o), 300, replace=False))
    ot, dt = eng.tensor(o[idx]), eng.tensor(d[idx])
    y = np.random.default_rng(3).normal(size=len(idx))
    rays = O.straight_rays(o[idx], d[idx], w["tmax"], 129)
    ref = O.adjoint_tec(rays, w["xvec"], w["yvec"], w["zvec"], y, kind=O.INTERP_TRICUBIC)
    eng.plan_adjoint(ot, dt, w["tmax"], 129)
    g = eng.adjoint(ot, dt, eng.tensor(y), w["tmax"], 129).cpu().numpy()
    assert rel(g, ref) < 1e-11
    assert not eng.check_oob()


def test_pending_unit_range_is_refused_by_every_launch_but_the_planned_trilinear_one(monkeypatch):
    """iono_adjoint_unit_range (one z-slab of the plan) means something to the planned trilinear back-projection only.  Any other
    launch used to DISCARD the range and add the whole back-projection -- once per slab under exchange="overlap" (ADVICE r4): now
    IONO_ERR_ARG, nothing launched, and the range does not linger."""
    w = syn.make_workload(antennas="lofar", na=12, nd=5, nt=3, n=40)
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    R = len(o)
    for interp, replan, variant in (("cubic", False, None), ("linear", True, None), ("linear", False, "2")):
        if variant:
            monkeypatch.setenv("IONOTOMO_VARIANT", variant)
        eng = engine(w["xvec"], w["yvec"], w["zvec"], interp=interp)
        monkeypatch.delenv("IONOTOMO_VARIANT", raising=False)
        eng.set_values(eng.tensor(w["ne"] / 1e13))
        ot, dt = eng.tensor(o), eng.tensor(d)
        eng.plan_adjoint(ot, dt, w["tmax"], 41, slabs=4)
        r = eng.tensor(np.random.default_rng(0).normal(size=R))
        scale = torch.ones_like(r)
        # a ray pass that leaves its weights in the library (out=None), as the solvers' overlapped steps do
        eng.adjoint_sirt_step(ot, dt, r, torch.zeros_like(r), scale, scale, 12, 0, w["tmax"], 41, None, want_dot=False)
        if replan:                                                 # the engine's single plan now belongs to other tensors
            o2, d2 = ot.clone(), dt.clone()
            eng.plan_adjoint(o2, d2, w["tmax"], 41)
        out = torch.zeros(eng.shape, dtype=torch.float64, device=eng.device)
        with pytest.raises(ValueError, match="work-unit range"):
            eng.adjoint_planned_weights(ot, dt, w["tmax"], 41, out, unit_range=(0, 1))
        assert float(out.abs().max()) == 0.0                       # nothing launched
        full = eng.adjoint_planned_weights(ot, dt, w["tmax"], 41, out)          # no range pending any more: the whole back-projection
        assert float(full.abs().max()) > 0.0

"""Independent solves stacked along x (ionotomo_amd/inversion/parallel_solves.py, the counterpart of the reference pipeline's
``num_parallel_solves``: inversion/inversion_pipeline.py:41-50,131-216): every launch of the stacked problem must give, block by
block, what the solves give one at a time -- forward TEC against the C oracle and against per-solve engines, the exact transpose,
and SIRT iterates.  Needs a real MI355X: -m gpu."""
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


def solves(B, n=40, na=12, nd=9, seed=0, same_geometry=False):
    """B single-time-step solves in the shape the pipeline forms them: the same array and facet directions seen at B times (the
    field rotates about the pole), every solve with its own domain (same shape and spacing, its own origin) and its own model."""
    rng = np.random.default_rng(seed)
    ants = syn.example_antennas_km(na, 3)
    dirs = syn.rotate_about_pole(syn.facet_directions(nd, 4.0, 1), 1 if same_geometry else B)
    out = []
    tmax = 600.0
    o0, d0 = syn.ray_bundle(ants, dirs[:1])
    xv, yv, zv = syn.domain_for(o0, d0, n, 590.0, 5)
    for b in range(B):
        o, d = syn.ray_bundle(ants, dirs[0:1] if same_geometry else dirs[b:b + 1])
        o, d = o.reshape(na, nd, 3), d.reshape(na, nd, 3)
        # the same spacing, another origin (a pipeline builds every time step's domain from that step's rays)
        off = np.zeros(3) if same_geometry else rng.uniform(-3, 3, 3) * np.array([xv[1] - xv[0], yv[1] - yv[0], 0.0])
        g = (xv + off[0], yv + off[1], zv + off[2])
        ne = syn.ne_model(*g, seed=100 + b, corr=60.0)
        out.append(dict(grid=g, ne=ne, o=o + off, d=d))
    return out, tmax, n + 1


def one_by_one(sv, tmax, Ns, fn):
    from ionotomo_amd.engine import RayEngine
    res = []
    for s in sv:
        e = RayEngine(0)
        e.set_grid(*s["grid"])
        res.append(fn(e, s))
    return res


def test_stacked_forward_and_transpose_equal_the_separate_solves(OC):
    from ionotomo_amd.inversion.parallel_solves import StackedSolves
    sv, tmax, Ns = solves(5)
    sv[1]["o"], sv[1]["d"] = sv[1]["o"][:, :4], sv[1]["d"][:, :4]          # ragged: the solves need not hold the same number of pairs
    sv[3]["o"], sv[3]["d"] = sv[3]["o"][:, 2:3], sv[3]["d"][:, 2:3]        # ... down to a single direction
    st = StackedSolves([s["grid"] for s in sv])
    o, d = st.rays([s["o"] for s in sv], [s["d"] for s in sv], tmax)
    assert st.pairs == [9, 4, 9, 1, 9]
    eng = st.engine
    Na = o.shape[0]
    eng.set_values(st.stack_grids([s["ne"] for s in sv]).reshape(-1))
    ot, dt = eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3))
    tec = eng.forward(ot, dt, tmax, Ns)
    assert not eng.check_oob()
    parts = st.split_rays(tec, Na)
    rng = np.random.default_rng(5)
    w = [rng.normal(size=(Na, s["o"].shape[1])) for s in sv]
    g = eng.adjoint(ot, dt, eng.tensor(st.stack_rays(w).reshape(-1)), tmax, Ns)
    gparts = st.split_grid(g)

    def single(e, s):
        k = len(single.done)
        e.set_values(e.tensor(s["ne"]).reshape(-1))
        o1, d1 = e.tensor(s["o"].reshape(-1, 3)), e.tensor(s["d"].reshape(-1, 3))
        t1 = e.forward(o1, d1, tmax, Ns).clone()
        g1 = e.adjoint(o1, d1, e.tensor(w[k].reshape(-1)), tmax, Ns).clone()
        single.done.append(k)
        return t1, g1
    single.done = []
    ref = one_by_one(sv, tmax, Ns, single)
    for b, s in enumerate(sv):
        t1, g1 = ref[b]
        oc = OC.forward_tec_straight(*s["grid"], s["ne"], s["o"].reshape(-1, 3), s["d"].reshape(-1, 3), tmax, Ns)
        got = parts[b].reshape(-1).cpu().numpy()
        # the oracle on the solve's OWN grid and rays (float64: the moved coordinates round differently, nothing else differs)
        assert np.max(np.abs(got - oc) / np.abs(oc)) < 1e-11, b
        assert float((parts[b].reshape(-1) - t1).abs().max() / t1.abs().max()) < 1e-12, b
        assert float((gparts[b] - g1.reshape(gparts[b].shape)).abs().max() / g1.abs().max()) < 1e-11, b
    # nothing of a solve lands in another one's block: a back-projection of ONE solve's weights leaves the other blocks zero
    w1 = [np.zeros_like(a) for a in w]
    w1[2] = w[2]
    g2 = st.split_grid(eng.adjoint(ot, dt, eng.tensor(st.stack_rays(w1).reshape(-1)), tmax, Ns))
    for b in range(len(sv)):
        if b != 2:
            assert float(g2[b].abs().max()) == 0.0, b
    assert float((g2[2] - gparts[2]).abs().max()) <= 1e-12 * float(gparts[2].abs().max())
    # a stack of ONE solve is that solve
    s1 = StackedSolves([sv[4]["grid"]])
    o1, d1 = s1.rays([sv[4]["o"]], [sv[4]["d"]], tmax)
    assert np.array_equal(o1, sv[4]["o"]) and np.array_equal(s1.xvec, sv[4]["grid"][0])
    s1.engine.set_values(s1.stack_grids([sv[4]["ne"]]).reshape(-1))
    t4 = s1.engine.forward(s1.engine.tensor(o1.reshape(-1, 3)), s1.engine.tensor(d1.reshape(-1, 3)), tmax, Ns)
    assert torch.equal(t4, ref[4][0])


def test_stacked_planned_kernels_equal_the_unplanned_ones():
    """The bundle plan and the back-projection plan are built on the stacked geometry like on any other."""
    from ionotomo_amd.inversion.parallel_solves import StackedSolves
    sv, tmax, Ns = solves(8, n=48, na=20, nd=12, seed=2)
    st = StackedSolves([s["grid"] for s in sv])
    o, d = st.rays([s["o"] for s in sv], [s["d"] for s in sv], tmax)
    eng = st.engine
    eng.set_values(st.stack_grids([s["ne"] for s in sv]).reshape(-1))
    ot, dt = eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3))
    t0 = eng.forward(ot, dt, tmax, Ns).clone()
    y = torch.randn(ot.shape[0], dtype=torch.float64, device=eng.device)
    g0 = eng.adjoint(ot, dt, y, tmax, Ns).clone()
    eng.plan_forward(ot, dt, tmax, Ns)
    eng.plan_adjoint(ot, dt, tmax, Ns)
    t1 = eng.forward(ot, dt, tmax, Ns)
    g1 = eng.adjoint(ot, dt, y, tmax, Ns)
    assert float((t1 - t0).abs().max() / t0.abs().max()) < 1e-13
    assert float((g1 - g0).abs().max() / g0.abs().max()) < 1e-11
    eng.check_plans()


def test_stacked_sirt_is_the_separate_sirt_solves():
    from ionotomo_amd import parallel, solvers
    from ionotomo_amd.engine import RayEngine
    from ionotomo_amd.inversion.parallel_solves import StackedSolves
    B = 4
    sv, tmax, Ns = solves(B, same_geometry=True)            # one geometry, B models and data sets (see the module docstring: the
    st = StackedSolves(sv[0]["grid"], count=B)              # one shared number of SIRT is the column cut-off relative to max(col))
    o, d = st.rays([s["o"] for s in sv], [s["d"] for s in sv], tmax)
    eng = st.engine
    Na, P = sv[0]["o"].shape[:2]
    x0 = [torch.as_tensor(syn.ne_model(*s["grid"], turbulent=False) / 1e11) for s in sv]
    dobs, cdct = [], []
    for b, s in enumerate(sv):
        e = RayEngine(0)
        e.set_grid(*s["grid"])
        e.set_values(e.tensor(s["ne"] / 1e11).reshape(-1))
        t = e.forward(e.tensor(s["o"].reshape(-1, 3)), e.tensor(s["d"].reshape(-1, 3)), tmax, Ns).reshape(Na, P)
        dobs.append((t - t[0:1]).cpu().numpy())
        cdct.append(np.full((Na, P), 1e-4 * (1 + b)))
    prob = parallel.ShardedRays(eng, o, d, tmax, Ns, dobs=st.stack_rays(dobs), cdct=st.stack_rays(cdct), i0=0, tune=False)
    xs, hist = solvers.sirt(prob, st.stack_grids(x0), n_iter=6)
    blocks = st.split_grid(xs)
    total = 0.0
    for b, s in enumerate(sv):
        e = RayEngine(0)
        e.set_grid(*s["grid"])
        p1 = parallel.ShardedRays(e, s["o"], s["d"], tmax, Ns, dobs=dobs[b], cdct=cdct[b], i0=0, tune=False)
        x1, h1 = solvers.sirt(p1, x0[b].to(e.device), n_iter=6)
        assert float((blocks[b] - x1).abs().max() / x1.abs().max()) < 1e-10, b
        total = total + np.asarray(h1, dtype=np.float64)
        assert float((x1 - x0[b].to(e.device)).abs().max()) > 0          # (the solve moved)
    # the stacked objective is the sum of the solves' objectives, iteration by iteration
    np.testing.assert_allclose(np.asarray(hist, dtype=np.float64), total, rtol=1e-10)


def test_stacked_tricubic_equals_the_separate_solves():
    """interp="cubic": rays two cells clear of the slab's x faces see the same Lekien-Marsden interpolant as in their own grid."""
    from ionotomo_amd.engine import RayEngine
    from ionotomo_amd.inversion.parallel_solves import StackedSolves
    sv, tmax, Ns = solves(3, n=36, na=10, nd=7, seed=4)
    st = StackedSolves([s["grid"] for s in sv], interp="cubic")
    o, d = st.rays([s["o"] for s in sv], [s["d"] for s in sv], tmax)
    eng = st.engine
    Na = o.shape[0]
    eng.set_values(st.stack_grids([s["ne"] for s in sv]).reshape(-1))
    ot, dt = eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3))
    parts = st.split_rays(eng.forward(ot, dt, tmax, Ns), Na)
    rng = np.random.default_rng(9)
    w = [rng.normal(size=(Na, s["o"].shape[1])) for s in sv]
    gparts = st.split_grid(eng.adjoint(ot, dt, eng.tensor(st.stack_rays(w).reshape(-1)), tmax, Ns))
    for b, s in enumerate(sv):
        e = RayEngine(0, interp="cubic")
        e.set_grid(*s["grid"])
        e.set_values(e.tensor(s["ne"]).reshape(-1))
        o1, d1 = e.tensor(s["o"].reshape(-1, 3)), e.tensor(s["d"].reshape(-1, 3))
        t1 = e.forward(o1, d1, tmax, Ns)
        assert float((parts[b].reshape(-1) - t1).abs().max() / t1.abs().max()) < 1e-12, b
        g1 = e.adjoint(o1, d1, e.tensor(w[b].reshape(-1)), tmax, Ns).reshape(gparts[b].shape)
        assert float((gparts[b] - g1).abs().max() / g1.abs().max()) < 1e-11, b
    # a ray inside the slab but within two cells of its face is refused for the tricubic (the trilinear stack takes it)
    s0 = sv[0]
    o2 = s0["o"].copy()
    o2[0, 0, 0] = s0["grid"][0][0] + 0.5 * (s0["grid"][0][1] - s0["grid"][0][0])
    d2 = s0["d"].copy()
    d2[0, 0] = [0.0, 0.0, 1.0]
    with pytest.raises(ValueError, match="two cells"):
        st.rays([o2] + [s["o"] for s in sv[1:]], [d2] + [s["d"] for s in sv[1:]], tmax)
    StackedSolves([s["grid"] for s in sv]).rays([o2] + [s["o"] for s in sv[1:]], [d2] + [s["d"] for s in sv[1:]], tmax)


def test_stacked_float32_fast_mode(OC):
    """storage="f32" + a plan on the stacked geometry: k_forward_bundle_f32 for every bundle (IONOTOMO_HYBRID_MIN=1), each solve's TEC
    within the fast mode's 1e-6 of the float64 oracle on its own grid (north_star's tolerance for float32 storage)."""
    import os
    from ionotomo_amd.inversion.parallel_solves import StackedSolves
    sv, tmax, Ns = solves(6, n=48, na=24, nd=10, seed=6)
    st = StackedSolves([s["grid"] for s in sv], storage="f32")
    o, d = st.rays([s["o"] for s in sv], [s["d"] for s in sv], tmax)
    old = os.environ.get("IONOTOMO_HYBRID_MIN")
    os.environ["IONOTOMO_HYBRID_MIN"] = "1"
    try:
        eng = st.engine
    finally:
        if old is None:
            os.environ.pop("IONOTOMO_HYBRID_MIN", None)
        else:
            os.environ["IONOTOMO_HYBRID_MIN"] = old
    eng.set_values(st.stack_grids([s["ne"] for s in sv]).reshape(-1))
    ot, dt = eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3))
    info = eng.plan_forward(ot, dt, tmax, Ns)
    assert info[0] > 0 and eng.describe("forward", ot, dt, tmax, Ns)[0].startswith("k_forward_bundle_f32")
    parts = st.split_rays(eng.forward(ot, dt, tmax, Ns), o.shape[0])
    assert not eng.check_oob()
    for b, s in enumerate(sv):
        oc = OC.forward_tec_straight(*s["grid"], s["ne"], s["o"].reshape(-1, 3), s["d"].reshape(-1, 3), tmax, Ns)
        assert np.max(np.abs(parts[b].reshape(-1).cpu().numpy() - oc) / np.abs(oc)) < 1e-6, b


def test_stacked_cgls_with_per_solve_step_lengths_is_the_separate_cgls_solves():
    """StackedSolves.cgls: one alpha and one beta per solve -- block b's iterates and objective history are those of solve b alone,
    for solves of DIFFERENT geometry and weights (solvers.cgls on the stacked problem is one solve of the block system instead)."""
    from ionotomo_amd import parallel, solvers
    from ionotomo_amd.engine import RayEngine
    from ionotomo_amd.inversion.parallel_solves import StackedSolves
    B = 4
    sv, tmax, Ns = solves(B, seed=11)
    st = StackedSolves([s["grid"] for s in sv])
    o, d = st.rays([s["o"] for s in sv], [s["d"] for s in sv], tmax)
    Na, P = sv[0]["o"].shape[:2]
    x0 = [torch.as_tensor(syn.ne_model(*s["grid"], turbulent=False) / 1e11) for s in sv]
    dobs, cdct = [], []
    for b, s in enumerate(sv):
        e = RayEngine(0)
        e.set_grid(*s["grid"])
        e.set_values(e.tensor(s["ne"] / 1e11).reshape(-1))
        t = e.forward(e.tensor(s["o"].reshape(-1, 3)), e.tensor(s["d"].reshape(-1, 3)), tmax, Ns).reshape(Na, P)
        dobs.append((t - t[0:1]).cpu().numpy())
        cdct.append(np.full((Na, P), 1e-4 * 3 ** b))                # weights three times apart from solve to solve
    prob = parallel.ShardedRays(st.engine, o, d, tmax, Ns, dobs=st.stack_rays(dobs), cdct=st.stack_rays(cdct), i0=0, tune=False)
    xs, hist = st.cgls(prob, st.stack_grids(x0), n_iter=8)
    assert hist.shape == (8, B)
    blocks = st.split_grid(xs)
    shared, _ = solvers.cgls(prob, st.stack_grids(x0), n_iter=8)
    differs = 0
    for b, s in enumerate(sv):
        e = RayEngine(0)
        e.set_grid(*s["grid"])
        p1 = parallel.ShardedRays(e, s["o"], s["d"], tmax, Ns, dobs=dobs[b], cdct=cdct[b], i0=0, tune=False)
        x1, h1 = solvers.cgls(p1, x0[b].to(e.device), n_iter=8)
        scale = float((x1 - x0[b].to(e.device)).abs().max())
        assert scale > 0
        assert float((blocks[b] - x1).abs().max()) < 1e-7 * scale, b        # (8 CG steps amplify the summation order: solvers' own tests)
        np.testing.assert_allclose(hist[:, b], np.asarray(h1, dtype=np.float64)[:8], rtol=1e-7)
        differs += float((st.split_grid(shared)[b] - x1).abs().max()) > 1e-3 * scale
    assert differs >= B - 1          # the shared-scalar solve takes other steps (what this method is for)


@pytest.mark.parametrize("seed", range(SOAK * 3))
def test_stacked_random_solves(seed, OC):
    """Random numbers of solves, grid shapes, arrays and pair counts, with and without plans: block b of the stacked forward equals the
    C oracle on solve b's own grid, the stacked transpose is the transpose of the stacked forward (dot-product test) and leaves
    every other solve's block untouched."""
    from ionotomo_amd.inversion.parallel_solves import StackedSolves
    rng = np.random.default_rng(900 + seed)
    B = int(rng.integers(2, 9))
    sv, tmax, Ns = solves(B, n=int(rng.integers(20, 56)), na=int(rng.integers(3, 24)), nd=int(rng.integers(2, 14)), seed=seed)
    for s_ in sv:                                                           # ragged pair counts
        keep = int(rng.integers(1, s_["o"].shape[1] + 1))
        s_["o"], s_["d"] = s_["o"][:, :keep], s_["d"][:, :keep]
    st = StackedSolves([s_["grid"] for s_ in sv])
    o, d = st.rays([s_["o"] for s_ in sv], [s_["d"] for s_ in sv], tmax)
    eng = st.engine
    Na = o.shape[0]
    x = st.stack_grids([s_["ne"] for s_ in sv])
    eng.set_values(x.reshape(-1))
    ot, dt = eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3))
    if seed % 2:
        eng.plan_forward(ot, dt, tmax, Ns)
        eng.plan_adjoint(ot, dt, tmax, Ns)
    tec = eng.forward(ot, dt, tmax, Ns)
    assert not eng.check_oob()
    for b, (part, s_) in enumerate(zip(st.split_rays(tec, Na), sv)):
        oc = OC.forward_tec_straight(*s_["grid"], s_["ne"], s_["o"].reshape(-1, 3), s_["d"].reshape(-1, 3), tmax, Ns)
        assert np.max(np.abs(part.reshape(-1).cpu().numpy() - oc) / np.abs(oc)) < 1e-11, (seed, b)
    y = torch.as_tensor(rng.normal(size=tec.shape[0])).to(eng.device)
    g = eng.adjoint(ot, dt, y, tmax, Ns)
    lhs, rhs = float(torch.dot(tec, y)), float(torch.dot(x.reshape(-1), g.reshape(-1)))
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs), float(tec.abs().max() * y.abs().max())), (seed, lhs, rhs)
    k = int(rng.integers(0, B))
    yk = torch.zeros_like(y).reshape(Na, -1)
    lo = sum(st.pairs[:k])
    yk[:, lo:lo + st.pairs[k]] = y.reshape(Na, -1)[:, lo:lo + st.pairs[k]]
    gk = st.split_grid(eng.adjoint(ot, dt, yk.reshape(-1).contiguous(), tmax, Ns))
    for b in range(B):
        if b != k:
            assert float(gk[b].abs().max()) == 0.0, (seed, b, k)
    assert float((gk[k] - st.split_grid(g)[k]).abs().max()) <= 1e-11 * float(g.abs().max())

"""Device-resident engine, fused residual adjoint, solvers and size-independent properties at
BASELINE sizes.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest
import torch

from ionotomo_amd import parallel, solvers, synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def make_engine(w, storage="f64"):
    from ionotomo_amd.engine import RayEngine
    eng = RayEngine(0, storage=storage)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    return eng


def test_engine_forward_adjoint_vs_oracle(O):
    w = syn.make_workload("cfg1")
    eng = make_engine(w)
    eng.set_log_model(eng.tensor(w["m"]), w["K_ne"] / 1e13)
    o, d = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
    tec = eng.forward(o, d, w["tmax"], w["Ns"]).cpu().numpy()
    rays = O.straight_rays(w["origins"], w["directions"], w["tmax"], w["Ns"])
    ref = O.forward_tec(rays, w["xvec"], w["yvec"], w["zvec"], O.ne_from_log_model(w["m"], w["K_ne"]))
    assert np.max(np.abs(tec - ref.ravel()) / np.abs(ref.ravel())) < 1e-12
    assert not eng.check_oob()
    y = np.random.default_rng(0).normal(size=ref.shape)
    g = eng.adjoint(o, d, eng.tensor(y.ravel()), w["tmax"], w["Ns"]).cpu().numpy()
    gref = O.adjoint_tec(rays, w["xvec"], w["yvec"], w["zvec"], y)
    assert np.max(np.abs(g - gref)) < 1e-11 * np.max(np.abs(gref))
    g32 = eng.adjoint(o, d, eng.tensor(y.ravel()), w["tmax"], w["Ns"], accum=torch.float32).cpu().numpy()
    assert np.max(np.abs(g32 - gref)) < 1e-5 * np.max(np.abs(gref))          # f32 atomics (SURVEY 8d gate)
    # out-of-grid rays set the sticky flag instead of faulting
    eng.forward(o, d, w["zvec"][-1] + 100.0, w["Ns"])
    assert eng.check_oob() and not eng.check_oob()


def test_fused_residual_adjoint_equals_separate_steps(O):
    w = syn.make_workload(antennas="example", na=6, nd=5, nt=3, n=20)
    na, P = 6, 15
    o = w["origins"].reshape(na, P, 3)
    d = w["directions"].reshape(na, P, 3)
    eng = make_engine(w)
    eng.set_values(eng.tensor(w["ne"] / 1e13))
    rng = np.random.default_rng(4)
    dobs = rng.normal(size=(na, P)) * 0.1
    cdct = rng.uniform(0.5, 2.0, size=(na, P))
    for i0 in (0, 3, 5):
        prob = parallel.ShardedRays(eng, o, d, w["tmax"], 21, dobs=dobs, cdct=cdct, i0=i0)
        tec = prob.forward_tec()
        fused = prob.gradient_from_tec(tec).cpu().numpy()
        t = tec.cpu().numpy().reshape(na, P)
        dd = (t - t[i0] - dobs) / (cdct + 1e-15)
        rays = O.straight_rays(o, d, w["tmax"], 21)
        ref = O.adjoint_tec(rays, w["xvec"], w["yvec"], w["zvec"], O.differential_weights(dd, i0))
        assert np.max(np.abs(fused - ref)) < 1e-11 * np.max(np.abs(ref))
        sep = prob.adjoint(eng.tensor(dd.ravel())).cpu().numpy()
        assert np.max(np.abs(sep - ref)) < 1e-11 * np.max(np.abs(ref))


def test_solvers_on_gpu_match_dense_restatement():
    from oracle import oracle as Or, solvers as OS
    from problems import small_problem
    pb = small_problem()
    w = pb["w"]
    rays = Or.straight_rays(pb["o"], pb["d"], pb["tmax"], pb["Ns"])
    G, A = OS.dense_operator(rays, w["xvec"], w["yvec"], w["zvec"], pb["i0"])
    d = A @ pb["x_true"].ravel() + pb["rng"].normal(size=A.shape[0]) * 1e-3
    cd = np.full(A.shape[0], 1e-6)
    eng = make_engine(w)
    prob = parallel.ShardedRays(eng, pb["o"], pb["d"], pb["tmax"], pb["Ns"], dobs=d.reshape(pb["na"], pb["P"]),
                                cdct=cd.reshape(pb["na"], pb["P"]), i0=pb["i0"])
    x0 = eng.tensor(pb["x0"])
    xs, hs = solvers.sirt(prob, x0, n_iter=10)
    xr, hr = OS.sirt(G, A, d, cd, pb["x0"].ravel(), pb["na"], pb["P"], pb["i0"], 10)
    assert np.allclose(hs, hr, rtol=1e-9)
    assert np.max(np.abs(xs.cpu().numpy().ravel() - xr)) < 1e-9 * np.max(np.abs(xr))    # relative to the iterate's scale
    xc, hc = solvers.cgls(prob, x0, n_iter=10)
    xr, hr = OS.cgls(A, d, cd, pb["x0"].ravel(), 10)
    # CG amplifies rounding differences between the two forward implementations (1e-14 vs 1e-16):
    # compare the objective history tightly and the iterate relative to its scale
    assert np.allclose(hc, hr, rtol=1e-6)
    assert np.max(np.abs(xc.cpu().numpy().ravel() - xr)) < 1e-4 * np.max(np.abs(xr))
    K = float(np.median(pb["x0"]))
    mm, hm = solvers.steepest_descent_log_model(prob, eng.tensor(np.log(pb["x0"] / K)), K, max_iter=8)
    mr, hr = OS.steepest_descent_log_model(A, d, cd, np.log(pb["x0"] / K).ravel(), K, max_iter=8)
    assert len(hm) == len(hr) and np.allclose(hm, hr, rtol=1e-7)
    assert np.max(np.abs(mm.cpu().numpy().ravel() - mr)) < 1e-7 * np.max(np.abs(mr))
    # the same with the C_m-smoothed direction
    from ionotomo_amd.ionosphere.covariance import Covariance
    cov = Covariance(dx=w["xvec"][1] - w["xvec"][0], dy=w["yvec"][1] - w["yvec"][0], dz=w["zvec"][1] - w["zvec"][0], l=40.0)
    shape = pb["x0"].shape
    ms, hs2 = solvers.steepest_descent_log_model(prob, eng.tensor(np.log(pb["x0"] / K)), K, max_iter=6, covariance=cov)
    mr2, hr2 = OS.steepest_descent_log_model(A, d, cd, np.log(pb["x0"] / K).ravel(), K, max_iter=6,
                                             smooth=lambda v: Or.smooth(v.reshape(shape), cov.dx, cov.dy, cov.dz, l=40.0).ravel())
    assert len(hs2) == len(hr2) and np.allclose(hs2, hr2, rtol=1e-7)
    assert np.max(np.abs(ms.cpu().numpy().ravel() - mr2)) < 1e-7 * np.max(np.abs(mr2))


def test_full_size_properties_256_cubed():
    """BASELINE full size (256^3 grid, Ns = 257, 62 x 42 x 8 rays): size-independent properties
    instead of the (slow) oracle -- linearity, exactness on a linear field, dot-product test."""
    import bench
    w = bench.build_workload(0)
    sel = np.arange(62 * 100 * 42).reshape(62, 100, 42)[:, :8, :].ravel()
    eng = make_engine(w)
    o, d = eng.tensor(w["origins"][sel]), eng.tensor(w["directions"][sel])
    X, Y, Z = np.meshgrid(w["xvec"], w["yvec"], w["zvec"], indexing="ij")
    # (1) a field linear in x,y,z is reproduced exactly by trilinear interpolation and Simpson:
    #     TEC = path length * field at the ray midpoint
    lin = 2.0 + 0.01 * X - 0.02 * Y + 0.003 * Z
    eng.set_values(eng.tensor(lin))
    tec = eng.forward(o, d, 1000.0, 257).cpu().numpy()
    oo, dd = w["origins"][sel], w["directions"][sel]
    p = dd / np.linalg.norm(dd, axis=1, keepdims=True)
    L = (1000.0 - oo[:, 2]) / p[:, 2]
    mid = oo + p * (L / 2)[:, None]
    exact = L * (2.0 + 0.01 * mid[:, 0] - 0.02 * mid[:, 1] + 0.003 * mid[:, 2])
    assert np.max(np.abs(tec - exact) / np.abs(exact)) < 1e-12
    # (2) linearity in the grid values
    rng = np.random.default_rng(0)
    a, b = rng.uniform(0.5, 1.5, size=lin.shape), rng.uniform(0.5, 1.5, size=lin.shape)
    eng.set_values(eng.tensor(a))
    ta = eng.forward(o, d, 1000.0, 257).clone()
    eng.set_values(eng.tensor(b))
    tb = eng.forward(o, d, 1000.0, 257).clone()
    eng.set_values(eng.tensor(2.0 * a - 3.0 * b))
    tc = eng.forward(o, d, 1000.0, 257)
    assert float((tc - (2.0 * ta - 3.0 * tb)).abs().max()) < 1e-10 * float(ta.abs().max())
    # (3) <G x, y> == <x, G^T y>
    y = eng.tensor(rng.normal(size=sel.size))
    eng.set_values(eng.tensor(a))
    Gx = eng.forward(o, d, 1000.0, 257)
    Gty = eng.adjoint(o, d, y, 1000.0, 257)
    lhs, rhs = float(torch.dot(Gx, y)), float(torch.dot(eng.tensor(a).reshape(-1), Gty.reshape(-1)))
    assert abs(lhs - rhs) < 1e-10 * float(Gx.norm()) * float(y.norm())
    assert not eng.check_oob()
    # (4) the FULL per-GPU batch (260,400 rays): a random sample of rays against the C oracle, and the
    #     adjoint of the full batch against the oracle adjoint of a thinned batch is covered by (3)
    from oracle import oracle_c as OC
    eng.set_log_model(eng.tensor(w["m"]), w["K_ne"] / 1e13)
    of, df = eng.tensor(w["origins"]), eng.tensor(w["directions"])
    order = eng.locality_order(of, df, 1000.0)
    tec_full = eng.forward(of, df, 1000.0, 257).cpu().numpy()
    tec_ord = eng.forward(of, df, 1000.0, 257, order=order).cpu().numpy()
    assert np.array_equal(tec_full, tec_ord)                      # the walk order never changes a ray's TEC
    # dot-product test with EVERY ray of the batch, ordered (LDS-tiled) and unordered walks
    yf = eng.tensor(rng.normal(size=w["origins"].shape[0]))
    ne_t = eng.tensor(np.exp(w["m"]) * (w["K_ne"] / 1e13)).reshape(-1)
    for od in (order, None):
        Gty = eng.adjoint(of, df, yf, 1000.0, 257, order=od)
        lhs, rhs = float(torch.dot(torch.from_numpy(tec_full).cuda(), yf)), float(torch.dot(ne_t, Gty.reshape(-1)))
        assert abs(lhs - rhs) < 1e-10 * float(torch.from_numpy(tec_full).norm()) * float(yf.norm())
    pick = rng.choice(w["origins"].shape[0], 600, replace=False)
    ne = np.exp(w["m"]) * (w["K_ne"] / 1e13)
    ref = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], ne, w["origins"][pick], w["directions"][pick], 1000.0, 257)
    assert np.max(np.abs(tec_full[pick] - ref) / np.abs(ref)) < 1e-12


def test_tiled_adjoint_cfg2_ordered_and_unordered(O):
    """config 2 geometry (62 LOFAR stations x 42 directions, 128^3): the LDS-privatised adjoint must
    give the same back-projection whatever the walk order (in-tile and out-of-tile contributions,
    bundles of coincident and of scattered rays), in float64 and float32 accumulation."""
    from oracle import oracle_c as OC
    w = syn.make_workload(antennas="lofar", na=62, nd=42, nt=3, n=128)
    eng = make_engine(w)
    eng.set_values(eng.tensor(w["ne"] / 1e13))
    oo, dd = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    o, d = eng.tensor(oo), eng.tensor(dd)
    rng = np.random.default_rng(11)
    y = rng.normal(size=oo.shape[0])
    y[rng.integers(0, y.size, 200)] = 0.0                      # zero-weight rays are skipped
    ref = OC.adjoint_straight(w["xvec"], w["yvec"], w["zvec"], oo, dd, y, w["tmax"], 129)
    scale = np.max(np.abs(ref))
    yt = eng.tensor(y)
    order = eng.locality_order(o, d, w["tmax"])
    assert sorted(order.cpu().numpy().tolist()) == list(range(oo.shape[0]))
    rev = torch.flip(order, dims=[0]).contiguous()
    perm = eng.tensor(rng.permutation(oo.shape[0])).to(torch.int32)
    for od in (None, order, rev, perm):
        g = eng.adjoint(o, d, yt, w["tmax"], 129, order=od).cpu().numpy()
        assert np.max(np.abs(g - ref)) < 1e-11 * scale
        tec = eng.forward(o, d, w["tmax"], 129, order=od).cpu().numpy()
        assert np.all(np.isfinite(tec))
    g32 = eng.adjoint(o, d, yt, w["tmax"], 129, order=order, accum=torch.float32).cpu().numpy()
    assert np.max(np.abs(g32 - ref)) < 2e-5 * scale
    # even sample count (Ns = 128: no tail) and a tail longer than 8 samples (Ns = 140: masked slab)
    for Ns in (128, 140):
        ref2 = OC.adjoint_straight(w["xvec"], w["yvec"], w["zvec"], oo[:500], dd[:500], y[:500], w["tmax"], Ns)
        g2 = eng.adjoint(o[:500].contiguous(), d[:500].contiguous(), yt[:500].contiguous(), w["tmax"], Ns).cpu().numpy()
        assert np.max(np.abs(g2 - ref2)) < 1e-11 * np.max(np.abs(ref2))
        t2 = eng.forward(o[:500].contiguous(), d[:500].contiguous(), w["tmax"], Ns).cpu().numpy()
        r2 = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], w["ne"] / 1e13, oo[:500], dd[:500], w["tmax"], Ns)
        assert np.max(np.abs(t2 - r2) / np.abs(r2)) < 1e-12
    assert not eng.check_oob()


def test_edge_geometry_on_grid_faces_and_tiny_batches(O):
    """Samples exactly on the outer faces of the grid (cell index n-1, weight 0 on the far corner: the
    forward's unclamped read relies on the padded allocation), rays along a node line, R smaller than
    one wave group, Ns below 64, and zero-length batches."""
    from oracle import oracle_c as OC
    n = 24
    xv = np.linspace(-10.0, 13.0, n)
    yv = np.linspace(-7.0, 16.0, n)
    zv = np.linspace(0.0, 46.0, n)
    rng = np.random.default_rng(5)
    M = rng.uniform(1.0, 2.0, size=(n, n, n))
    from ionotomo_amd.engine import RayEngine
    eng = RayEngine(0)
    eng.set_grid(xv, yv, zv)
    eng.set_values(eng.tensor(M))
    oo = np.array([[xv[-1], yv[-1], zv[0]],        # runs up the far corner edge, ends on the top corner node
                   [xv[0], yv[0], zv[0]],          # near corner edge
                   [xv[3], yv[5], zv[0]],          # exactly along a node line
                   [xv[-1] - 4.0, yv[-1], zv[0]],  # ends exactly on the x = max face
                   [1.234, 2.345, 0.5]])
    dd = np.array([[0, 0, 1.0], [0, 0, 1.0], [0, 0, 2.0], [4.0, 0, zv[-1]], [0.05, -0.03, 1.0]])
    for Ns in (2, 5, 24, 47, 64, 65, 130):
        ref = OC.forward_tec_straight(xv, yv, zv, M, oo, dd, zv[-1], Ns)
        tec = eng.forward(eng.tensor(oo), eng.tensor(dd), zv[-1], Ns).cpu().numpy()
        assert np.max(np.abs(tec - ref) / np.abs(ref)) < 1e-12, Ns
        y = rng.normal(size=len(oo))
        g = eng.adjoint(eng.tensor(oo), eng.tensor(dd), eng.tensor(y), zv[-1], Ns).cpu().numpy()
        gref = OC.adjoint_straight(xv, yv, zv, oo, dd, y, zv[-1], Ns)
        assert np.max(np.abs(g - gref)) < 1e-12 * np.max(np.abs(gref)), Ns
    assert not eng.check_oob()
    empty = eng.forward(eng.tensor(np.zeros((0, 3))), eng.tensor(np.zeros((0, 3))), 10.0, 9)
    assert empty.shape == (0,)
    # one ray pokes out of the top: flagged, its TEC is NaN, the others are untouched
    tec = eng.forward(eng.tensor(oo), eng.tensor(dd), zv[-1] + 1.0, 33).cpu().numpy()
    assert eng.check_oob() and np.all(np.isnan(tec))
    tec = eng.forward(eng.tensor(oo), eng.tensor(dd * [1, 1, 1]), zv[-1] - 1.0, 33).cpu().numpy()
    assert not eng.check_oob() and np.all(np.isfinite(tec))


def test_very_long_rays_fall_back_to_the_table_kernels():
    """Ns above the LDS weight-table limit of the v2 kernels (4096) takes the table-uniform kernels."""
    from oracle import oracle_c as OC
    w = syn.make_workload("cfg1")
    eng = make_engine(w)
    eng.set_values(eng.tensor(w["ne"] / 1e13))
    oo, dd = w["origins"].reshape(-1, 3)[:5], w["directions"].reshape(-1, 3)[:5]
    for Ns in (4096, 4097, 6001):
        ref = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], w["ne"] / 1e13, oo, dd, w["tmax"], Ns)
        tec = eng.forward(eng.tensor(oo), eng.tensor(dd), w["tmax"], Ns).cpu().numpy()
        assert np.max(np.abs(tec - ref) / np.abs(ref)) < 1e-12
        y = np.arange(1.0, 6.0)
        g = eng.adjoint(eng.tensor(oo), eng.tensor(dd), eng.tensor(y), w["tmax"], Ns).cpu().numpy()
        gref = OC.adjoint_straight(w["xvec"], w["yvec"], w["zvec"], oo, dd, y, w["tmax"], Ns)
        assert np.max(np.abs(g - gref)) < 1e-11 * np.max(np.abs(gref))


def test_fused_vector_update_with_device_scalars():
    """iono_vec_axpby_dev: y = (sa a_num / a_den) x + (b_num / b_den) y, odd and even lengths, missing scalars = 1."""
    from ionotomo_amd.engine import RayEngine
    eng = RayEngine(0)
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 4096, 100003):
        x, y = rng.normal(size=n), rng.normal(size=n)
        an, ad, bn, bd = rng.uniform(0.5, 2.0, 4)
        t = lambda v: eng.tensor(np.asarray(v, dtype=np.float64))            # noqa: E731
        for kw, a, b in ((dict(a_num=t(an), a_den=t(ad)), an / ad, 1.0),
                         (dict(a_num=t(an), a_den=t(ad), a_sign=-1.0), -an / ad, 1.0),
                         (dict(b_num=t(bn), b_den=t(bd)), 1.0, bn / bd),
                         (dict(a_num=t(an), b_den=t(bd), a_sign=2.0), 2.0 * an, 1.0 / bd)):
            yt = eng.tensor(y)
            out = eng.axpby_(yt, eng.tensor(x), **kw)
            assert out is yt
            assert np.max(np.abs(yt.cpu().numpy() - (a * x + b * y))) < 1e-14 * (1 + abs(a) + abs(b)) * 5


def test_adjoint_partition_never_changes_results(O):
    """iono_walk_partition_set / iono_walk_cycles: any valid chunking of the walk -- equal counts, empty
    chunks, many more chunks than workgroups (handed out dynamically), the tuner's own -- gives the same gradient."""
    w = syn.make_workload("cfg2")
    eng = make_engine(w)
    o, d = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
    R = o.shape[0]
    order = eng.locality_order(o, d, w["tmax"])
    y = eng.tensor(np.random.default_rng(1).normal(size=R))
    eng.ctx.walk_partition_set(1, None, R)
    ref = eng.adjoint(o, d, y, w["tmax"], w["Ns"], order=order).clone()
    cyc, wg = eng.ctx.walk_cycles(1)
    assert wg >= 2 and cyc.size == wg and np.all(cyc > 0)
    rng = np.random.default_rng(2)
    scale = float(ref.abs().max())
    for n_chunks in (wg, wg + 1, 3 * wg, 7 * wg + 5):
        cuts = np.sort(rng.integers(0, R + 1, n_chunks - 1))
        cuts[: n_chunks // 3] = cuts[n_chunks // 3]                     # a run of empty chunks
        starts = np.concatenate([[0], np.sort(cuts), [R]]).astype(np.int64)
        eng.ctx.walk_partition_set(1, starts, R)
        g = eng.adjoint(o, d, y, w["tmax"], w["Ns"], order=order)
        c2, wg2 = eng.ctx.walk_cycles(1)
        assert wg2 == wg and c2.size == n_chunks
        assert float((g - ref).abs().max()) < 1e-12 * scale
        # fused-residual launch shares the partition
    stats = eng.tune_adjoint_partition(lambda: eng.adjoint(o, d, y, w["tmax"], w["Ns"], order=order), R)
    assert stats is not None and stats["tuned_ms"] <= stats["equal_count_ms"] * 1.0001
    g = eng.adjoint(o, d, y, w["tmax"], w["Ns"], order=order)
    assert float((g - ref).abs().max()) < 1e-12 * scale
    # a partition for another ray count is ignored, malformed ones are refused
    g = eng.adjoint(o[:-7].contiguous(), d[:-7].contiguous(), y[:-7].contiguous(), w["tmax"], w["Ns"])
    gref = O.adjoint_tec(O.straight_rays(w["origins"].reshape(-1, 3)[:-7], w["directions"].reshape(-1, 3)[:-7], w["tmax"], w["Ns"]),
                         w["xvec"], w["yvec"], w["zvec"], y[:-7].cpu().numpy())
    assert np.max(np.abs(g.cpu().numpy() - gref)) < 1e-11 * np.max(np.abs(gref))
    for bad in (np.array([1, R]), np.array([0, R - 1]), np.array([0, 50, 20, R])):
        with pytest.raises(ValueError):
            eng.ctx.walk_partition_set(1, bad.astype(np.int64), R)
    eng.ctx.walk_partition_set(1, None, R)


def test_forward_partition_never_changes_results():
    """The same mechanism for the forward kernel (one chunk per resident wave): arbitrary boundaries, same TEC."""
    from ionotomo_amd import _lib
    w = syn.make_workload("cfg2")
    eng = make_engine(w)
    eng.set_values(eng.tensor(w["ne"] / 1e13))
    o, d = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
    R = o.shape[0]
    ref = eng.forward(o, d, w["tmax"], w["Ns"]).clone()
    cyc, units = eng.ctx.walk_cycles(_lib.WALK_FORWARD)
    assert units >= 2 and cyc.size == units and np.all(cyc > 0)
    rng = np.random.default_rng(3)
    cuts = np.sort(rng.integers(0, R + 1, units - 1))
    cuts[: units // 4] = cuts[units // 4]
    eng.ctx.walk_partition_set(_lib.WALK_FORWARD, np.concatenate([[0], np.sort(cuts), [R]]), R)
    assert torch.equal(eng.forward(o, d, w["tmax"], w["Ns"]), ref)
    stats = eng.tune_forward_partition(lambda: eng.forward(o, d, w["tmax"], w["Ns"]), R, refine=1)
    assert stats is not None and stats["tuned_ms"] <= stats["equal_count_ms"]
    assert torch.equal(eng.forward(o, d, w["tmax"], w["Ns"]), ref)
    eng.ctx.walk_partition_set(_lib.WALK_FORWARD, None, R)


def test_cabi_collective_single_rank():
    """iono_comm_*: RCCL behind the C-ABI (hosts without torch.distributed).  One rank is all a 1-GPU box can run: the id
    round trip, the in-place sum on the ctx stream (identity for one rank, ordered after the kernel that produced the
    buffer), both dtypes, and the argument errors."""
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload("cfg1")
    eng = RayEngine(0)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    eng.set_values(eng.tensor(w["ne"] / 1e13))
    o, d = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
    y = torch.ones(o.shape[0], dtype=torch.float64, device="cuda")
    g = eng.adjoint(o, d, y, w["tmax"], 65)
    ref = g.clone()
    with pytest.raises(Exception, match="communicator"):
        eng.comm_allreduce_(g)
    cid = eng.comm_unique_id()
    assert len(cid) == 128 and any(cid)
    eng.comm_init(cid, 0, 1)
    with pytest.raises(Exception, match="already"):
        eng.comm_init(cid, 0, 1)
    eng.comm_allreduce_(g)
    g32 = ref.to(torch.float32)
    eng.comm_allreduce_(g32)
    torch.cuda.synchronize()
    assert torch.equal(g, ref) and torch.equal(g32, ref.to(torch.float32))
    with pytest.raises(Exception):
        eng.comm_init(cid, 3, 2)
    eng.comm_destroy()
    eng.comm_destroy()                                   # idempotent
    with pytest.raises(Exception, match="communicator"):
        eng.comm_allreduce_(g)


@pytest.mark.parametrize("interp", ["linear", "cubic"])
def test_graph_captured_solver_iterations_equal_the_eager_loop(interp):
    """``graph=True``: iterations 1 .. n-1 of the fused SIRT / CGLS loops are captured into one hipGraph (the library
    launches on the stream torch is capturing) and replayed -- the same kernels in the same order, so iterates and objective
    history equal the eager loop's to the rounding of the back-projection's floating-point atomics (whose order differs from
    run to run anyway); a second solve on the same engine works too."""
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload(antennas="lofar", na=62, nd=6, nt=3, n=40)
    eng = RayEngine(0, interp=interp)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    o, d = w["origins"].reshape(62, -1, 3), w["directions"].reshape(62, -1, 3)
    rng = np.random.default_rng(3)
    x_true = w["ne"] / 1e13
    eng.set_values(eng.tensor(x_true))
    P = o.shape[1]
    tmax = w["tmax"] if interp == "linear" else w["zvec"][-3]
    t = eng.forward(eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3)), tmax, 41).cpu().numpy().reshape(62, P)
    dobs = t - t[0] + rng.normal(size=t.shape) * 1e-3
    prob = parallel.ShardedRays(eng, o, d, tmax, 41, dobs=dobs, cdct=np.full_like(dobs, 1e-6), i0=0)
    x0 = eng.tensor(x_true * 0.8)
    for solve in (solvers.cgls, solvers.sirt):
        xa, ha = solve(prob, x0, n_iter=9)
        xb, hb = solve(prob, x0, n_iter=9, graph=True)
        xc, hc = solve(prob, x0, n_iter=9, graph=True)
        assert len(ha) == len(hb) == len(hc) == 9
        if solve is solvers.cgls or interp == "linear":         # (SIRT's normalisation assumes non-negative weights: no guarantee for cubic)
            assert ha[-1] < ha[0]
        assert np.allclose(ha, hb, rtol=1e-9, atol=1e-9 * ha[0]) and np.allclose(ha, hc, rtol=1e-9, atol=1e-9 * ha[0])
        scale = float(xa.abs().max())
        assert float((xa - xb).abs().max()) < 1e-9 * scale and float((xa - xc).abs().max()) < 1e-9 * scale


@pytest.mark.parametrize("interp", ["linear", "cubic"])
def test_small_problem_ray_pass_equals_the_separate_passes(interp):
    """At most 32 768 rays on one rank: residual / search-direction pass, dot products and the differential back-projection's ray
    weights run as ONE launch of one workgroup (iono_small_ray_pass_dev) instead of three kernels + k_ray_weights.  Iterates and
    objective history of CGLS and SIRT equal the separate passes' (``small_pass=False``) to the rounding of the back-projection's
    atomics, eager and graph-replayed; the entry point refuses larger problems."""
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload(antennas="lofar", na=62, nd=6, nt=3, n=40)
    eng = RayEngine(0, interp=interp)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    o, d = w["origins"].reshape(62, -1, 3), w["directions"].reshape(62, -1, 3)
    rng = np.random.default_rng(5)
    x_true = w["ne"] / 1e13
    eng.set_values(eng.tensor(x_true))
    P = o.shape[1]
    tmax = w["tmax"] if interp == "linear" else w["zvec"][-3]
    t = eng.forward(eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3)), tmax, 41).cpu().numpy().reshape(62, P)
    dobs = t - t[3] + rng.normal(size=t.shape) * 1e-3
    cdct = rng.uniform(0.5e-6, 2e-6, size=t.shape)
    prob = parallel.ShardedRays(eng, o, d, tmax, 41, dobs=dobs, cdct=cdct, i0=3)
    assert solvers._small(prob, True) and not solvers._small(prob, False)          # (cgls: on by default; sirt: opt-in)
    x0 = eng.tensor(x_true * 0.8)
    for solve in (solvers.cgls, solvers.sirt):
        xa, ha = solve(prob, x0, n_iter=8, small_pass=False)
        for graph in (False, True):
            xb, hb = solve(prob, x0, n_iter=8, small_pass=True, graph=graph)
            assert len(ha) == len(hb) == 8
            assert np.allclose(ha, hb, rtol=1e-9, atol=1e-9 * ha[0]), (solve.__name__, graph)
            assert float((xa - xb).abs().max()) < 1e-9 * float(xa.abs().max()), (solve.__name__, graph)
    big = torch.zeros(40000, dtype=torch.float64, device=eng.device)
    with pytest.raises(ValueError):
        eng.small_ray_pass(1, big, big, big.clone(), 4, 0, dobs=big, weight=big)


def test_coherent_order_is_a_permutation_and_never_changes_results():
    """RayEngine.coherent_order (rays grouped by antenna, direction along a Morton curve; the forward then interleaves the
    waves of an XCD in it): a permutation, and TEC per ray is bit-identical with and without it, for float64, float32
    (2 x 2 block layout) and tricubic."""
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload(antennas="lofar", na=62, nd=7, nt=5, n=48)
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    R = o.shape[0]
    for kw, tmax in (({}, w["tmax"]), ({"storage": "f32"}, w["tmax"]), ({"interp": "cubic"}, w["zvec"][-3])):
        eng = RayEngine(0, **kw)
        eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
        eng.set_values(eng.tensor(w["ne"] / 1e13))
        ot, dt = eng.tensor(o), eng.tensor(d)
        order = eng.coherent_order(ot, dt)
        assert order.dtype == torch.int32 and torch.equal(torch.sort(order.long()).values, torch.arange(R, device="cuda"))
        # neighbours in the order share their antenna far more often than in a random permutation
        same = (ot[order.long()][1:] == ot[order.long()][:-1]).all(dim=1).float().mean()
        assert float(same) > 0.95
        a = eng.forward(ot, dt, tmax, 97)
        b = eng.forward(ot, dt, tmax, 97, order=order)
        assert not eng.check_oob() and torch.equal(a, b)

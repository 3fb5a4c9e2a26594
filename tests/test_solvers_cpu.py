"""parallel.py + solvers.py (product host logic) driven by the CPU stand-in engine, against the
dense-matrix restatement in oracle/solvers.py.  CPU only."""
import numpy as np
import torch

from ionotomo_amd import parallel, solvers
from oracle import oracle as O
from oracle import solvers as OS
from cpu_engine import OracleEngine
from problems import small_problem


def setup():
    pb = small_problem()
    w = pb["w"]
    rays = O.straight_rays(pb["o"], pb["d"], pb["tmax"], pb["Ns"])
    G, A = OS.dense_operator(rays, w["xvec"], w["yvec"], w["zvec"], pb["i0"])
    d = A @ pb["x_true"].ravel() + pb["rng"].normal(size=A.shape[0]) * 1e-3
    cd = np.full(A.shape[0], 1e-6)
    eng = OracleEngine(w["xvec"], w["yvec"], w["zvec"])
    prob = parallel.ShardedRays(eng, pb["o"], pb["d"], pb["tmax"], pb["Ns"], dobs=d.reshape(pb["na"], pb["P"]),
                                cdct=cd.reshape(pb["na"], pb["P"]), i0=pb["i0"])
    return pb, G, A, d, cd, prob


def test_pair_block_partitions_evenly():
    for P in (1, 7, 10, 4200):
        for world in (1, 2, 3, 8):
            blocks = [parallel.pair_block(P, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == P
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b[1] - b[0] for b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_forward_adjoint_match_dense_operator():
    pb, G, A, d, cd, prob = setup()
    x = torch.from_numpy(pb["x_true"].copy())
    prob.engine.set_values(x)
    assert np.allclose(prob.forward().numpy(), A @ pb["x_true"].ravel(), rtol=1e-12, atol=1e-14)
    y = torch.from_numpy(pb["rng"].normal(size=A.shape[0]))
    assert np.allclose(prob.adjoint(y).numpy().ravel(), A.T @ y.numpy(), rtol=1e-11, atol=1e-13)
    tec = prob.forward_tec()
    g = prob.gradient_from_tec(tec).numpy().ravel()
    ref = A.T @ ((A @ pb["x_true"].ravel() - d) / (cd + 1e-15))
    assert np.allclose(g, ref, rtol=1e-10, atol=1e-8 * np.abs(ref).max())


def test_sirt_cgls_sd_match_dense_restatement():
    pb, G, A, d, cd, prob = setup()
    x0 = torch.from_numpy(pb["x0"].copy())
    xs, hs = solvers.sirt(prob, x0, n_iter=8)
    xr, hr = OS.sirt(G, A, d, cd, pb["x0"].ravel(), pb["na"], pb["P"], pb["i0"], 8)
    assert np.allclose(hs, hr, rtol=1e-9) and np.allclose(xs.numpy().ravel(), xr, rtol=1e-9, atol=1e-12)
    assert hs[-1] < hs[0]
    xc, hc = solvers.cgls(prob, x0, n_iter=8)
    xr, hr = OS.cgls(A, d, cd, pb["x0"].ravel(), 8)
    assert np.allclose(hc, hr, rtol=1e-7) and np.allclose(xc.numpy().ravel(), xr, rtol=1e-6, atol=1e-9)
    assert hc[-1] < 0.5 * hc[0]
    K = float(np.median(pb["x0"]))
    m0 = torch.from_numpy(np.log(pb["x0"] / K))                 # start at the prior, like the reference
    mm, hm = solvers.steepest_descent_log_model(prob, m0, K, max_iter=8)
    mr, hr = OS.steepest_descent_log_model(A, d, cd, np.log(pb["x0"] / K).ravel(), K, max_iter=8)
    assert len(hm) == len(hr) and np.allclose(hm, hr, rtol=1e-8)
    assert np.allclose(mm.numpy().ravel(), mr, rtol=1e-7, atol=1e-10)
    assert hm[-1] < hm[0]


def test_reference_stop_rule_in_sirt_and_cgls():
    """stop="reference" = the loop condition of inversion/iterative_newton.py:959-962,993: at least 5 updates, then
    stop on a small relative decrease, a small model change (pgtol) or the iteration cap."""
    pb, G, A, d, cd, prob = setup()
    x0 = torch.from_numpy(pb["x0"].copy())
    assert not solvers.reference_stop(10.0, 9.0, 1.0, 3)                      # fewer than 5 updates: go on
    assert solvers.reference_stop(10.0, 9.0, 1.0, 20) and not solvers.reference_stop(10.0, 9.0, 1.0, 19)
    assert solvers.reference_stop(10.0, 10.0 - 1e-10, 1.0, 7)                 # relative decrease <= 1e7 eps
    assert solvers.reference_stop(10.0, 9.0, 5e-3, 7)                         # max |dm| <= pgtol
    for pgtol, cap in ((1e-2, 20), (1e-7, 10), (1e-7, 8), (0.0, 6)):
        xs, hs = solvers.sirt(prob, x0, n_iter=cap, stop="reference", pgtol=pgtol)
        xr, hr = OS.sirt(G, A, d, cd, pb["x0"].ravel(), pb["na"], pb["P"], pb["i0"], cap, stop=True, pgtol=pgtol)
        assert len(hs) == len(hr) and 6 <= len(hs) <= cap + 1
        assert np.allclose(hs, hr, rtol=1e-9) and np.allclose(xs.numpy().ravel(), xr, rtol=1e-9, atol=1e-12)
        xc, hc = solvers.cgls(prob, x0, n_iter=cap, stop="reference", pgtol=pgtol)
        xr, hr = OS.cgls(A, d, cd, pb["x0"].ravel(), cap, stop=True, pgtol=pgtol)
        assert len(hc) == len(hr) and 6 <= len(hc) <= cap + 1
        assert np.allclose(hc, hr, rtol=1e-6, atol=1e-9 * hr[0])
        assert np.max(np.abs(xc.numpy().ravel() - xr)) < 1e-4 * np.max(np.abs(xr))
    # the default pgtol = 1e-2 ends both solvers right after the 5 mandatory updates on this problem (tiny steps)
    _, h = solvers.sirt(prob, x0, n_iter=20, stop="reference")
    assert len(h) == 6


def test_sirt_is_a_contraction_in_its_own_norms():
    """SIRT x += C A^T L (d - A x) with L, C from the row / column sums: for consistent data the C^-1-weighted model
    error and the L-weighted residual decrease at EVERY iteration (rho(C A^T L A) <= 1); the plain 2-norm of the
    model error carries no such guarantee -- the update C A^T(...) is orthogonal to null(A) only in the C^-1 inner
    product (explains the config-5 observation: objective falls, plain model error grows)."""
    pb, G, A, d, cd, prob = setup()
    d = A @ pb["x_true"].ravel()                                   # consistent data
    Na, P, i0 = pb["na"], pb["P"], pb["i0"]
    rows = G.sum(1).reshape(Na, P)
    L = 1.0 / (rows + rows[i0:i0 + 1]).ravel()
    wcol = np.ones((Na, P))
    wcol[i0] += Na
    col = G.T @ wcol.ravel()
    live = col > 1e-9 * col.max()
    prob.dobs = torch.from_numpy(d.copy())
    errs, res = [], []

    def cb(k, x, S):
        e = x.numpy().ravel() - pb["x_true"].ravel()
        errs.append(np.sum(col[live] * e[live] ** 2))
        r = d - A @ x.numpy().ravel()
        res.append(np.sum(L * r * r))
    solvers.sirt(prob, torch.from_numpy(pb["x0"].copy()), n_iter=25, callback=cb)
    errs, res = np.array(errs), np.array(res)
    assert np.all(np.diff(errs) <= 1e-12 * errs[0]) and errs[-1] < errs[0]
    assert np.all(np.diff(res) <= 1e-12 * res[0]) and res[-1] < 0.05 * res[0]


def test_walk_orders_are_permutations_on_cpu():
    """RayEngine.coherent_order / locality_order are host-side torch plumbing (no kernel involved): permutations of the rays,
    the coherent one grouping by antenna with neighbouring directions adjacent."""
    import torch
    from ionotomo_amd.engine import RayEngine
    from ionotomo_amd import synthetic as syn
    w = syn.make_workload(antennas="lofar", na=20, nd=6, nt=7, n=16)
    o = torch.from_numpy(w["origins"].reshape(-1, 3).copy())
    d = torch.from_numpy(w["directions"].reshape(-1, 3).copy())
    R = o.shape[0]
    for order in (RayEngine.coherent_order(o, d), RayEngine.locality_order(o, d, w["tmax"])):
        assert order.dtype == torch.int32 and torch.equal(torch.sort(order.long()).values, torch.arange(R))
    co = RayEngine.coherent_order(o, d).long()
    same_antenna = (o[co][1:] == o[co][:-1]).all(dim=1)
    assert int((~same_antenna).sum()) == 20 - 1                      # one boundary between consecutive antenna groups
    # inside a group neighbours point almost the same way: far closer than two random rays of the group
    dn = d / d.norm(dim=1, keepdim=True)
    step = (dn[co][1:] - dn[co][:-1]).norm(dim=1)[same_antenna].median()
    rand = (dn[co][torch.randperm(R)][1:] - dn[co][:-1]).norm(dim=1).median()
    assert float(step) < 0.25 * float(rand)


def test_fermat_step_doubling_control_host_logic():
    """inversion/fermat.py's step control (the reference integrates with adaptive LSODA at rtol = atol = 1.49e-8,
    inversion/fermat.py:163-167; here fixed-step RK4 with the step count chosen by step doubling on a sample): a synthetic tracer
    whose error falls like C / s^4 -- the smallest power of two that meets the tolerance is chosen, every level traced once."""
    from ionotomo_amd.inversion import fermat as F
    assert F.sample_indices(2604).size == 208 and F.sample_indices(620000).size == 6200 and F.sample_indices(5).tolist() == [0, 1, 2, 3, 4]
    assert F.sample_indices(0).size == 0
    calls = []

    def tracer(C):
        def tr(s):
            calls.append(s)
            r = np.zeros((3, 4, 5))
            r[:, 0], r[:, 1], r[:, 2], r[:, 3] = 100 + C / s ** 4, 50.0, np.linspace(0, 1, 5), 1000 + 10 * C / s ** 4
            return r
        return tr
    sub, rep = F.choose_substeps(tracer(1e-4))
    assert sub == 4 and rep["met"] and [l["substeps"] for l in rep["levels"]] == [1, 2, 4] and calls == [1, 2, 4, 8]
    assert rep["levels"][-1]["error_over_tolerance"] <= 0.5 < rep["levels"][-2]["error_over_tolerance"]
    sub, rep = F.choose_substeps(tracer(1e-9))
    assert sub == 1 and rep["met"] and len(rep["levels"]) == 1
    sub, rep = F.choose_substeps(tracer(1e3), max_substeps=8)
    assert sub == 8 and not rep["met"]
    sub, rep = F.choose_substeps(tracer(1e-4), rtol=1e-5)
    assert sub == 1 and rep["rtol"] == rep["atol"] == 1e-5
    assert F.doubling_error(np.zeros((0, 4, 3)), np.zeros((0, 4, 3)), 1e-8, 1e-8) == 0.0

"""Randomised geometry / grid / quadrature sweep of the HIP forward and adjoint against the numpy
oracle (all three kernel tiers are reached: ideal-uniform, table-uniform, non-uniform).  -m gpu."""
import numpy as np
import pytest

from ionotomo_amd import _lib

pytestmark = pytest.mark.gpu
SOAK = int(__import__("os").environ.get("IONO_SOAK", "1"))      # IONO_SOAK=20: twenty times the seeds (a soak run on the GPU box)


def random_axis(rng, n, kind, lo, hi):
    if kind == "ideal":
        return np.linspace(lo, hi, n)
    if kind == "table":                       # uniform to ~1e-8: not "ideal", still guess + verify
        v = np.linspace(lo, hi, n)
        v[1:-1] += rng.uniform(-1, 1, n - 2) * 1e-8 * (hi - lo) / n
        return v
    v = np.cumsum(rng.uniform(0.3, 1.7, n))   # non-uniform
    return lo + (v - v[0]) * (hi - lo) / (v[-1] - v[0])


@pytest.mark.parametrize("seed", range(SOAK * 12))
def test_random_problem(seed):
    from oracle import oracle as O
    rng = np.random.default_rng(100 + seed)
    kinds = [("ideal", "ideal", "ideal"), ("table", "ideal", "ideal"), ("ideal", "nonuniform", "table"),
             ("nonuniform", "nonuniform", "nonuniform")][seed % 4]
    nx, ny, nz = rng.integers(6, 40, 3)
    xv = random_axis(rng, nx, kinds[0], -30.0, 25.0)
    yv = random_axis(rng, ny, kinds[1], -18.0, 33.0)
    zv = random_axis(rng, nz, kinds[2], -5.0, 120.0)
    M = rng.uniform(0.5, 2.0, size=(nx, ny, nz))
    R = int(rng.integers(1, 90))
    Ns = int(rng.choice([2, 3, 7, 8, 9, 31, 64, 65, 72, 100, 129, 200]))
    rule = ["avg", "scipy", "trapz"][seed % 3]
    tmax = zv[-1] - rng.uniform(0.0, 10.0)
    o = np.stack([rng.uniform(xv[0] + 14, xv[-1] - 14, R), rng.uniform(yv[0] + 14, yv[-1] - 14, R), rng.uniform(zv[0], zv[0] + 10, R)], -1)
    d = np.stack([rng.uniform(-0.05, 0.05, R), rng.uniform(-0.05, 0.05, R), rng.uniform(0.5, 1.5, R)], -1)
    ctx = _lib.Context(0)
    storage = "f32" if seed % 5 == 4 else "f64"
    ctx.set_grid(xv, yv, zv, M, storage=storage)
    rays = O.straight_rays(o, d, tmax, Ns)
    ref = O.forward_tec(rays, xv, yv, zv, M, _lib.quad_rule(rule))
    tol = 3e-7 if storage == "f32" else 1e-12
    tec = ctx.forward_tec_straight(o, d, tmax, Ns, rule=rule)
    assert np.max(np.abs(tec - ref) / np.abs(ref)) < tol
    tec2 = ctx.forward_tec_rays(rays, rule=rule)
    assert np.max(np.abs(tec2 - ref) / np.abs(ref)) < tol
    y = rng.normal(size=R)
    gref = O.adjoint_tec(rays, xv, yv, zv, y, _lib.quad_rule(rule))
    g = ctx.adjoint_straight(o, d, y, tmax, Ns, rule=rule)
    assert np.max(np.abs(g - gref)) < 1e-11 * np.max(np.abs(gref))
    g2 = ctx.adjoint_rays(rays, y, rule=rule)
    assert np.max(np.abs(g2 - gref)) < 1e-11 * np.max(np.abs(gref))
    ctx.close()


@pytest.mark.parametrize("seed", range(SOAK * 4))
def test_tracer_lane_mappings_agree_on_hard_geometry(seed, monkeypatch):
    """The cell-cached tracers (4 / 8 lanes per ray) against the plain lanes = rays kernels and the oracle: strongly
    tilted rays that change (i, j) column every few cells, origins exactly on grid nodes, the top reached exactly
    (z-run clamp at the axis end), non-uniform axes, smooth refractive-index field with real bending."""
    from oracle import oracle as O
    rng = np.random.default_rng(500 + seed)
    kinds = [("ideal", "ideal", "ideal"), ("table", "ideal", "nonuniform"), ("nonuniform", "nonuniform", "nonuniform"),
             ("ideal", "nonuniform", "ideal")][seed % 4]
    nx, ny, nz = 30 + seed % 9, 27, 33 + 2 * (seed % 7)
    xv = random_axis(rng, nx, kinds[0], -60.0, 60.0)
    yv = random_axis(rng, ny, kinds[1], -55.0, 65.0)
    zv = random_axis(rng, nz, kinds[2], 0.0, 160.0)
    X, Y, Z = np.meshgrid(xv, yv, zv, indexing="ij")
    freq = 40e6
    ne = 4e11 * np.exp(-((Z - 80.0) / 40.0) ** 2) * (1.0 + 0.3 * np.sin(X / 17.0) * np.cos(Y / 23.0))
    assert np.all(8.980 ** 2 * ne / freq ** 2 < 0.5)
    R = 37
    o = np.stack([rng.uniform(-12, 12, R), rng.uniform(-12, 12, R), rng.uniform(zv[2], zv[3], R)], -1)
    o[:5] = np.stack([xv[nx // 2 + np.arange(5)], yv[ny // 2 - np.arange(5)], np.full(5, zv[2])], -1)   # on nodes
    d = np.stack([rng.uniform(-0.25, 0.25, R), rng.uniform(-0.25, 0.25, R), np.ones(R)], -1)
    # (soak seeds: a ray whose straight end would come within three cells of the tricubic domain's side faces is made less oblique --
    #  bending moves an end by less than a cell here, and a ray that leaves the domain is an error by design, not a tracer difference)
    for ax, v in ((0, xv), (1, yv)):
        lo, hi = v[5], v[-6]
        end = o[:, ax] + d[:, ax] * (zv[-3] - o[:, 2])
        over = np.maximum(np.maximum(lo - end, end - hi), 0.0)
        d[:, ax] -= np.sign(d[:, ax]) * over / (zv[-3] - o[:, 2])
    # The gradient of a TRILINEAR field jumps across cell faces, so a Runge-Kutta stage that lands on a face to within rounding
    # is evaluated in one cell or the other depending on the last bit of z (fused vs separate multiply-add): 1e-3 km of
    # difference in s between two correct implementations.  Seeds 0-3 have exactly representable steps (or no such hits); the
    # soak seeds end a random distance below the top of the tricubic domain, so steps and cell widths are incommensurate.
    tmax = zv[-3] if seed < 4 else zv[-3] - rng.uniform(0.01, 0.5)
    o[5:8, 2] = zv[0]
    Ns, sub = 41, 3
    ctxs = []
    for env in ({}, {"IONOTOMO_FERMAT_COOP_MAX": "0", "IONOTOMO_FERMAT_LIN4_MAX": "0"}, {"IONOTOMO_FERMAT_LIN4_RPW": "16"}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        c = _lib.Context(0)
        for k_ in env:
            monkeypatch.delenv(k_)
        c.set_grid(xv, yv, zv, ne)
        ctxs.append(c)
    nM = O.ne_to_n(ne, freq)
    for kind, field in (("linear", O.n_field_trilinear(xv, yv, zv, nM)), ("cubic", O.n_field_tricubic(xv, yv, zv, nM))):
        out = [c.trace_fermat(o, d, tmax, Ns, freq, bend=True, kind=kind, substeps=sub) for c in ctxs]
        assert not any(c.check_oob() for c in ctxs)
        ref = O.fermat_trace(o, d, tmax, Ns, field, bend=True, substeps=sub)
        for r in out:
            assert np.max(np.abs(r.reshape(ref.shape) - ref)) < 1e-8
        straight = O.straight_rays(o, d, tmax, Ns)
        assert np.max(np.abs(out[0].reshape(ref.shape)[..., 0, -1] - straight[..., 0, -1])) > 1e-3   # it does bend
    # tracing to the very top of the grid: every mapping must stay in bounds (the run of cached nodes is clamped)
    top = [c.trace_fermat(o, d * np.array([0.02, 0.02, 1.0]), zv[-1], Ns, freq, bend=True, kind="linear", substeps=sub) for c in ctxs]
    assert np.max(np.abs(top[0] - top[1])) < 1e-8 and np.max(np.abs(top[0] - top[2])) < 1e-8
    for c in ctxs:
        c.close()


def test_tiled_adjoint_soak():
    """Many bundle shapes through the LDS-privatised adjoint in one process: dense coincident fans, sparse
    fans, mixed zero weights, ragged last bundles, with and without a walk order -- each against the C oracle."""
    import torch
    from oracle import oracle_c as OC
    from ionotomo_amd.engine import RayEngine
    rng = np.random.default_rng(2024)
    n = 48
    xv, yv, zv = np.linspace(-60, 60, n), np.linspace(-55, 65, n), np.linspace(-2, 210, n)
    eng = RayEngine(0)
    eng.set_grid(xv, yv, zv)
    eng.set_values(eng.tensor(rng.uniform(1, 2, size=(n, n, n))))
    for case in range(24):
        R = int(rng.choice([1, 3, 16, 17, 63, 64, 65, 130, 400, 1000]))
        Ns = int(rng.choice([9, 64, 65, 70, 129]))
        nant = int(rng.integers(1, 6))
        ants = np.stack([rng.uniform(-8, 8, nant), rng.uniform(-8, 8, nant), rng.uniform(0, 1.0, nant)], -1)
        spread = float(rng.choice([0.002, 0.02, 0.045]))             # coincident ... wide fan (stays inside the grid)
        a = rng.integers(0, nant, R)
        o = ants[a] + rng.normal(scale=0.01, size=(R, 3)) * [1, 1, 0]
        d = np.stack([np.clip(rng.normal(scale=spread, size=R), -0.2, 0.2), np.clip(rng.normal(scale=spread, size=R), -0.2, 0.2),
                      np.ones(R)], -1)
        y = rng.normal(size=R)
        y[rng.random(R) < 0.2] = 0.0
        ref = OC.adjoint_straight(xv, yv, zv, o, d, y, 200.0, Ns)
        ot, dt, yt = eng.tensor(o), eng.tensor(d), eng.tensor(y)
        order = eng.locality_order(ot, dt, 200.0) if case % 2 else None
        g = eng.adjoint(ot, dt, yt, 200.0, Ns, order=order).cpu().numpy()
        assert not eng.check_oob(), (case, R, Ns, spread)
        scale = max(np.max(np.abs(ref)), 1e-300)
        assert np.max(np.abs(g - ref)) < 1e-11 * scale, (case, R, Ns, spread)
        if ref.any():
            t = eng.forward(ot, dt, 200.0, Ns, order=order)
            lhs = float(torch.dot(t, yt))
            rhs = float((eng.tensor(g) * eng.tensor(eng.ctx.get_values())).sum())
            assert abs(lhs - rhs) < 1e-9 * (abs(lhs) + abs(rhs) + 1e-300)
    assert not eng.check_oob()


def test_library_then_torch_in_a_fresh_process():
    """Load order regression: the C-ABI context first, torch afterwards (torch bundles a HIP runtime with the
    same soname as the system one; _lib.load() makes sure a single runtime serves both)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from ionotomo_amd import _lib\n"
            "c = _lib.Context(0)\n"
            "import torch\n"
            "assert torch.cuda.is_available()\n"
            "from ionotomo_amd.engine import RayEngine\n"
            "e = RayEngine(0)\n"
            "print('ok')\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("seed", range(SOAK * 6))
def test_random_curved_ray_launches_fused_against_traced(seed, monkeypatch):
    """Random grids (ideal-uniform, any parity, some too small for the scatter window), ray sets, sample counts, rules and independent
    variables: the fused curved-ray forward and transpose -- every route the library picks: lanes = rays, the record stepper with 8 and
    with 2 lanes per ray, the windowed and the plain scatter -- against trace + explicit-sample kernels, and <A x, w> = <x, A^T w>."""
    import torch
    from ionotomo_amd.engine import RayEngine
    rng = np.random.default_rng(900 + seed)
    nx, ny, nz = (int(v) for v in rng.integers(9, 34, 3))
    xv, yv, zv = np.linspace(-30.0, 25.0, nx), np.linspace(-18.0, 33.0, ny), np.linspace(0.0, 120.0, nz)
    # a smooth ionosphere (bending stays a fraction of a cell) + roughness; values in m^-3
    X, Y, Z = np.meshgrid(xv, yv, zv, indexing="ij")
    ne = 1e11 * (1.0 + 0.5 * np.exp(-((Z - 60.0) / 25.0) ** 2) * (1.0 + 0.2 * np.sin(X / 9.0) * np.cos(Y / 7.0))) * rng.uniform(0.97, 1.03, size=X.shape)
    R = int(rng.integers(3, 150))
    Ns = int(rng.choice([5, 8, 9, 16, 21, 30]))
    typ = "zs"[seed % 2]
    rule = ["avg", "scipy", "trapz"][seed % 3]
    kind = ["cubic", "linear"][(seed // 2) % 2]
    interp = ["linear", "cubic"][(seed // 3) % 2]
    lanes = [None, "2", "8"][seed % 3]
    if lanes:
        monkeypatch.setenv("IONOTOMO_FERMAT_LM_LANES", lanes)
    eng = RayEngine(0, interp=interp, quad=rule)
    eng.set_grid(xv, yv, zv)
    eng.set_values(eng.tensor(ne))
    hx, hy = xv[1] - xv[0], yv[1] - yv[0]
    o = np.stack([rng.uniform(xv[0] + 4 * hx, xv[-1] - 4 * hx, R), rng.uniform(yv[0] + 4 * hy, yv[-1] - 4 * hy, R), rng.uniform(zv[2], zv[3], R)], -1)
    d = np.stack([rng.uniform(-0.02, 0.02, R), rng.uniform(-0.02, 0.02, R), rng.uniform(0.8, 1.2, R)], -1)
    tmax = float(zv[-3] - rng.uniform(0.0, 5.0)) if typ == "z" else float(0.8 * (zv[-3] - zv[3]))
    ot, dt = eng.tensor(o), eng.tensor(d)
    y = eng.tensor(rng.normal(size=R))
    kw = dict(bend=True, kind=kind, substeps=int(rng.integers(1, 4)), type=typ, ne_scale=1e-13)
    a = eng.forward_fermat(ot, dt, tmax, Ns, 80e6, fused=True, **kw)
    b = eng.forward_fermat(ot, dt, tmax, Ns, 80e6, fused=False, **kw)
    assert not eng.check_oob(), "test geometry: rays must stay inside"
    assert float((a - b).abs().max()) < 1e-11 * float(b.abs().max()), (seed, "forward")
    ga = eng.adjoint_fermat(ot, dt, y, tmax, Ns, 80e6, fused=True, **kw)
    gb = eng.adjoint_fermat(ot, dt, y, tmax, Ns, 80e6, fused=False, **kw)
    assert float((ga - gb).abs().max()) < 1e-11 * float(gb.abs().max()), (seed, "transpose")
    lhs, rhs = float((a * y).sum()), float((ga * eng.tensor(ne)).sum()) * 1.0
    assert abs(lhs - rhs) < 1e-10 * float((a.abs() * y.abs()).sum()), (seed, "dot")
    del torch

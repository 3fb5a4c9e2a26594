"""Randomised geometry / grid / quadrature sweep of the HIP forward and adjoint against the numpy
oracle (all three kernel tiers are reached: ideal-uniform, table-uniform, non-uniform).  -m gpu."""
import numpy as np
import pytest

from ionotomo_amd import _lib

pytestmark = pytest.mark.gpu


def random_axis(rng, n, kind, lo, hi):
    if kind == "ideal":
        return np.linspace(lo, hi, n)
    if kind == "table":                       # uniform to ~1e-8: not "ideal", still guess + verify
        v = np.linspace(lo, hi, n)
        v[1:-1] += rng.uniform(-1, 1, n - 2) * 1e-8 * (hi - lo) / n
        return v
    v = np.cumsum(rng.uniform(0.3, 1.7, n))   # non-uniform
    return lo + (v - v[0]) * (hi - lo) / (v[-1] - v[0])


@pytest.mark.parametrize("seed", range(12))
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

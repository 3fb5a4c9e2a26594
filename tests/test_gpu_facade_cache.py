"""The reference-signature facade is EXACT by default: forward_equation / compute_gradient upload ``rays`` and ``m_tci.M`` on
every call, so in-place edits -- one node of the model (the reference's own finite-difference loop, tests/test_inversion.py:71-87),
one sample of a ray -- are always seen.  ``assume_unchanged=True`` is the opt-in that keeps operands
resident between calls: the SAME rays array re-uses its device copy, the node values are recomputed only when
the model object / scale / sampled fingerprint changed, and every cached answer equals the uncached one bit for bit as long as
the caller keeps the promise.  Needs a real MI355X: -m gpu."""
import json
import os
import time

import numpy as np
import pytest

import ionotomo_amd as it
from ionotomo_amd import _lib, synthetic as syn
from ionotomo_amd.inversion.forward_equation import forward_equation
_fe = forward_equation
from ionotomo_amd.inversion.gradient import compute_gradient

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg2():
    from oracle import oracle as O
    w = syn.make_workload("cfg2")
    rays = O.straight_rays(w["origins"], w["directions"], w["tmax"], w["Ns"])            # [62,1,42,4,129]
    return w, np.ascontiguousarray(rays)


def uncached(fn, *a, **k):
    _lib.default_context().forget()
    out = fn(*a, **k)
    _lib.default_context().forget()
    return out


def test_repeated_forward_equation_is_resident_and_identical(cfg2):
    w, rays = cfg2
    forward_equation = lambda *a, **k: _fe(*a, assume_unchanged=True, **k)      # this test is about the opt-in path
    tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["m"].copy())
    ref = uncached(forward_equation, rays, w["K_ne"], tci, 0)
    first = forward_equation(rays, w["K_ne"], tci, 0)
    ctx = _lib.default_context()
    assert len(ctx._dev_arrays) == 1 and ctx._values_key is not None
    t = []
    for _ in range(20):
        t0 = time.perf_counter()
        again = forward_equation(rays, w["K_ne"], tci, 0)
        t.append(time.perf_counter() - t0)
    assert np.array_equal(first, ref) and np.array_equal(again, ref)
    assert len(ctx._dev_arrays) == 1
    assert np.median(t) < 300e-6, "second call took %.0f us" % (np.median(t) * 1e6)      # (measured ~50 us; 0.66 ms uncached)
    # the model changes in place (m += step): seen through the fingerprint, node values recomputed, rays stay resident
    tci.M *= 1.01
    changed = forward_equation(rays, w["K_ne"], tci, 0)
    assert len(ctx._dev_arrays) == 1
    assert np.max(np.abs(changed - ref)) > 1e-6 * np.max(np.abs(ref))
    assert np.array_equal(changed, uncached(forward_equation, rays, w["K_ne"], tci, 0))
    # a new model OBJECT (what a line search passes: m + alpha dm)
    tci2 = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["m"] + 0.02)
    assert np.array_equal(forward_equation(rays, w["K_ne"], tci2, 0), uncached(forward_equation, rays, w["K_ne"], tci2, 0))
    # another K_ne with the same model: the scale is part of the key
    assert np.allclose(forward_equation(rays, 2.0 * w["K_ne"], tci2, 0), 2.0 * forward_equation(rays, w["K_ne"], tci2, 0), rtol=1e-13)
    # a new rays OBJECT is uploaded afresh; the reference antenna is applied on the device
    rays2 = rays[:, :, ::-1].copy()
    out2 = forward_equation(rays2, w["K_ne"], tci2, 3)
    assert np.array_equal(out2, uncached(forward_equation, rays2, w["K_ne"], tci2, 3))
    assert np.all(out2[3] == 0.0)
    # operands without an identity to key on (a list, a non-contiguous view) take the plain path
    view = rays[:, :, ::2]
    assert np.array_equal(forward_equation(view, w["K_ne"], tci2, 0), forward_equation(np.ascontiguousarray(view), w["K_ne"], tci2, 0))
    # out-of-grid samples still raise, as scipy's bounds_error=True does in the reference
    bad = rays.copy()
    bad[0, 0, 0, 2, -1] = w["zvec"][-1] + 50.0
    with pytest.raises(ValueError):
        forward_equation(bad, w["K_ne"], tci2, 0)
    assert np.array_equal(forward_equation(rays, w["K_ne"], tci2, 0), uncached(forward_equation, rays, w["K_ne"], tci2, 0))


def test_gradient_on_resident_rays_equals_the_plain_path(cfg2):
    w, rays = cfg2
    tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["m"].copy())
    g = forward_equation(rays, w["K_ne"], tci, 1, assume_unchanged=True)
    rng = np.random.default_rng(0)
    dobs = g + rng.normal(size=g.shape) * 0.01
    CdCt = np.full(g.shape, 1e-4)
    args = (rays, g, dobs, 1, w["K_ne"], tci, None, CdCt, 1.0, 3, 1.0)
    grad = compute_gradient(*args, assume_unchanged=True)
    assert len(_lib.default_context()._dev_arrays) == 1                          # the rays forward_equation left on the device
    ref = uncached(lambda: compute_gradient(np.array(rays), *args[1:]))          # a fresh rays object on a fresh cache
    assert grad.shape == tci.M.shape
    assert np.max(np.abs(grad - ref)) <= 1e-12 * np.max(np.abs(ref))             # (atomics: summation order differs run to run)


def test_short_lived_models_never_alias_in_the_cache():
    """A finite difference builds m + e, drops it and builds m - e: CPython hands the second array the id AND the data address of
    the first, and one perturbed node sits between the fingerprint's probes -- the cache must key the model on a weak reference
    to the live object, not on id() / address (it returned the m + e answer for m - e: a zero derivative)."""
    w = syn.make_workload(antennas="example", na=4, nd=3, nt=1, n=12)
    from oracle import oracle as O
    rays = np.ascontiguousarray(O.straight_rays(w["origins"], w["directions"], w["tmax"], 13))
    m = w["m"]
    node = int(np.argmax(np.abs(compute_gradient(rays, forward_equation(rays, w["K_ne"], it.TriCubic(w["xvec"], w["yvec"], w["zvec"], m), 1),
                                                 np.zeros((4, 1, 3)), 1, w["K_ne"], it.TriCubic(w["xvec"], w["yvec"], w["zvec"], m), None,
                                                 np.full((4, 1, 3), 1e-4), None, None, None))))

    def g_of(mm):
        return forward_equation(rays, w["K_ne"], it.TriCubic(w["xvec"], w["yvec"], w["zvec"], mm), 1, assume_unchanged=True)
    e = np.zeros(m.size)
    e[node] = 1e-3
    e = e.reshape(m.shape)
    for _ in range(30):
        gp, gm = g_of(m + e), g_of(m - e)          # the temporaries die between the calls
        assert np.max(np.abs(gp - gm)) > 0.0
        assert np.array_equal(gp, uncached(g_of, m + e)) and np.array_equal(gm, uncached(g_of, m - e))


def _fresh(fn, rays, K_ne, tci, *a, **k):
    """The same call on brand-new objects and an empty cache: what `uncached` means for the default path."""
    _lib.default_context().forget()
    t2 = it.TriCubic(tci.xvec.copy(), tci.yvec.copy(), tci.zvec.copy(), tci.M.copy(), kind=tci.kind)
    return fn(np.array(rays), K_ne, t2, *a, **k)


def test_reference_finite_difference_loop_replayed_literally():
    """/root/reference/src/ionotomo/tests/test_inversion.py:71-87: 20 random nodes with a non-zero gradient, ``m_tci.M[i,j,k] += 1e-7``
    IN PLACE on the same object, ``forward_equation(rays, K_ne, m_tci, i0)`` again with the same ``rays``: every call must see
    the edit (bit-equal to the call on fresh copies) and the numerical gradient must match ``compute_gradient``."""
    from oracle import oracle as O
    w = syn.make_workload("cfg1")
    rays = np.ascontiguousarray(O.straight_rays(w["origins"], w["directions"], w["tmax"], w["Ns"]))
    m_tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["m"].copy())
    K_ne, i0 = w["K_ne"], 0
    d = forward_equation(rays, K_ne, m_tci, i0)
    rng = np.random.default_rng(5)
    dobs = d + rng.normal(size=d.shape) * 0.01
    CdCt = (0.01 * np.ones(dobs.shape)) ** 2
    gradient = compute_gradient(rays, d, dobs, i0, K_ne, m_tci, m_tci.M, CdCt, 1, 4, 5., None)
    S0 = np.sum((d - dobs) ** 2 / (CdCt + 1e-15)) / 2.
    gmax = np.max(np.abs(gradient))
    i, Ncheck, worst = 0, 20, 0.0
    while i < Ncheck:
        xi, yi, zi = rng.integers(m_tci.nx), rng.integers(m_tci.ny), rng.integers(m_tci.nz)
        while abs(gradient[xi, yi, zi]) < 1e-3 * gmax:
            xi, yi, zi = rng.integers(m_tci.nx), rng.integers(m_tci.ny), rng.integers(m_tci.nz)
        m_tci.M[xi, yi, zi] += 1e-7
        g = forward_equation(rays, K_ne, m_tci, i0)
        assert np.array_equal(g, _fresh(forward_equation, rays, K_ne, m_tci, i0)), "in-place edit of node %s not seen" % ((xi, yi, zi),)
        assert np.max(np.abs(g - d)) > 0.0
        S = np.sum((g - dobs) ** 2 / (CdCt + 1e-15)) / 2.
        grad_num = (S - S0) / 1e-7
        m_tci.M[xi, yi, zi] -= 1e-7
        worst = max(worst, abs(grad_num - gradient[xi, yi, zi]) / gmax)
        i += 1
    # forward difference with h = 1e-7: truncation ~ h |S''|/2 + rounding of S (1e-16 S0 / h)
    assert worst < 1e-5, worst
    # and back at the start: the unperturbed model gives the first answer again
    assert np.array_equal(forward_equation(rays, K_ne, m_tci, i0), d)


def test_in_place_edit_of_a_ray_sample_is_seen_by_both_facade_functions(cfg2):
    w, rays0 = cfg2
    rays = rays0.copy()
    tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["m"].copy())
    g0 = forward_equation(rays, w["K_ne"], tci, 0)
    dobs, CdCt = np.zeros_like(g0), np.full(g0.shape, 1e-4)
    grad0 = compute_gradient(rays, g0, dobs, 0, w["K_ne"], tci, None, CdCt, 1.0, 3, 1.0)
    rays[5, 0, 7, 0, 40] += 3.0                         # one x sample of one ray, in place, same object
    g1 = forward_equation(rays, w["K_ne"], tci, 0)
    assert g1[5, 0, 7] != g0[5, 0, 7]
    assert np.array_equal(g1, _fresh(forward_equation, rays, w["K_ne"], tci, 0))
    grad1 = compute_gradient(rays, g0, dobs, 0, w["K_ne"], tci, None, CdCt, 1.0, 3, 1.0)
    assert np.max(np.abs(grad1 - grad0)) > 0.0
    rays[...] = rays0[:, :, ::-1]                        # a full in-place rewrite
    assert np.array_equal(forward_equation(rays, w["K_ne"], tci, 0), _fresh(forward_equation, rays, w["K_ne"], tci, 0))
    # in-place edit of an AXIS array of the TriCubic is seen too (no identity shortcut on the axes)
    tci.zvec[:] = tci.zvec + 1e-3
    g2 = forward_equation(rays0, w["K_ne"], tci, 0)
    assert np.array_equal(g2, _fresh(forward_equation, rays0, w["K_ne"], tci, 0)) and not np.array_equal(g2, forward_equation(rays0, w["K_ne"], it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["m"]), 0))


def test_cost_of_the_exact_default_is_reported(cfg2):
    """What exactness costs, measured: repeated config-2 calls on the default path against the opt-in."""
    w, rays = cfg2
    tci = it.TriCubic(w["xvec"], w["yvec"], w["zvec"], w["m"].copy())
    out = {}
    for name, kw in (("default_exact", {}), ("assume_unchanged", {"assume_unchanged": True})):
        forward_equation(rays, w["K_ne"], tci, 0, **kw)
        t = []
        for _ in range(30):
            t0 = time.perf_counter()
            forward_equation(rays, w["K_ne"], tci, 0, **kw)
            t.append(time.perf_counter() - t0)
        out[name + "_us"] = round(float(np.median(t)) * 1e6, 1)
    assert out["default_exact_us"] < 5000.0
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/facade_cost.json", "w") as f:
        json.dump(dict(out, workload="config 2: 2604 rays x 129 samples, 128^3 f64 model", note="median of 30 repeated forward_equation calls"), f)
    print("facade cost", out)

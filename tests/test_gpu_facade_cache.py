"""The reference-signature facade keeps its operands resident between calls: forward_equation /
compute_gradient with the SAME rays array re-use the device copy, the node values are recomputed only when the model changed,
in-place changes of the model are seen, a new rays object is uploaded afresh -- and every cached answer equals the uncached
one bit for bit.  Needs a real MI355X: -m gpu."""
import time

import numpy as np
import pytest

import ionotomo_amd as it
from ionotomo_amd import _lib, synthetic as syn
from ionotomo_amd.inversion.forward_equation import forward_equation
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
    g = forward_equation(rays, w["K_ne"], tci, 1)
    rng = np.random.default_rng(0)
    dobs = g + rng.normal(size=g.shape) * 0.01
    CdCt = np.full(g.shape, 1e-4)
    args = (rays, g, dobs, 1, w["K_ne"], tci, None, CdCt, 1.0, 3, 1.0)
    grad = compute_gradient(*args)
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
        return forward_equation(rays, w["K_ne"], it.TriCubic(w["xvec"], w["yvec"], w["zvec"], mm), 1)
    e = np.zeros(m.size)
    e[node] = 1e-3
    e = e.reshape(m.shape)
    for _ in range(30):
        gp, gm = g_of(m + e), g_of(m - e)          # the temporaries die between the calls
        assert np.max(np.abs(gp - gm)) > 0.0
        assert np.array_equal(gp, uncached(g_of, m + e)) and np.array_equal(gm, uncached(g_of, m - e))

"""``Fermat`` -- drop-in for ``ionotomo.inversion.fermat.Fermat`` (inversion/fermat.py:5-174).

``integrate_ray`` returns (x, y, z, s) sampled at ``N`` points of the independent variable:
z from z0 to ``tmax`` (``type='z'``, what every reference call site uses) or arc length from 0 to
``tmax`` (``type='s'``, inversion/fermat.py:74-82,165-166).

* ``straight_line_approx=True``  : closed-form straight ray (n = 1), what every reference call
  site passes (inversion/inversion_pipeline.py:197, astro/simulate_observables.py:62).
* ``straight_line_approx=False`` : as SHIPPED the reference zeroes grad n (fermat.py:54-55), so
  the path stays straight and only the s (type 'z': s = int n/pz dz) or position (type 's':
  x' = p/n) parametrisation changes.  That is the DEFAULT here (``bend=False``, trilinear n like
  ``n_tci.interp``): a drop-in caller gets the shipped reference's rays, pinned to its output in
  tests/golden/fermat_shipped.npz and fermat_type_s.npz.
  ``bend=True`` is opt-in and integrates the true Fermat equations the reference's notebooks specify
  (notebooks/FermatClass.ipynb c0:60-96) with grad n from the ``kind`` interpolant ('linear' or
  'cubic'; default 'cubic' when bending).
The ODE is integrated on the GPU with fixed-step RK4 (``substeps`` steps per output sample)
instead of per-ray LSODA calls.  The reference's integrator is adaptive (``odeint``, LSODA, default
``rtol = atol = 1.49e-8``: inversion/fermat.py:163-167); ``rtol=...`` restores that contract for the
batch: ``choose_substeps`` traces a strided sample of the rays at ``substeps`` and ``2 substeps``
and takes the smallest step count whose step-doubling difference meets the tolerance.
"""
import numpy as np

from .. import _lib

ODEINT_RTOL = 1.49012e-8          # scipy.integrate.odeint's default rtol and atol (what inversion/fermat.py:167 runs with)


def sample_indices(R, fraction=0.01, at_least=208):
    """A strided sample of the batch for the step-doubling estimate: ``fraction`` of the rays, at least ``at_least``."""
    R = int(R)
    n = min(R, max(int(at_least), int(np.ceil(R * fraction))))
    return np.unique(np.linspace(0, R - 1, n).round().astype(np.int64)) if R > 0 else np.zeros(0, dtype=np.int64)


def doubling_error(coarse, fine, rtol, atol):
    """max over rays, samples and components (x, y, z, s) of |coarse - fine| / (rtol |fine| + atol): the
    mixed test LSODA applies per step, here on the whole trajectory.  <= 1: the coarse solution meets the tolerance (its own global
    error is 16/15 of the difference for a 4th-order scheme; on a piecewise-trilinear index, whose gradient jumps at cell faces, the
    observed order is lower and the difference is the safer figure)."""
    c, f = np.asarray(coarse), np.asarray(fine)
    e = np.abs(c - f) / (rtol * np.abs(f) + atol)            # (the independent variable -- z, or s for type 's' -- is sampled exactly: 0)
    return float(e.max()) if e.size else 0.0


def choose_substeps(trace, rtol=ODEINT_RTOL, atol=None, start=1, max_substeps=32, safety=0.5, observe=None):
    """Step-doubling control of the fixed-step RK4 tracer.  ``trace(substeps)`` returns rays[n,4,N] of the SAMPLE rays.  Returns
    (substeps, report): the smallest power-of-two multiple of ``start`` whose trajectory differs from the one at twice as many steps
    by at most ``safety`` x the tolerance (``doubling_error`` <= safety; the factor covers first-order convergence, where the
    difference is only half the coarse solution's error), or ``max_substeps`` with ``report['met'] = False``.  Each level costs one
    launch of the sample; levels already traced are reused as the next level's coarse side.  ``observe(substeps)`` (optional): an
    observable of the sample at that step count -- the TEC along the traced rays -- whose relative change per level is reported too."""
    atol = rtol if atol is None else atol
    s = max(1, int(start))
    coarse = trace(s)
    oc = None if observe is None else np.asarray(observe(s))
    levels = []
    while True:
        fine = trace(2 * s)
        err = doubling_error(coarse, fine, rtol, atol)
        diff = np.abs(np.asarray(coarse) - np.asarray(fine))
        lev = {"substeps": s, "against": 2 * s, "error_over_tolerance": err,
               "max_abs_diff": {k: float(diff[..., i, :].max()) if diff.size else 0.0 for i, k in enumerate("xyzs")}}
        if observe is not None:
            of = np.asarray(observe(2 * s))
            lev["observable_max_rel_diff"] = float(np.max(np.abs(oc - of) / np.abs(of))) if of.size else 0.0
            oc = of
        levels.append(lev)
        if err <= safety or 2 * s > max_substeps:
            met = err <= safety
            break
        s, coarse = 2 * s, fine
    return s, {"rtol": rtol, "atol": atol, "safety": safety, "chosen_substeps": s, "met": met, "levels": levels, "sample_rays": int(np.asarray(coarse).shape[0])}


class Fermat(object):
    def __init__(self, ne_tci, frequency=120e6, type='z', straight_line_approx=True, bend=False, kind=None,
                 substeps=4, rtol=None, atol=None):
        if type not in ('z', 's'):
            raise ValueError("type must be 'z' or 's'")
        self.type = type
        self.frequency = frequency
        self.straight_line_approx = straight_line_approx
        self.bend = bend
        self.kind = kind if kind is not None else ("cubic" if bend else "linear")
        self.substeps = substeps
        self.rtol, self.atol = rtol, atol         # rtol given: ``substeps`` is chosen per batch by step doubling (``choose_substeps``)
        self.step_report = None                   # ... and what the last choice was based on
        self.ne_tci = ne_tci

    def ne2n(self, ne_tci):
        """Refractive index at the nodes, n = sqrt(1 - 8.980^2 ne / nu^2) (inversion/fermat.py:36-46)."""
        n_tci = ne_tci.copy()
        n_tci.M = np.sqrt(1.0 + n_tci.M * (-8.980 ** 2 / self.frequency ** 2))
        return n_tci

    def integrate_rays(self, origins, directions, tmax, N=100):
        """Batched form: origins/directions [..., 3] -> rays [..., 4, N]."""
        o = np.asarray(origins, dtype=np.float64)
        ctx = _lib.default_context()
        if self.straight_line_approx:
            rays = ctx.trace_straight(o, directions, tmax, N, type=self.type)
        else:
            self.ne_tci.bind(ctx)
            substeps = self.substeps
            if self.rtol is not None:
                of, df = o.reshape(-1, 3), np.asarray(directions, dtype=np.float64).reshape(-1, 3)
                idx = sample_indices(of.shape[0])
                substeps, self.step_report = choose_substeps(
                    lambda sub: ctx.trace_fermat(of[idx], df[idx], tmax, N, self.frequency, bend=self.bend, kind=self.kind,
                                                 substeps=sub, type=self.type), self.rtol, self.atol)
            rays = ctx.trace_fermat(o, directions, tmax, N, self.frequency, bend=self.bend, kind=self.kind,
                                    substeps=substeps, type=self.type)
        return rays.reshape(o.shape[:-1] + (4, int(N)))

    def integrate_ray(self, origin, direction, tmax, N=100):
        r = self.integrate_rays(np.asarray(origin)[None, :], np.asarray(direction)[None, :], tmax, N)[0]
        return r[0], r[1], r[2], r[3]

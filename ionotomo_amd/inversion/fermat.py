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
instead of per-ray LSODA calls.
"""
import numpy as np

from .. import _lib


class Fermat(object):
    def __init__(self, ne_tci, frequency=120e6, type='z', straight_line_approx=True, bend=False, kind=None,
                 substeps=4):
        if type not in ('z', 's'):
            raise ValueError("type must be 'z' or 's'")
        self.type = type
        self.frequency = frequency
        self.straight_line_approx = straight_line_approx
        self.bend = bend
        self.kind = kind if kind is not None else ("cubic" if bend else "linear")
        self.substeps = substeps
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
            rays = ctx.trace_fermat(o, directions, tmax, N, self.frequency, bend=self.bend, kind=self.kind,
                                    substeps=self.substeps, type=self.type)
        return rays.reshape(o.shape[:-1] + (4, int(N)))

    def integrate_ray(self, origin, direction, tmax, N=100):
        r = self.integrate_rays(np.asarray(origin)[None, :], np.asarray(direction)[None, :], tmax, N)[0]
        return r[0], r[1], r[2], r[3]

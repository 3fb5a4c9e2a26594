"""``Covariance`` -- the model-covariance operator C_m of ionotomo.ionosphere.covariance that the
inversion applies to the back-projection (ionosphere/covariance.py:8-63,383-385).

The reference's default kernel is the product of three exponential (Matern p = 0, l = 20 km, sigma = 1)
factors, one per axis (covariance.py:22), sampled on a (2h+1)^3 stencil that grows until its corner is
below 5 % of its centre (``create_c_stencil``, :46-63); ``smooth(phi)`` is
``scipy.ndimage.convolve(phi, c_stencil, mode='nearest')``.  Because the stencil is separable the GPU
applies three 1-D passes.  Only this default (separable exponential) kernel is built; the symbolic GP
kernel machinery (utils/gaussian_process.py) is outside the hot path.

``contract(phi)`` = C_m^{-1} phi (used by the prior term of ``neg_log_like(full=True)``,
inversion/iterative_newton.py:49-52).  The reference approximates it with ten sweeps of a gain-0.1 CLEAN
of the truncated stencil over every voxel in sorted order (covariance.py:284-331): a sequential O(N h^3)
Python loop whose result depends on the sweep order (and the truncated stencil is not positive definite,
so it has no stable exact inverse).  Here ``contract`` is the EXACT inverse of the underlying, untruncated
kernel: an exponential covariance on a uniform axis is a first-order Markov (Ornstein-Uhlenbeck) process
whose precision matrix is tridiagonal, (1 - a^2)^-1 sigma^-2 tridiag(-a, 1 + a^2, -a) with 1 at the two
ends and a = exp(-d/l); the separable 3-D inverse is three such 3-point passes (host numpy, O(N)).
DEVIATION, parity unpinned (the reference's CLEAN needs its symbolic-kernel machinery to run at all).
"""
import numpy as np

from .. import _lib


class Covariance(object):
    def __init__(self, K=None, dx=None, dy=None, dz=None, tci=None, l=20.0, sigma=1.0):
        if K is not None:
            raise NotImplementedError("only the reference's default separable exponential kernel is built")
        self.l, self.sigma = float(l), float(sigma)
        self.dx, self.dy, self.dz = dx, dy, dz
        self.tci = tci
        if tci is not None:
            self.dx = tci.xvec[1] - tci.xvec[0]
            self.dy = tci.yvec[1] - tci.yvec[0]
            self.dz = tci.zvec[1] - tci.zvec[0]
        self.h = None
        if self.dx is not None and self.dy is not None and self.dz is not None:
            self.create_c_stencil()

    def create_c_stencil(self):
        """Stencil half width: start at 5 points, add 2 while corner/centre > 0.05 (covariance.py:46-63)."""
        h = 2
        while np.exp(-h * (self.dx + self.dy + self.dz) / self.l) > 0.05:
            h += 1
        self.h = h
        t = np.arange(-h, h + 1)
        # sigma^2 per factor, three factors (MaternPSep(3, d, l, sigma, p=0) for d = 0, 1, 2)
        self.kx = self.sigma ** 2 * np.exp(-np.abs(t * self.dx) / self.l)
        self.ky = self.sigma ** 2 * np.exp(-np.abs(t * self.dy) / self.l)
        self.kz = self.sigma ** 2 * np.exp(-np.abs(t * self.dz) / self.l)

    @property
    def c_stencil(self):
        return self.kx[:, None, None] * self.ky[None, :, None] * self.kz[None, None, :]

    def __call__(self, X, Y=None):
        """Covariance between all pairs of points in X ([M1,3]) and Y ([M2,3])."""
        X = np.asarray(X, dtype=np.float64)
        Y = X if Y is None else np.asarray(Y, dtype=np.float64)
        r = np.abs(X[:, None, :] - Y[None, :, :])
        return self.sigma ** 6 * np.exp(-(r[..., 0] + r[..., 1] + r[..., 2]) / self.l)

    def smooth(self, phi):
        """C_m . phi with the numerical stencil, edge values replicated (covariance.py:383-385)."""
        phi = np.asarray(phi, dtype=np.float64)
        ctx = _lib.default_context()
        if ctx.grid_shape != phi.shape:
            ctx.set_grid(np.arange(phi.shape[0], dtype=float), np.arange(phi.shape[1], dtype=float),
                         np.arange(phi.shape[2], dtype=float), None)
        return ctx.smooth_separable(phi, self.kx, self.ky, self.kz)

    def contract(self, phi):
        """C_m^{-1} . phi: exact inverse of the untruncated separable exponential kernel (module docstring)."""
        out = np.array(phi, dtype=np.float64)
        for axis, d in enumerate((self.dx, self.dy, self.dz)):
            a = np.exp(-abs(d) / self.l)
            x = np.moveaxis(out, axis, 0)
            y = (1.0 + a * a) * x
            y[0] = x[0]
            y[-1] = x[-1]
            y[1:] -= a * x[:-1]
            y[:-1] -= a * x[1:]
            out = np.moveaxis(y / (self.sigma ** 2 * (1.0 - a * a)), 0, axis)
        return np.ascontiguousarray(out)

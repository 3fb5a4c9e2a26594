"""``RayOp`` / ``TECForwardEquation`` -- drop-ins for ionotomo.tomography.linear_operators
(tomography/linear_operators.py:7-98), without TensorFlow: ``matmul`` returns numpy arrays.

    h[i1..ir] = int_R ds M(x) v(x)          (class docstring, :8-19)

``matmul(x)`` = simps(interp(M * x ; rays), dx).  Unlike the reference, which accepts
``adjoint=`` and ignores it, ``matmul(y, adjoint=True)`` applies the exact transpose.
``rays`` is [..., 3, N] (x,y,z); ``dx`` defaults to the cumulative chord length (:26-28).
"""
import numpy as np

from .. import _lib


class RayOp(object):
    def __init__(self, grid, M, rays, dx=None, weight=None, transpose=False, quad="avg", storage="f64"):
        self.grid = tuple(np.asarray(g, dtype=np.float64) for g in grid)
        rays = np.asarray(rays, dtype=np.float64)
        assert rays.shape[-2] == 3
        self.ray_shape = rays.shape[:-2]
        N = rays.shape[-1]
        if dx is None:
            seg = np.sqrt(np.sum(np.square(rays[..., 1:] - rays[..., :-1]), axis=-2))
            s = np.concatenate([np.zeros_like(seg[..., :1]), np.cumsum(seg, axis=-1)], axis=-1)
        else:
            s = np.broadcast_to(np.asarray(dx, dtype=np.float64).reshape((1,) * len(self.ray_shape) + (N,)),
                                self.ray_shape + (N,))
        self.rays4 = np.ascontiguousarray(np.concatenate([rays, s[..., None, :]], axis=-2))
        self.weight = None if weight is None else np.asarray(weight, dtype=np.float64).reshape(self.ray_shape)
        self.M = np.asarray(M, dtype=np.float64)
        self.transpose = transpose
        self.quad = quad
        self.storage = storage

    def domain_shape(self):
        return self.M.shape

    def range_shape(self):
        return self.ray_shape

    def shape(self):
        return tuple(self.ray_shape) + tuple(self.M.shape)

    def _bind(self, values):
        ctx = _lib.default_context()
        ctx.set_grid(self.grid[0], self.grid[1], self.grid[2], values, storage=self.storage)
        return ctx

    def matmul(self, x, adjoint=False, adjoint_arg=False):
        if adjoint != self.transpose:
            y = np.asarray(x, dtype=np.float64).reshape(self.ray_shape)
            if self.weight is not None:
                y = y * self.weight
            ctx = self._bind(self.M)
            return ctx.adjoint_rays(self.rays4, y, rule=self.quad, scale_by_grid=True)
        ctx = self._bind(self.M * np.asarray(x, dtype=np.float64))
        Ax = ctx.forward_tec_rays(self.rays4, rule=self.quad)
        if self.weight is not None:
            Ax = self.weight * Ax
        return Ax


class TECForwardEquation(RayOp):
    def __init__(self, i0, grid, M, rays, dx=None, weight=None, transpose=False, **kw):
        super(TECForwardEquation, self).__init__(grid, M, rays, dx, weight, transpose, **kw)
        self.i0 = int(i0)

    def matmul(self, x, adjoint=False, adjoint_arg=False):
        if adjoint != self.transpose:
            y = np.array(x, dtype=np.float64).reshape(self.ray_shape)
            y[self.i0] -= y.sum(axis=0)           # transpose of  Ax - Ax[i0]
            return super(TECForwardEquation, self).matmul(y, adjoint=adjoint)
        Ax = super(TECForwardEquation, self).matmul(x)
        return Ax - Ax[self.i0:self.i0 + 1, ...]   # tomography/linear_operators.py:96-97

"""Back-projection gradient -- drop-in for ionotomo.inversion.gradient (inversion/gradient.py:15-102).

DEVIATION (documented in SURVEY.md 8a A7/A7' and DESIGN.md): the reference builds a voxel
chord-length tensor ``dirac[N1,N2,nx,ny,nz]`` (geometry/ray_dirac.py) -- a different
discretisation from its own forward model, and infeasible beyond toy grids (1 GB per direction at
128^3).  This build returns the EXACT transpose of the forward operator (trilinear weights x
Simpson weights), which is what an optimiser needs: d/dm of
S = 1/2 sum (g - dobs)^2 / (CdCt + 1e-15) with g = forward_equation(rays, K_ne, m_tci, i0).
The reference's i0 differencing of the gradient is commented out (gradient.py:18); here the
differencing is applied (it is part of g).  ``sigma_m, Nkernel, size_cell, cov_obj`` are accepted
and unused, as in the reference body (its prior term is computed and discarded, :96-97).
"""
import numpy as np

from .. import _lib
from .forward_equation import TECU


def differential_weights(dd, i0):
    w = np.array(dd, dtype=np.float64)
    w[i0] -= w.sum(axis=0)
    return w


def compute_gradient(rays, g, dobs, i0, K_ne, m_tci, m_prior, CdCt, sigma_m, Nkernel, size_cell, cov_obj=None,
                     quad="avg", method="transpose", assume_unchanged=False):
    """``method="transpose"`` (default): the exact transpose of the forward model (module docstring).
    ``method="chords"``: the reference's own discretisation, ``do_gradient`` = einsum(dirac, ne, dd) over voxel chord
    lengths (inversion/gradient.py:15-20, geometry/ray_dirac.py), pinned to the reference's output in
    tests/golden/ray_dirac.npz -- for comparison with the shipped code, not for optimisation.  (The reference then
    subtracts ``gradient[i0, ...]``, i.e. indexes the GRID's first axis with an antenna index (:62); that slip is not
    reproduced.)

    ``rays`` and ``m_tci.M`` are uploaded on every call (exact against in-place edits) unless ``assume_unchanged=True``
    (opt-in, same contract as ``forward_equation``: operands keyed on the array OBJECTS, the rays ``forward_equation`` left
    on the device are re-used)."""
    rays = np.asarray(rays, dtype=np.float64)
    dd = g - dobs
    dd /= (CdCt + 1e-15)                       # inversion/gradient.py:77-81
    ctx = _lib.default_context()
    ctx.set_grid(m_tci.xvec, m_tci.yvec, m_tci.zvec, None, storage=m_tci.storage)
    if assume_unchanged:
        ctx.set_values_exp_cached(m_tci.M, K_ne / TECU)
    else:
        ctx.set_values_exp(m_tci.M, K_ne / TECU)
    if method == "chords":
        return ctx.gradient_chords(rays, dd)
    w = np.ascontiguousarray(differential_weights(dd, i0).reshape(-1))
    rays_c = np.ascontiguousarray(rays)
    rays_dev = ctx.resident(rays_c) if (assume_unchanged and rays_c is rays) else None
    if rays_dev is None:
        rays_dev = ctx.staged("rays", rays_c)
    n = int(np.prod(ctx.grid_shape))
    w_dev, g_dev = ctx.scratch("w", w.nbytes), ctx.scratch("grad", n * 8)
    ctx.call("iono_dev_upload", w_dev, _lib._V(w.ctypes.data), w.nbytes)
    ctx.call("iono_dev_zero", g_dev, n * 8)
    ctx.call("iono_adjoint_rays_dev", rays_dev, w_dev, w.size, int(rays_c.shape[-1]), _lib.interp_kind(m_tci.kind),
             _lib.quad_rule(quad), g_dev, _lib.F64)
    ctx.call("iono_scale_by_grid_dev", g_dev)
    grad = np.empty(ctx.grid_shape, dtype=np.float64)
    ctx.call("iono_dev_download", _lib._V(grad.ctypes.data), g_dev, n * 8)
    return grad


compute_gradient_dask = compute_gradient

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
                     quad="avg", method="transpose"):
    """``method="transpose"`` (default): the exact transpose of the forward model (module docstring).
    ``method="chords"``: the reference's own discretisation, ``do_gradient`` = einsum(dirac, ne, dd) over voxel chord
    lengths (inversion/gradient.py:15-20, geometry/ray_dirac.py), pinned to the reference's output in
    tests/golden/ray_dirac.npz -- for comparison with the shipped code, not for optimisation.  (The reference then
    subtracts ``gradient[i0, ...]``, i.e. indexes the GRID's first axis with an antenna index (:62); that slip is not
    reproduced.)"""
    rays = np.asarray(rays, dtype=np.float64)
    dd = g - dobs
    dd /= (CdCt + 1e-15)                       # inversion/gradient.py:77-81
    ctx = _lib.default_context()
    ctx.set_grid(m_tci.xvec, m_tci.yvec, m_tci.zvec, None, storage=m_tci.storage)
    ctx.set_values_exp(m_tci.M, K_ne / TECU)
    if method == "chords":
        return ctx.gradient_chords(rays, dd)
    return ctx.adjoint_rays(rays, differential_weights(dd, i0), rule=quad, scale_by_grid=True, kind=m_tci.kind)


compute_gradient_dask = compute_gradient

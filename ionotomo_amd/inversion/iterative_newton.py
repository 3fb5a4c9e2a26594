"""Phase forward model + objective -- the hot-path part of ionotomo.inversion.iterative_newton
(inversion/iterative_newton.py:17-55,86-127).

NOTE on parity: the reference's ``forward_equation`` here passes 4-D coordinate arrays to
``TriCubic.interp``, whose ``np.reshape(rgi(np.array([x,y,z]).T), np.shape(x))``
(geometry/tri_cubic.py:70) returns the values in transposed order for ndim > 1 -- every sample
lands on the wrong ray.  This build implements the evidently intended semantics by default (sample k
of ray (a,t,d) stays with that ray); ``forward_equation(..., compat="reference")`` reproduces the
reference's ACTUAL output bit-for-formula (the same permutation applied to the sample coordinates,
path lengths untouched) and is pinned to the golden vector directly on the GPU
(tests/test_gpu_parity.py::test_phase_forward_reference_compat_matches_the_golden_directly).
"""
import numpy as np

from .. import _lib

TECU = 1e13
speedoflight = 299792458.


def forward_equation(model, tci, rays, freqs, K=1e11, i0=0, quad="avg", compat=None):
    """g[Na,Nt,Nd,Nf] = const_i + 2 pi nu clock_ij - (2 pi nu / c) [int (1-n) ds - ref]
    (inversion/iterative_newton.py:86-127).  Like the reference it leaves ``tci.M = K exp(mu)``.

    ``compat="reference"``: the output of the reference AS SHIPPED -- its ``TriCubic.interp`` hands the 4-D coordinate arrays
    to scipy transposed and reshapes the values back without transposing (geometry/tri_cubic.py:70, called from
    iterative_newton.py:108), so the electron density integrated along ray (a, t, d) at sample k is the one found at the
    sample whose flat index in [N, Nd, Nt, Na] order equals the flat index of (a, t, d, k) in [Na, Nt, Nd, N] order, while the
    path lengths ``s`` stay in place.  Same kernels, coordinates permuted on the host."""
    rays = np.asarray(rays, dtype=np.float64)
    if compat == "reference":
        if rays.ndim != 5:
            raise ValueError("compat='reference' needs rays[Na,Nt,Nd,4,N]")
        rays = rays.copy()
        shp = rays[..., 0, :].shape
        for c in range(3):
            rays[..., c, :] = np.reshape(rays[..., c, :].transpose(3, 2, 1, 0), shp)
    elif compat is not None:
        raise ValueError("compat must be None or 'reference'")
    mu, clock, const = model
    ne = np.exp(mu)
    ne *= K
    tci.M = ne                                   # reference side effect (:106)
    ctx = tci.bind()
    return ctx.forward_phase_rays(rays, freqs, clock, const, i0, rule=quad)


def neg_log_like(g, dobs, CdCt, covariance=None, model=None, model_prior=None, tci=None, full=False):
    """S = 1/2 sum (dobs - g)^2 / CdCt, plus with ``full=True`` the prior terms
    1/2 sum (clock - clock_prior)^2 / c_clock + 1/2 <dmu, C_mu^{-1} dmu> with the Simpson^3 inner product
    of ``tci`` (inversion/iterative_newton.py:17-54).  ``covariance = (Covariance, c_clock)``,
    ``model = (mu, clock, const)``.  C_mu^{-1} is ``Covariance.contract`` (see its deviation note).
    Like the reference, ``full=True`` leaves ``tci.M = C_mu^{-1} dmu``."""
    dd = dobs - g
    l2 = float(np.sum(dd * dd / CdCt) / 2.)
    if full:
        c_mu, c_clock = covariance
        mu, clock, const = model
        mu_prior, clock_prior, const_prior = model_prior
        dclock = clock - clock_prior
        l2 += float(np.sum(dclock * dclock / c_clock) / 2.)
        dmu = np.reshape(mu_prior - mu, (tci.nx, tci.ny, tci.nz))
        tci.M = c_mu.contract(dmu)
        l2 += tci.inner(dmu, inplace=False) / 2.
    return l2

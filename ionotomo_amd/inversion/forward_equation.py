"""dTEC forward model -- drop-in for ionotomo.inversion.forward_equation (inversion/forward_equation.py:13-67).

``forward_equation(rays, K_ne, m_tci, i0)``: ne = K_ne exp(m)/TECU at the nodes, TEC = Simpson
integral of the interpolated ne along every ray, dTEC = TEC - TEC[i0].  One GPU launch per call
(exp at nodes) + one (all rays) + one (reference-antenna subtraction) instead of the reference's
per-ray Python loop.

``quad`` pins the quadrature rule for EVEN sample counts -- which is the reference's default, since
``calc_rays`` takes ``N = ne_tci.nz`` (SURVEY.md surprise 3); for odd counts all Simpson variants
coincide.
* ``quad='avg'`` (DEFAULT): ``scipy.integrate.simps(y, s)`` as it behaved when the reference was
  written (scipy <= 1.10: even='avg'), i.e. what the reference computes on the scipy it was
  developed against.  Pinned by tests/golden/forward_tec_even_avg.npz (the reference module run with
  that rule).
* ``quad='scipy'``: what ``simps`` resolves to on a current scipy (>= 1.11 ``simpson``, Cartwright
  end correction), i.e. what the unmodified reference computes when run TODAY.  Pinned by
  tests/golden/forward_tec.npz.  The two differ by ~2e-6 relative at N = 64 on the cfg1 field.
Pick 'scipy' to compare against a reference run on a modern stack.
"""
import numpy as np

from .. import _lib

TECU = 1e13    # inversion/forward_equation.py:12


def do_forward_equation(rays, ne_tci, quad="avg"):
    """tec[N1,N2] for rays[N1,N2,4,Ns] through ``ne_tci`` as it stands (forward_equation.py:13-33)."""
    ctx = ne_tci.bind()
    return ctx.forward_tec_rays(rays, kind=ne_tci.kind, rule=quad)


def forward_equation(rays, K_ne, m_tci, i0, quad="avg", assume_unchanged=False):
    """dtec[Na,Nt,Nd] using reference antenna ``i0`` (forward_equation.py:36-51).

    DEFAULT (exact): ``rays`` and ``m_tci.M`` are uploaded on EVERY call -- whatever the caller did to them since the last
    call, in place or not, is seen (the reference's own finite-difference loop perturbs one node of ``m_tci.M`` in place and
    calls again: tests/test_inversion.py:81-82).  Only the device buffers are kept between calls; the [Na,Nt,Nd] result
    comes back through pinned memory.

    ``assume_unchanged=True`` (opt-in, for a line search that passes the SAME ``rays`` object again and again:
    inversion/line_search.py:56,71,83): the caller promises that arrays handed over before have not been edited IN PLACE
    since.  ``rays`` is then keyed on the array OBJECT (a weak reference + shape, no content check: ``Context.resident``)
    and the node values are recomputed only when the model object, the scale or a sampled fingerprint of it changed
    (``Context.set_values_exp_cached``) -- a single-node in-place edit is NOT seen on this path.  For device-resident
    inversion loops use ``engine.RayEngine`` instead."""
    rays = np.asarray(rays, dtype=np.float64)
    rays_c = np.ascontiguousarray(rays)
    Na, Nt, Nd, _, Ns = rays_c.shape
    ctx = _lib.default_context()
    ctx.set_grid(m_tci.xvec, m_tci.yvec, m_tci.zvec, None, storage=m_tci.storage)
    if assume_unchanged:
        ctx.set_values_exp_cached(m_tci.M, K_ne / TECU)
        rays_dev = ctx.resident(rays_c) if rays_c is rays else None    # (a converted copy has no identity to key on)
    else:
        ctx.set_values_exp(m_tci.M, K_ne / TECU)
        rays_dev = None
    if rays_dev is None:
        rays_dev = ctx.staged("rays", rays_c)
    R = Na * Nt * Nd
    tec_dev = ctx.scratch("tec", R * 8)
    ctx.call("iono_forward_tec_rays_dev", rays_dev, R, int(Ns), _lib.interp_kind(m_tci.kind), _lib.quad_rule(quad), tec_dev)
    ctx.call("iono_subtract_reference_dev", tec_dev, Na, Nt * Nd, int(i0))
    tec = np.empty((Na, Nt, Nd), dtype=np.float64)
    ctx.call("iono_dev_download", _lib._V(tec.ctypes.data), tec_dev, R * 8)
    return tec


def forward_equation_dask(rays, K_ne, m_tci, i0, quad="avg", assume_unchanged=False):
    """The reference's dask-multiprocessing variant computes the same numbers
    (tests/test_forward_equation.py:27 asserts exact equality); one GPU needs no task split."""
    return forward_equation(rays, K_ne, m_tci, i0, quad=quad, assume_unchanged=assume_unchanged)

"""``calc_rays`` / ``cast_ray`` -- drop-ins for ionotomo.geometry.calc_rays (geometry/calc_rays.py:61-145).

Output layout is the reference's: ``rays[Na, Nt, Nd, 4, N]`` = x, y, z, s in km.

The reference's ``calc_rays`` takes astropy objects (ITRS antennas, ICRS patches, Time) and turns
them into model-frame origins/directions with its ``Pointing`` frame (:122-139).  astropy is a
host-side coordinate library outside the hot path (SURVEY.md 8f "next" #1); here the first two
arguments may be given directly as model-frame arrays -- ``antennas`` [Na,3] km and ``patches``
[Nd,3] or [Nt,Nd,3] direction vectors -- or as (origins, directions) [Na,Nt,Nd,3] via ``cast_ray``.
"""
import numpy as np

from ..inversion.fermat import Fermat


def cast_ray(batch, fermat, tmax, N):
    """rays[Na,Nt,Nd,4,N] for ``batch = (origins, directions)``, both [Na,Nt,Nd,3]
    (geometry/calc_rays.py:61-96).  One GPU launch instead of Na*Nt*Nd ODE solves."""
    origins, directions = batch
    origins = np.asarray(origins, dtype=np.float64)
    assert origins.ndim == 4 and origins.shape[-1] == 3
    return fermat.integrate_rays(origins, directions, tmax, N)


def model_frame_bundle(antennas, patches, times=None):
    """origins, directions [Na,Nt,Nd,3] from antennas [Na,3] (km) and unit-ish direction vectors
    patches [Nd,3] (same for every time) or [Nt,Nd,3] (geometry/calc_rays.py:122-139 layout)."""
    ants = np.asarray(antennas, dtype=np.float64)
    dirs = np.asarray(patches, dtype=np.float64)
    if dirs.ndim == 2:
        nt = 1 if times is None else len(times)
        dirs = np.broadcast_to(dirs[None], (nt,) + dirs.shape)
    na, (nt, nd, _) = ants.shape[0], dirs.shape
    origins = np.broadcast_to(ants[:, None, None, :], (na, nt, nd, 3)).copy()
    directions = np.broadcast_to(dirs[None], (na, nt, nd, 3)).copy()
    return origins, directions


def calc_rays(antennas, patches, times, array_center, fixtime, phase, ne_tci, frequency, straight_line_approx, tmax,
              N=None, **fermat_kwargs):
    """Same signature as geometry/calc_rays.py:109.  Two input conventions:
    * ``array_center is None``: ``antennas``/``patches`` are model-frame arrays (module docstring);
    * ``array_center`` (ITRS [3] m) and ``phase`` ((ra, dec) rad) given: ``antennas`` are ITRS [Na,3] m,
      ``patches`` (ra, dec) [Nd,2] rad, ``times`` UTC unix seconds -- transformed to the Pointing frame
      per observation time as the reference does (plain arrays instead of astropy objects)."""
    if N is None:
        N = ne_tci.nz
    if hasattr(antennas, "transform_to") or hasattr(patches, "transform_to"):
        raise NotImplementedError(
            "astropy coordinate objects are not transformed here (astropy is not part of this build); pass "
            "model-frame arrays: antennas [Na,3] km, patches [Nd,3] or [Nt,Nd,3] direction vectors")
    if array_center is not None and phase is not None:
        # ITRS antennas [Na,3] m, (ra, dec) patches [Nd,2] rad, UTC unix times, centre ITRS [3] m,
        # phase centre (ra, dec): the reference's Pointing-frame set-up without astropy (astro/frames.py)
        from ..astro.frames import model_frame_bundle_from_sky
        origins, directions = model_frame_bundle_from_sky(antennas, patches, times, array_center, phase, fixtime)
    else:
        origins, directions = model_frame_bundle(antennas, patches, times)
    fermat = Fermat(ne_tci=ne_tci, frequency=frequency, type='z', straight_line_approx=straight_line_approx,
                    **fermat_kwargs)
    return cast_ray((origins, directions), fermat, tmax, N)


def calc_rays_dask(*args, **kwargs):
    """The reference's ``calc_rays_dask`` is a stub (``pass``, geometry/calc_rays.py:105-107)."""
    kwargs.pop("get", None)
    return calc_rays(*args, **kwargs)

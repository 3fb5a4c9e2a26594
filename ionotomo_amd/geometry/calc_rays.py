"""``calc_rays`` / ``cast_ray`` -- drop-ins for ionotomo.geometry.calc_rays (geometry/calc_rays.py:61-145).

Output layout is the reference's: ``rays[Na, Nt, Nd, 4, N]`` = x, y, z, s in km.

The reference's ``calc_rays`` takes astropy objects (ITRS antennas, ICRS patches, Time) and turns
them into model-frame origins/directions with its ``Pointing`` frame (:122-139).  Such objects are
accepted as they are: their numbers are read by attribute (astro/coords.py; astropy itself is never
imported) and the frame rotation is astro/frames.py.  The first two arguments may also be given
directly as model-frame arrays -- ``antennas`` [Na,3] km and ``patches`` [Nd,3] or [Nt,Nd,3] direction
vectors -- or as (origins, directions) [Na,Nt,Nd,3] via ``cast_ray``.
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
    """Same signature as geometry/calc_rays.py:109.  Three input conventions:
    * the reference's own: astropy-typed arguments -- ``antennas`` ITRS coordinates, ``patches`` ICRS coordinates, ``times`` /
      ``fixtime`` Time objects, ``array_center`` an ITRS coordinate with ``.earth_location``, ``phase`` an ICRS coordinate -- as
      inversion/inversion_pipeline.py:195-197 and astro/simulate_observables.py:62 pass them.  Their numbers are read by attribute
      (``.cartesian.xyz``, ``.ra`` / ``.dec``, ``.unix`` / ``.gps``: astro/coords.py) without importing astropy, then treated as
      the next case;
    * ``array_center`` (ITRS [3] m) and ``phase`` ((ra, dec) rad) given as arrays: ``antennas`` are ITRS [Na,3] m,
      ``patches`` (ra, dec) [Nd,2] rad, ``times`` UTC unix seconds -- transformed to the Pointing frame
      per observation time as the reference does (:122-139);
    * ``array_center is None``: ``antennas``/``patches`` are model-frame arrays (module docstring)."""
    from ..astro import coords
    if N is None:
        N = ne_tci.nz
    if array_center is not None and phase is not None:
        # ITRS antennas [Na,3] m, (ra, dec) patches [Nd,2] rad, UTC unix times, centre ITRS [3] m,
        # phase centre (ra, dec): the reference's Pointing-frame set-up without astropy (astro/frames.py)
        from ..astro.frames import model_frame_bundle_from_sky
        origins, directions = model_frame_bundle_from_sky(
            coords.itrs_metres(antennas), coords.icrs_radec(patches), np.atleast_1d(coords.unix_seconds(times)),
            coords.itrs_metres(array_center).reshape(3), coords.icrs_radec(phase).reshape(2),
            None if fixtime is None else float(np.asarray(coords.unix_seconds(fixtime)).reshape(-1)[0]))
    else:
        if coords.is_coordinate(antennas) or coords.is_coordinate(patches):
            raise ValueError("sky coordinates need `array_center` and `phase` to define the model (Pointing) frame "
                             "(geometry/calc_rays.py:124-125)")
        origins, directions = model_frame_bundle(antennas, patches, times)
    fermat = Fermat(ne_tci=ne_tci, frequency=frequency, type='z', straight_line_approx=straight_line_approx,
                    **fermat_kwargs)
    return cast_ray((origins, directions), fermat, tmax, N)


def calc_rays_dask(*args, **kwargs):
    """The reference's ``calc_rays_dask`` is a stub (``pass``, geometry/calc_rays.py:105-107)."""
    kwargs.pop("get", None)
    return calc_rays(*args, **kwargs)

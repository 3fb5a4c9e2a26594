"""Model-frame ("Pointing") ray set-up without astropy -- SURVEY.md 8f #1.

The reference turns ITRS antenna positions and ICRS facet directions into its ``Pointing`` frame with
astropy (astro/frames/pointing_frame.py:140-190; geometry/calc_rays.py:122-139) and then sizes the
inversion box (inversion/initial_model.py:13-36).  The rotation itself is plain linear algebra:

    ha = LST - RA_phase,   R = [east, north, up] evaluated at (longitude - ha, latitude = Dec_phase)
    p_pointing = R (p_itrs - p_centre)                                  (pointing_frame.py:149-186)

so the frame's w axis points at the phase centre, u is "east" there and v "north".  What astropy adds
on top is the Earth-orientation chain (precession, nutation, polar motion, UT1-UTC) inside
``sidereal_time`` and inside the ICRS -> ITRS step for the directions.  Here both use the IAU-1982 GMST
polynomial and a pure rotation about the pole -- good to ~1e-4 rad in absolute pointing, while facet
offsets RELATIVE to the phase centre (what the ray fan depends on) are unaffected at that level.
PARITY UNPINNED against the reference (astropy is not installed in the build image); the axis
conventions the reference's own tests pin (tests/test_frames.py:40-164, cases a-d) are restated in
tests/test_frames_conventions.py.  Host-side numpy, runs once per timestep: not on the hot path.
"""
import numpy as np

from ..synthetic import enu_rotation

WGS84_A, WGS84_F = 6378137.0, 1.0 / 298.257223563


def geodetic_from_itrs(xyz_m):
    """WGS-84 longitude, latitude [rad] and height [m] of an ITRS point (Bowring's closed form)."""
    x, y, z = np.asarray(xyz_m, dtype=np.float64)
    a, f = WGS84_A, WGS84_F
    b = a * (1 - f)
    e2, ep2 = 1 - (b / a) ** 2, (a / b) ** 2 - 1
    p = np.hypot(x, y)
    th = np.arctan2(z * a, p * b)
    lat = np.arctan2(z + ep2 * b * np.sin(th) ** 3, p - e2 * a * np.cos(th) ** 3)
    lon = np.arctan2(y, x)
    N = a / np.sqrt(1 - e2 * np.sin(lat) ** 2)
    return lon, lat, p / np.cos(lat) - N


def gmst_rad(unix_utc):
    """Greenwich mean sidereal time [rad] from UTC seconds since 1970 (IAU 1982 polynomial, UT1 ~ UTC)."""
    jd = np.asarray(unix_utc, dtype=np.float64) / 86400.0 + 2440587.5
    d = jd - 2451545.0
    T = d / 36525.0
    gmst_deg = 280.46061837 + 360.98564736629 * d + 0.000387933 * T ** 2 - T ** 3 / 38710000.0
    return np.deg2rad(gmst_deg % 360.0)


def pointing_rotation(lon, lst, ra, dec):
    """R = [east, north, up] at (lon - (lst - ra), dec): rows are the Pointing axes u, v, w in ITRS
    (astro/frames/pointing_frame.py:149-165)."""
    return enu_rotation(lon - (lst - ra), dec)


def itrs_to_pointing_km(xyz_m, centre_m, R):
    """Positions [N,3] ITRS metres -> Pointing-frame km about ``centre_m`` (pointing_frame.py:175-186)."""
    return (np.asarray(xyz_m, dtype=np.float64) - np.asarray(centre_m, dtype=np.float64)) @ R.T / 1000.0


def icrs_to_itrs_direction(ra, dec, gmst):
    """Unit vectors in ITRS for equatorial (ra, dec) [rad] at Greenwich sidereal angle ``gmst``."""
    ra, dec = np.asarray(ra, dtype=np.float64), np.asarray(dec, dtype=np.float64)
    lon = ra - gmst
    return np.stack([np.cos(dec) * np.cos(lon), np.cos(dec) * np.sin(lon), np.sin(dec)], axis=-1)


def model_frame_bundle_from_sky(antennas_itrs_m, patches_radec, times_unix, centre_itrs_m, phase_radec, fixtime_unix=None):
    """origins, directions [Na,Nt,Nd,3] in the Pointing frame (km / unit vectors) from ITRS antennas,
    (ra, dec) facet directions [Nd,2] and UTC times -- the coordinate part of geometry/calc_rays.py:122-139.
    As in the reference the frame is re-built for every observation time (obstime drives the LST);
    ``fixtime`` is accepted for signature compatibility (the reference stores but does not use it in the
    rotation)."""
    ants = np.asarray(antennas_itrs_m, dtype=np.float64)
    pat = np.asarray(patches_radec, dtype=np.float64)
    times = np.atleast_1d(np.asarray(times_unix, dtype=np.float64))
    lon, _, _ = geodetic_from_itrs(centre_itrs_m)
    na, nt, nd = ants.shape[0], times.size, pat.shape[0]
    origins = np.empty((na, nt, nd, 3))
    directions = np.empty((na, nt, nd, 3))
    for j, t in enumerate(times):
        g = gmst_rad(t)
        R = pointing_rotation(lon, g + lon, phase_radec[0], phase_radec[1])
        origins[:, j, :, :] = itrs_to_pointing_km(ants, centre_itrs_m, R)[:, None, :]
        directions[:, j, :, :] = (icrs_to_itrs_direction(pat[:, 0], pat[:, 1], g) @ R.T)[None, :, :]
    return origins, directions


def determine_inversion_domain(spacing, antennas_km, directions, zmax, padding=20):
    """Axis vectors of the inversion box around every straight ray up to height ``zmax`` plus
    ``padding`` cells (inversion/initial_model.py:13-36).  ``antennas_km`` [Na,3], ``directions`` [Nd,3]
    are already in the model frame."""
    ants = np.asarray(antennas_km, dtype=np.float64)
    dirs = np.asarray(directions, dtype=np.float64)
    ends = [np.add.outer(ants[:, a], dirs[:, a] * zmax / dirs[:, 2]) for a in range(3)]
    vecs = []
    for a in range(3):
        lo = min(ants[:, a].min(), ends[a].min()) - spacing * padding
        hi = max(ants[:, a].max(), ends[a].max()) + spacing * padding
        vecs.append(np.linspace(lo, hi, int(np.ceil((hi - lo) / spacing))))
    return vecs

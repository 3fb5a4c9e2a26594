"""Model-frame ("Pointing") ray set-up without astropy -- SURVEY.md 8f #1.

The reference turns ITRS antenna positions and ICRS facet directions into its ``Pointing`` frame with
astropy (astro/frames/pointing_frame.py:140-190; geometry/calc_rays.py:122-139) and then sizes the
inversion box (inversion/initial_model.py:13-36).  The rotation itself is plain linear algebra:

    ha = LST - RA_phase,   R = [east, north, up] evaluated at (longitude - ha, latitude = Dec_phase)
    p_pointing = R (p_itrs - p_centre)                                  (pointing_frame.py:149-186)

so the frame's w axis points at the direction whose OF-DATE right ascension / declination equal the phase centre's ICRS
numbers (the reference subtracts an ICRS right ascension from a mean sidereal time of date: pointing_frame.py:151-152 --
reproduced as is), u is "east" there and v "north".  What astropy adds on top is the Earth-orientation chain inside the
ICRS -> ITRS step for the facet directions (geometry/calc_rays.py:122-139).  Here that chain is

    r_itrs = R3(GAST) N(t) P(t) r_icrs

with the IAU-1976 precession matrix P (Lieske angles), the IAU-1980 nutation N truncated to its four largest terms
(0.05 arc-seconds), GAST = GMST82 + dpsi cos(eps) and UT1 = UTC, no polar motion, no frame bias.  Pinned to the published
IAU SOFA check values (t_sofa_c.c: iauGmst82, iauPmat76, iauNut80, iauObl80, iauGd2gc, iauGc2gd) in
tests/test_frames_conventions.py.  Residual against a full IAU-2006/2000A chain with IERS data: UT1-UTC (<= 0.9 s = 13.5
arc-seconds of Earth rotation, not knowable offline), polar motion (<= 0.3"), IAU-1976 vs 2006 precession (0.08" by 2026),
truncated nutation (0.05"), frame bias (0.02"): <= 14" = 7e-5 rad in absolute direction, common to every facet; a ray at 5
degrees from the zenith changes its slant factor by tan(z) x 7e-5 = 6e-6.  PARITY otherwise UNPINNED against the reference
(astropy is not installed in the build image); the axis conventions the reference's own tests pin (tests/test_frames.py:40-164,
cases a-d) are restated in tests/test_frames_conventions.py.  Host-side numpy, runs once per timestep: not on the hot path.
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


ARCSEC = np.pi / 180.0 / 3600.0


def _r1(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, s], [0.0, -s, c]])


def _r2(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0.0, -s], [0.0, 1.0, 0.0], [s, 0.0, c]])


def _r3(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])


def julian_date(unix_utc):
    return float(unix_utc) / 86400.0 + 2440587.5


def precession_matrix(jd):
    """Mean equator and equinox of J2000.0 -> of date: IAU 1976 (Lieske et al. 1977 angles; = iauPmat76)."""
    t = (jd - 2451545.0) / 36525.0
    zeta = (2306.2181 * t + 0.30188 * t ** 2 + 0.017998 * t ** 3) * ARCSEC
    z = (2306.2181 * t + 1.09468 * t ** 2 + 0.018203 * t ** 3) * ARCSEC
    theta = (2004.3109 * t - 0.42665 * t ** 2 - 0.041833 * t ** 3) * ARCSEC
    return _r3(-z) @ _r2(theta) @ _r3(-zeta)


def mean_obliquity(jd):
    """IAU 1980 mean obliquity of the ecliptic [rad] (= iauObl80)."""
    t = (jd - 2451545.0) / 36525.0
    return (84381.448 - 46.8150 * t - 0.00059 * t ** 2 + 0.001813 * t ** 3) * ARCSEC


def nutation(jd):
    """(dpsi, deps) [rad]: the four largest terms of the IAU 1980 series (0.05 arc-seconds against iauNut80)."""
    t = (jd - 2451545.0) / 36525.0
    om = np.deg2rad(125.04452 - 1934.136261 * t)
    ls = np.deg2rad(280.4665 + 36000.7698 * t)
    lm = np.deg2rad(218.3165 + 481267.8813 * t)
    dpsi = (-17.20 * np.sin(om) - 1.32 * np.sin(2 * ls) - 0.23 * np.sin(2 * lm) + 0.21 * np.sin(2 * om)) * ARCSEC
    deps = (9.20 * np.cos(om) + 0.57 * np.cos(2 * ls) + 0.10 * np.cos(2 * lm) - 0.09 * np.cos(2 * om)) * ARCSEC
    return dpsi, deps


def icrs_to_itrs_matrix(unix_utc):
    """Celestial (ICRS ~ mean J2000) -> terrestrial rotation at a UTC instant: R3(GAST) N P (module docstring for what is left out)."""
    jd = julian_date(unix_utc)
    eps = mean_obliquity(jd)
    dpsi, deps = nutation(jd)
    N = _r1(-(eps + deps)) @ _r3(-dpsi) @ _r1(eps)
    gast = float(gmst_rad(unix_utc)) + dpsi * np.cos(eps)
    return _r3(gast) @ N @ precession_matrix(jd)


def icrs_direction_in_itrs(ra, dec, unix_utc):
    """Unit vectors in ITRS of ICRS (ra, dec) [rad] at a UTC instant (what astropy's ICRS -> ITRS does to a direction)."""
    ra, dec = np.asarray(ra, dtype=np.float64), np.asarray(dec, dtype=np.float64)
    r = np.stack([np.cos(dec) * np.cos(ra), np.cos(dec) * np.sin(ra), np.sin(dec)], axis=-1)
    return r @ icrs_to_itrs_matrix(unix_utc).T


def itrs_direction_to_icrs(v_itrs, unix_utc):
    """(ra, dec) [N,2] in ICRS of ITRS direction vectors [N,3] at a UTC instant: the inverse of ``icrs_direction_in_itrs``."""
    v = np.atleast_2d(np.asarray(v_itrs, dtype=np.float64)) @ icrs_to_itrs_matrix(unix_utc)      # rows: M^T v
    v = v / np.linalg.norm(v, axis=1)[:, None]
    return np.stack([np.arctan2(v[:, 1], v[:, 0]) % (2 * np.pi), np.arcsin(np.clip(v[:, 2], -1, 1))], axis=-1)


def pointing_rotation(lon, lst, ra, dec):
    """R = [east, north, up] at (lon - (lst - ra), dec): rows are the Pointing axes u, v, w in ITRS
    (astro/frames/pointing_frame.py:149-165)."""
    return enu_rotation(lon - (lst - ra), dec)


def itrs_to_pointing_km(xyz_m, centre_m, R):
    """Positions [N,3] ITRS metres -> Pointing-frame km about ``centre_m`` (pointing_frame.py:175-186)."""
    return (np.asarray(xyz_m, dtype=np.float64) - np.asarray(centre_m, dtype=np.float64)) @ R.T / 1000.0


def icrs_to_itrs_direction(ra, dec, gmst):
    """Unit vectors in ITRS for OF-DATE equatorial (ra, dec) [rad] at Greenwich sidereal angle ``gmst`` (a pure rotation about the
    pole; ICRS directions go through ``icrs_direction_in_itrs``)."""
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
        directions[:, j, :, :] = (icrs_direction_in_itrs(pat[:, 0], pat[:, 1], t) @ R.T)[None, :, :]
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

"""Axis conventions of the Pointing frame, restated from the reference's tests/test_frames.py:100-164
(cases a-d) without astropy.  CPU only."""
import numpy as np

from ionotomo_amd.astro import frames
from ionotomo_amd.synthetic import enu_rotation


def axes_in_local_enu(lon, lat, ha, dec):
    """Pointing axes U, V, W expressed in the ENU frame of the observatory at (lon, lat)."""
    lst = 1.234                                  # any value: only ha = lst - ra matters
    R = frames.pointing_rotation(lon, lst, lst - ha, dec)      # rows u, v, w in ITRS
    return (enu_rotation(lon, lat) @ R.T).T                      # rows U, V, W in ENU


def test_pointing_axis_conventions():
    lon, lat = np.deg2rad(10.0), np.deg2rad(10.0)
    east = np.array([1.0, 0, 0])
    ncp = np.array([0, np.cos(lat), np.sin(lat)])                 # z of the reference's test
    down = np.array([0, np.sin(lat), -np.cos(lat)])                # y of the reference's test
    # a) ha = 0, dec = 90: u = east, v = "down", w = north celestial pole
    U, V, W = axes_in_local_enu(lon, lat, 0.0, np.pi / 2)
    assert np.allclose(U, east) and np.allclose(V, down) and np.allclose(W, ncp)
    # b) v, w and the pole lie on one great circle
    assert abs(np.cross(V, W) @ ncp) < 1e-10
    # c) ha = 0: u points east whatever the declination
    U, V, W = axes_in_local_enu(lon, lat, 0.0, np.deg2rad(35.0))
    assert np.allclose(U, east) and abs(np.cross(V, W) @ ncp) < 1e-10
    # d) dec = 0, ha = -6 h: w points east
    U, V, W = axes_in_local_enu(lon, lat, -np.pi / 2, 0.0)
    assert np.allclose(W, east) and abs(np.cross(V, W) @ ncp) < 1e-10


def test_rotation_is_orthonormal_and_w_points_at_phase_centre():
    rng = np.random.default_rng(0)
    for _ in range(20):
        lon, lst, ra, dec = rng.uniform(-np.pi, np.pi, 3).tolist() + [rng.uniform(-1.4, 1.4)]
        R = frames.pointing_rotation(lon, lst, ra, dec)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-14
        gmst = lst - lon
        d = frames.icrs_to_itrs_direction(ra, dec, gmst)
        assert np.allclose(R @ d, [0, 0, 1], atol=1e-13)            # the phase centre is the frame's w axis


def test_geodesy_gmst_and_bundle_shapes():
    import ionotomo_amd as it
    ra_ = it.RadioArray(array_file=it.RadioArray.lofar_array)
    lon, lat, h = frames.geodetic_from_itrs(ra_.get_center())
    assert abs(np.rad2deg(lon) - 6.84) < 0.1 and abs(np.rad2deg(lat) - 52.91) < 0.1 and -100 < h < 200   # LOFAR core
    # J2000.0 epoch: GMST = 280.46061837 deg
    assert abs(np.rad2deg(frames.gmst_rad((2451545.0 - 2440587.5) * 86400.0)) - 280.46061837) < 1e-6
    # one sidereal day later the angle repeats
    t0 = 1.7e9
    assert abs(((frames.gmst_rad(t0 + 86164.0905) - frames.gmst_rad(t0) + np.pi) % (2 * np.pi)) - np.pi) < 1e-5
    g = frames.gmst_rad(t0)
    phase = (g + lon + 0.01, lat - 0.02)                           # near the zenith
    pat = np.stack([phase[0] + np.array([0.0, 0.01, -0.02]), phase[1] + np.array([0.0, 0.015, 0.01])], -1)
    o, d = frames.model_frame_bundle_from_sky(ra_.get_antenna_locs(), pat, [t0, t0 + 8.0], ra_.get_center(), phase)
    assert o.shape == d.shape == (62, 2, 3, 3)
    assert np.allclose(d[0, 0, 0], [0, 0, 1], atol=1e-12)           # the phase centre itself
    assert np.allclose(np.linalg.norm(d, axis=-1), 1.0)
    assert np.abs(o[:, 0, 0, :]).max() < 80.0                       # stations within ~80 km of the centre
    # 8 s of Earth rotation moves the antennas in the frame by ~ 8 s * 7.29e-5 rad/s * 60 km
    assert 0 < np.abs(o[:, 1] - o[:, 0]).max() < 0.1
    xv, yv, zv = frames.determine_inversion_domain(5.0, o[:, 0, 0, :], d[0, 0], 1000.0, padding=20)
    assert zv[0] < o[:, 0, 0, 2].min() - 90 and zv[-1] > 1000 + 90 and abs((xv[1] - xv[0]) - 5.0) < 0.2

"""Reference-typed (astropy) arguments at the Python boundary, read by attribute (ionotomo_amd/astro/coords.py): stand-ins that
expose exactly what the reference touches (tests/astropy_standins.py) must give the numbers the plain-array call gets.  CPU only;
the GPU side -- calc_rays on such objects == calc_rays on arrays, bit for bit -- is tests/test_gpu_parity.py."""
import numpy as np
import pytest

from astropy_standins import ICRSCoord, ITRSCoord, Quantity, Time
from ionotomo_amd.astro import coords, frames


def test_no_astropy_is_imported():
    import sys
    import ionotomo_amd  # noqa: F401
    assert "astropy" not in sys.modules


def test_itrs_positions_any_unit_scalar_or_array():
    rng = np.random.default_rng(0)
    xyz = 6.4e6 * rng.normal(size=(7, 3))
    assert np.array_equal(coords.itrs_metres(ITRSCoord(xyz, "m")), xyz)
    assert np.allclose(coords.itrs_metres(ITRSCoord(xyz / 1e3, "km")), xyz, rtol=1e-15, atol=0)
    one = coords.itrs_metres(ITRSCoord(xyz[2], "m"))                        # a scalar coordinate: [3]
    assert one.shape == (3,) and np.array_equal(one, xyz[2])
    centre = ITRSCoord(xyz[0], "km")                                       # .earth_location (calc_rays.py:124) wins, in metres
    assert np.allclose(coords.itrs_metres(centre), xyz[0] * 1e3)
    assert np.array_equal(coords.itrs_metres(xyz), xyz) and coords.itrs_metres(None) is None
    assert np.array_equal(coords.itrs_metres(centre.earth_location), xyz[0] * 1e3)      # EarthLocation-like: .x .y .z

    class Bare:                                                            # a Quantity without .to: .value + .unit name
        value, unit = xyz.T / 1e3, "km"

    class C:
        class cartesian:
            xyz = Bare()
    assert np.allclose(coords.itrs_metres(C()), xyz, rtol=1e-15, atol=0)
    Bare.unit = "furlong"
    with pytest.raises(ValueError):
        coords.itrs_metres(C())


def test_icrs_angles_rad_deg_quantity():
    ra, dec = np.array([0.1, 6.2, 3.0]), np.array([-0.3, 0.9, 1.2])
    assert np.array_equal(coords.icrs_radec(ICRSCoord(ra, dec)), np.stack([ra, dec], -1))
    assert coords.icrs_radec(ICRSCoord(ra[1], dec[1])).shape == (2,)

    class DegOnly:
        def __init__(self, rad):
            self.deg = np.rad2deg(rad)

    class P:
        pass
    p = P()
    p.ra, p.dec = DegOnly(ra), DegOnly(dec)
    assert np.allclose(coords.icrs_radec(p), np.stack([ra, dec], -1), rtol=1e-15, atol=1e-16)
    assert np.array_equal(coords.icrs_radec(np.stack([ra, dec], -1)), np.stack([ra, dec], -1))


def test_times_unix_or_gps_with_leap_seconds():
    # known pairs (UTC instant, GPS - UTC): 1980-01-06 -> 0 s; 1999-01-01 -> 13 s; 2015-03-01 -> 16 s; 2017-03-01T12:00:00 -> 18 s
    for unix, leap in ((315964800.0, 0), (915148800.0, 13), (1425168000.0, 16), (1488369600.0, 18), (1.7e9, 18)):
        gps = coords.gps_from_unix(unix)
        assert gps == unix - 315964800.0 + leap
        assert coords.unix_from_gps(gps) == unix
    u = 1488369600.0 + 8.0 * np.arange(5)
    assert np.array_equal(coords.unix_seconds(Time(u)), u)
    assert np.array_equal(coords.unix_seconds(Time(u, only="gps")), u)                  # a Time-like that only offers .gps
    assert float(coords.unix_seconds(Time(u)[3])) == u[3] and np.array_equal(coords.unix_seconds(u), u)
    # across a leap second (2016-12-31T23:59:60): unix repeats, gps does not
    around = 1483228800.0 + np.array([-2.0, -1.0, 0.0, 1.0])
    assert np.array_equal(np.diff(coords.gps_from_unix(around)), [1.0, 2.0, 1.0])
    assert np.array_equal(coords.unix_from_gps(coords.gps_from_unix(around)), around)


def _sky_case(seed=3, na=5, nd=4, nt=3):
    import ionotomo_amd as it
    rng = np.random.default_rng(seed)
    ra_ = it.RadioArray(array_file=it.RadioArray.lofar_array)
    ants = ra_.get_antenna_locs()[:na]
    centre = ra_.get_center()
    lon, lat, _ = frames.geodetic_from_itrs(centre)
    times = 1.49e9 + 8.0 * np.arange(nt)
    phase = np.array([(frames.gmst_rad(times[nt >> 1]) + lon) % (2 * np.pi), lat])
    pat = phase + np.deg2rad(rng.uniform(-2, 2, size=(nd, 2)))
    return ants, pat, times, centre, phase


def test_model_frame_bundle_identical_from_objects_and_arrays():
    """What calc_rays does with the objects before any kernel runs: the same origins / directions as from arrays."""
    ants, pat, times, centre, phase = _sky_case()
    o0, d0 = frames.model_frame_bundle_from_sky(ants, pat, times, centre, phase)
    o1, d1 = frames.model_frame_bundle_from_sky(coords.itrs_metres(ITRSCoord(ants)), coords.icrs_radec(ICRSCoord(pat[:, 0], pat[:, 1])),
                                                coords.unix_seconds(Time(times)), coords.itrs_metres(ITRSCoord(centre)),
                                                coords.icrs_radec(ICRSCoord(phase[0], phase[1])))
    assert np.array_equal(o0, o1) and np.array_equal(d0, d1)


def test_radio_array_and_datapack_take_reference_typed_members():
    import ionotomo_amd as it
    from ionotomo_amd.astro.real_data import DataPack
    ants, pat, times, centre, phase = _sky_case()
    ra0 = it.RadioArray(antenna_pos=ants)
    ra1 = it.RadioArray(antenna_pos=ITRSCoord(ants / 1e3, "km"))
    ra2 = it.RadioArray(earth_locs=ITRSCoord(ants).earth_location)
    for r in (ra1, ra2):
        assert r.Nantenna == ra0.Nantenna and np.allclose(r.get_antenna_locs(), ants, rtol=1e-15, atol=0)
        assert np.allclose(r.get_center(), ra0.get_center(), rtol=1e-15, atol=0)
    labels = np.array(["a%d" % i for i in range(len(ants))])
    names = np.array(["p%d" % i for i in range(len(pat))])
    shape = (len(ants), len(times), len(pat), 2)
    common = dict(radio_array=ra0, antenna_labels=labels, patch_names=names, freqs=[120e6, 150e6], phase=np.zeros(shape),
                  variance=np.ones(shape), clock=np.zeros(shape[:2]), const=np.zeros(shape[0]))
    plain = DataPack(dict(common, antennas=ants, directions=pat, times=times))
    typed = DataPack(dict(common, antennas=ITRSCoord(ants), directions=ICRSCoord(pat[:, 0], pat[:, 1]), times=Time(times)))
    for k in ("antennas", "directions", "times"):
        assert np.array_equal(getattr(plain, k), getattr(typed, k)), k
    assert list(plain.timestamps) == list(typed.timestamps)
    a, lab = typed.get_antennas(ant_idx=[2, 0])
    assert np.array_equal(a, ants[[0, 2]]) and list(lab) == ["a0", "a2"]
    assert np.array_equal(typed.get_times(time_idx=-1)[0], times)
    assert np.allclose(typed.get_center_direction(), plain.get_center_direction())


def test_sky_coordinates_without_a_frame_are_refused():
    from ionotomo_amd.geometry.calc_rays import calc_rays
    ants, pat, times, centre, phase = _sky_case()
    with pytest.raises(ValueError):
        calc_rays(ITRSCoord(ants), ICRSCoord(pat[:, 0], pat[:, 1]), Time(times), None, None, None, None, 120e6, True, 1000.0, 9)


def test_what_the_package_hands_back_answers_the_reference_side_attribute_chains():
    """RadioArray / DataPack getters return float64 arrays that ALSO answer what reference-side code reads off the astropy objects the
    reference returns there (astro/coords.py: ITRSArray, ICRSArray, TimeArray) -- and feed straight back into calc_rays."""
    import ionotomo_amd as it
    from ionotomo_amd.astro.real_data import DataPack, isot_from_unix
    ants, pat, times, centre, phase = _sky_case()
    ra_ = it.RadioArray(antenna_pos=ants)
    locs, cen = ra_.get_antenna_locs(), ra_.get_center()
    assert isinstance(locs, np.ndarray) and locs.dtype == np.float64 and np.array_equal(locs, ants)
    # geometry/calc_rays.py:129: antennas.transform_to(...).cartesian.xyz.to(au.km).value.transpose()
    assert np.allclose(locs.cartesian.xyz.to("km").value.transpose(), ants / 1e3, rtol=1e-15, atol=0)
    assert locs.cartesian.xyz.value.shape == (3, len(ants)) and locs.cartesian.xyz.unit == "m"
    assert np.array_equal(locs[2].cartesian.xyz.value, ants[2]) and np.array_equal(locs[1:3].cartesian.xyz.value, ants[1:3].T)
    # geometry/calc_rays.py:124: array_center.earth_location
    el = cen.earth_location
    assert np.allclose([el.x.to_value("m"), el.y.to_value("m"), el.z.to_value("m")], np.mean(ants, 0), rtol=1e-15, atol=0)
    with pytest.raises(NotImplementedError):
        locs.transform_to("itrs")
    shape = (len(ants), len(times), len(pat), 1)
    dp = DataPack(dict(radio_array=ra_, antennas=ants, antenna_labels=["a%d" % i for i in range(len(ants))], directions=pat,
                       patch_names=["p%d" % i for i in range(len(pat))], times=times, freqs=[150e6], phase=np.zeros(shape),
                       variance=np.ones(shape), clock=np.zeros(shape[:2]), const=np.zeros(shape[0])))
    a, _ = dp.get_antennas(ant_idx=-1)
    d, _ = dp.get_directions(dir_idx=[0, 2])
    t, stamps = dp.get_times(time_idx=-1)
    # astro/real_data.py:55-60: directions.ra.deg / .dec.deg, times.gps
    assert np.allclose(d.ra.deg, np.rad2deg(pat[[0, 2], 0])) and np.array_equal(d.dec.rad, pat[[0, 2], 1])
    assert np.array_equal(t.unix, times) and np.array_equal(t.gps, coords.gps_from_unix(times))
    assert list(t.isot) == [isot_from_unix(x) for x in times] == list(stamps)
    c = dp.get_center_direction()
    assert c.shape == (2,) and float(c.ra.rad) == float(c[0]) and float(c.dec.deg) == np.rad2deg(float(c[1]))
    # numpy sees plain arrays; the readers take them back unchanged
    assert np.array_equal(np.asarray(a), ants) and type(np.asarray(a)) is np.ndarray
    assert np.array_equal(coords.itrs_metres(a), ants) and np.array_equal(coords.icrs_radec(d), pat[[0, 2]])
    assert np.array_equal(coords.unix_seconds(t), times)
    o0, d0 = frames.model_frame_bundle_from_sky(ants, pat, times, centre, phase)
    o1, d1 = frames.model_frame_bundle_from_sky(coords.itrs_metres(a), coords.icrs_radec(dp.get_directions(dir_idx=-1)[0]),
                                                coords.unix_seconds(t), coords.itrs_metres(coords.ITRSArray(centre)), coords.icrs_radec(coords.ICRSArray(phase)))
    assert np.array_equal(o0, o1) and np.array_equal(d0, d1)

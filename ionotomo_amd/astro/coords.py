"""Reading reference-typed (astropy) arguments at the Python boundary WITHOUT importing astropy.

The reference hands ``calc_rays`` / ``RadioArray`` / ``DataPack`` astropy objects (geometry/calc_rays.py:109-145 as called from
inversion/inversion_pipeline.py:195-197 and astro/simulate_observables.py:50-62):

    antennas      ac.SkyCoord(..., frame='itrs')      -> ``.cartesian.xyz`` Quantity [3, Na]         (calc_rays.py:129)
    patches       ac.SkyCoord(..., frame='icrs')      -> ``.ra`` / ``.dec`` Angles (``.rad``, ``.deg``) (real_data.py:55-56)
    times         at.Time array                        -> ``.unix`` / ``.gps`` (real_data.py:60)
    array_center  ac.SkyCoord ITRS, scalar             -> ``.earth_location`` (``.x .y .z``) or ``.cartesian.xyz`` (calc_rays.py:124)
    phase         ac.SkyCoord ICRS, scalar             -> ``.ra.rad`` / ``.dec.rad``
    fixtime       at.Time scalar                       -> ``.unix`` / ``.gps``

Only ATTRIBUTES are read (duck typing): whatever object exposes them is accepted -- a real astropy object where astropy is
installed, a stand-in with the same attributes elsewhere -- and plain arrays pass through unchanged, so every entry point keeps
accepting what it accepted before.  The numbers that come out are the plain arrays the rest of this package works on:
ITRS metres [N,3], (ra, dec) radians [N,2], UTC unix seconds [N].  The frame mathematics that turns them into model-frame rays
is astro/frames.py (host numpy, once per observation time).
"""
import numpy as np

_DEG = np.pi / 180.0
# GPS - UTC [s] from each leap second's introduction (UTC unix seconds): GPS time ran level with UTC on 1980-01-06
_LEAPS_UNIX = np.array([362793600., 394329600., 425865600., 489024000., 567993600., 631152000., 662688000., 709948800., 741484800.,
                        773020800., 820454400., 867715200., 915148800., 1136073600., 1230768000., 1341100800., 1435708800.,
                        1483228800.])
UNIX_MINUS_GPS = 315964800.0       # 1980-01-06T00:00:00 UTC in unix seconds


def unix_from_gps(gps):
    """UTC unix seconds of GPS seconds (continuous since 1980-01-06): minus the leap seconds inserted since."""
    gps = np.asarray(gps, dtype=np.float64)
    u = gps + UNIX_MINUS_GPS
    # (a leap second introduced at unix time L is in force for GPS instants >= L + its own count)
    n = np.searchsorted(_LEAPS_UNIX + np.arange(1, _LEAPS_UNIX.size + 1), u, side="right")
    return u - n


def gps_from_unix(unix):
    unix = np.asarray(unix, dtype=np.float64)
    return unix - UNIX_MINUS_GPS + np.searchsorted(_LEAPS_UNIX, unix, side="right")


_LENGTH_IN_M = {"m": 1.0, "meter": 1.0, "metre": 1.0, "km": 1e3, "kilometer": 1e3, "cm": 1e-2, "mm": 1e-3}
_ANGLE_IN_RAD = {"rad": 1.0, "radian": 1.0, "deg": _DEG, "degree": _DEG, "arcmin": _DEG / 60, "arcsec": _DEG / 3600, "hourangle": 15 * _DEG}


def _value_in(q, unit, table):
    """The numbers of a Quantity-like ``q`` in ``unit`` ('m', 'rad'): ``q.to_value(unit)`` / ``q.to(unit).value`` where the object
    converts itself (astropy takes unit names as strings), else ``q.value`` scaled by its ``.unit`` name, else ``q`` as an array."""
    if hasattr(q, "to_value"):
        return np.asarray(q.to_value(unit), dtype=np.float64)
    if hasattr(q, "to"):
        out = q.to(unit)
        return np.asarray(getattr(out, "value", out), dtype=np.float64)
    if hasattr(q, "value"):
        name = str(getattr(q, "unit", unit)).strip()
        if name not in table:
            raise ValueError("cannot convert a quantity in '%s' to '%s' without astropy" % (name, unit))
        return np.asarray(q.value, dtype=np.float64) * (table[name] / table[unit])
    return np.asarray(q, dtype=np.float64)


def is_coordinate(obj):
    """A sky / Earth coordinate object (anything that is not a plain array or sequence of numbers)?"""
    return any(hasattr(obj, a) for a in ("cartesian", "ra", "earth_location", "transform_to", "x"))  \
        and not isinstance(obj, np.ndarray)


def is_time(obj):
    return (hasattr(obj, "unix") or hasattr(obj, "gps")) and not isinstance(obj, np.ndarray)


def itrs_metres(obj):
    """ITRS positions in metres, [N,3] (or [3] for a scalar coordinate), from

    * an object with ``.earth_location`` (the reference's ``array_center.earth_location``, calc_rays.py:124) or an
      EarthLocation-like with ``.x .y .z``,
    * an object with ``.cartesian.xyz`` (Quantity [3] or [3,N]; the reference: ``.cartesian.xyz.to(au.km).value.transpose()``),
    * a plain array [N,3] / [3] already in metres."""
    if obj is None:
        return None
    if isinstance(obj, np.ndarray):                  # (incl. this package's own ITRSArray: already metres)
        return np.asarray(obj, dtype=np.float64)
    if hasattr(obj, "earth_location"):
        obj = obj.earth_location
    if hasattr(obj, "cartesian"):
        xyz = _value_in(obj.cartesian.xyz, "m", _LENGTH_IN_M)
        return np.ascontiguousarray(np.moveaxis(xyz, 0, -1))
    if all(hasattr(obj, a) for a in ("x", "y", "z")) and not isinstance(obj, np.ndarray):
        return np.stack([_value_in(getattr(obj, a), "m", _LENGTH_IN_M) for a in ("x", "y", "z")], axis=-1)
    return np.asarray(obj, dtype=np.float64)


def _angle_rad(a):
    if hasattr(a, "rad"):
        return np.asarray(a.rad, dtype=np.float64)
    if hasattr(a, "radian"):
        return np.asarray(a.radian, dtype=np.float64)
    if hasattr(a, "deg"):
        return np.asarray(a.deg, dtype=np.float64) * _DEG
    return _value_in(a, "rad", _ANGLE_IN_RAD)


def icrs_radec(obj):
    """(ra, dec) in radians, [N,2] (or [2] for a scalar coordinate), from an object with ``.ra`` / ``.dec`` (Angles: ``.rad``,
    ``.deg``, or a Quantity) or a plain array already in radians."""
    if obj is None:
        return None
    if isinstance(obj, np.ndarray):                  # (incl. this package's own ICRSArray: already radians)
        return np.asarray(obj, dtype=np.float64)
    if hasattr(obj, "ra") and hasattr(obj, "dec"):
        return np.stack([_angle_rad(obj.ra), _angle_rad(obj.dec)], axis=-1)
    return np.asarray(obj, dtype=np.float64)


def unix_seconds(obj):
    """UTC unix seconds (array or scalar) from a Time-like (``.unix``; else ``.gps``) or plain numbers."""
    if obj is None:
        return None
    if not isinstance(obj, np.ndarray):
        if hasattr(obj, "unix"):
            return np.asarray(obj.unix, dtype=np.float64)
        if hasattr(obj, "gps"):
            return unix_from_gps(obj.gps)
    return np.asarray(obj, dtype=np.float64)


# ---- the other direction: what this package hands BACK -----------------------------------------------------------------------------
# The reference's getters return astropy objects (RadioArray.get_antenna_locs / get_center, DataPack.get_antennas / get_directions /
# get_times: astro/radio_array.py:96-124, astro/real_data.py:145-170), and reference-side code reads attributes off them
# (``antennas.cartesian.xyz.to(au.km).value.transpose()``, ``directions.ra.deg``, ``times.gps``: geometry/calc_rays.py:129,
# astro/real_data.py:55-60).  The arrays this package returns are ndarray SUBCLASSES that answer those same attribute chains -- plain
# float64 arrays for everything numpy does (slicing, ufuncs, np.asarray), so nothing inside the package changes -- without astropy.
class QuantityLike(object):
    """Numbers with a unit name; ``to`` / ``to_value`` take a unit object or its name (``str(unit)``: astropy units print as 'km')."""
    __slots__ = ("value", "unit", "_table")

    def __init__(self, value, unit, table):
        self.value, self.unit, self._table = np.asarray(value, dtype=np.float64), str(unit), table

    def to_value(self, unit):
        name = str(unit).strip()
        if name not in self._table:
            raise ValueError("unknown unit '%s'" % name)
        return self.value * (self._table[self.unit] / self._table[name])

    def to(self, unit):
        return QuantityLike(self.to_value(unit), str(unit).strip(), self._table)

    def transpose(self):
        return QuantityLike(self.value.transpose(), self.unit, self._table)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.value, dtype=dtype)


class AngleLike(object):
    __slots__ = ("rad",)

    def __init__(self, rad):
        self.rad = np.asarray(rad, dtype=np.float64)

    radian = property(lambda self: self.rad)
    deg = property(lambda self: self.rad / _DEG)
    degree = deg

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.rad, dtype=dtype)


class _CartesianLike(object):
    __slots__ = ("xyz",)

    def __init__(self, xyz):
        self.xyz = xyz


class _EarthLocationLike(object):
    __slots__ = ("x", "y", "z")

    def __init__(self, xyz_m):
        m = np.moveaxis(np.asarray(xyz_m, dtype=np.float64), -1, 0)
        self.x, self.y, self.z = (QuantityLike(m[a], "m", _LENGTH_IN_M) for a in range(3))


def _no_transform(self, *args, **kwargs):
    raise NotImplementedError("frame transformations are astropy's; this package rotates ITRS / ICRS numbers into the model frame "
                              "itself (ionotomo_amd.astro.frames, calc_rays)")


class ITRSArray(np.ndarray):
    """ITRS positions in metres, [N,3] or [3]: a float64 array that also answers ``.cartesian.xyz`` ([3,N] Quantity-like, as
    astropy orders it) and ``.earth_location`` (``.x .y .z``)."""
    def __new__(cls, xyz_m):
        return np.ascontiguousarray(xyz_m, dtype=np.float64).view(cls)

    cartesian = property(lambda self: _CartesianLike(QuantityLike(np.moveaxis(np.asarray(self), -1, 0), "m", _LENGTH_IN_M)))
    earth_location = property(lambda self: _EarthLocationLike(np.asarray(self)))
    transform_to = _no_transform


class ICRSArray(np.ndarray):
    """(ra, dec) in radians, [N,2] or [2]: a float64 array that also answers ``.ra`` / ``.dec`` (``.rad``, ``.deg``)."""
    def __new__(cls, radec_rad):
        return np.ascontiguousarray(radec_rad, dtype=np.float64).view(cls)

    ra = property(lambda self: AngleLike(np.asarray(self)[..., 0]))
    dec = property(lambda self: AngleLike(np.asarray(self)[..., 1]))
    transform_to = _no_transform


class TimeArray(np.ndarray):
    """UTC unix seconds: a float64 array that also answers ``.unix``, ``.gps`` and ``.isot``."""
    def __new__(cls, unix):
        return np.ascontiguousarray(unix, dtype=np.float64).view(cls)

    unix = property(lambda self: np.asarray(self))
    gps = property(lambda self: gps_from_unix(np.asarray(self)))

    @property
    def isot(self):
        import time as _t

        def one(t):
            t = float(t)
            return _t.strftime("%Y-%m-%dT%H:%M:%S", _t.gmtime(int(np.floor(t)))) + ".{:03d}".format(int(round((t - np.floor(t)) * 1000)) % 1000)
        a = np.asarray(self)
        return one(a) if a.ndim == 0 else np.array([one(t) for t in a.ravel()]).reshape(a.shape)

"""Stand-ins with exactly the attributes of the astropy objects the reference passes across the path's Python boundary
(geometry/calc_rays.py:109-145 as called from inversion/inversion_pipeline.py:195-197): no astropy in this image.  Each class
exposes ONLY what the reference itself touches on such an object -- ``.cartesian.xyz`` (a Quantity: ``.to(unit).value``),
``.ra`` / ``.dec`` (Angles: ``.rad``, ``.deg``), ``.unix`` / ``.gps`` / ``.isot``, ``.earth_location``, ``len()``, indexing,
``.transform_to`` (present, never callable here: calling it would need the frame machinery of astropy)."""
import numpy as np

_M = {"m": 1.0, "km": 1e3}


class Quantity:
    """Numbers with a length unit; like astropy's, ``to`` takes a unit object or its name."""
    def __init__(self, value, unit):
        self.value, self.unit = np.asarray(value, dtype=np.float64), str(unit)

    def to(self, unit):
        return Quantity(self.value * (_M[self.unit] / _M[str(unit)]), str(unit))

    def transpose(self):
        return Quantity(self.value.transpose(), self.unit)


class Angle:
    def __init__(self, rad):
        self.rad = np.asarray(rad, dtype=np.float64)

    @property
    def deg(self):
        return np.rad2deg(self.rad)


class _Cartesian:
    def __init__(self, xyz):
        self.xyz = xyz                                    # Quantity [3] or [3, N]


def _no_transform(*a, **k):
    raise AssertionError("transform_to needs astropy's frame graph: the build must read attributes instead")


class EarthLocation:
    def __init__(self, xyz_m):
        x, y, z = np.moveaxis(np.asarray(xyz_m, dtype=np.float64), -1, 0)
        self.x, self.y, self.z = Quantity(x, "m"), Quantity(y, "m"), Quantity(z, "m")


class ITRSCoord:
    """ac.SkyCoord(x, y, z, frame='itrs'): positions [N,3] given in ``unit`` (stored that way: the reader must convert)."""
    transform_to = staticmethod(_no_transform)

    def __init__(self, xyz, unit="m"):
        self._xyz, self._unit = np.asarray(xyz, dtype=np.float64), unit

    @property
    def cartesian(self):
        return _Cartesian(Quantity(np.moveaxis(self._xyz, -1, 0), self._unit))

    @property
    def earth_location(self):
        return EarthLocation(self._xyz * _M[self._unit])

    def __len__(self):
        return len(self._xyz)

    def __getitem__(self, i):
        return ITRSCoord(self._xyz[i], self._unit)


class ICRSCoord:
    """ac.SkyCoord(ra, dec, frame='icrs')."""
    transform_to = staticmethod(_no_transform)

    def __init__(self, ra_rad, dec_rad):
        self.ra, self.dec = Angle(ra_rad), Angle(dec_rad)

    def __len__(self):
        return len(self.ra.rad)

    def __getitem__(self, i):
        return ICRSCoord(self.ra.rad[i], self.dec.rad[i])


class Time:
    """at.Time: ``.unix`` (UTC), ``.gps``, ``.isot``; ``only`` removes all but one of them (a Time-like that offers just that one)."""
    def __init__(self, unix, only=None):
        from ionotomo_amd.astro.coords import gps_from_unix
        from ionotomo_amd.astro.real_data import isot_from_unix
        u = np.asarray(unix, dtype=np.float64)
        if only in (None, "unix"):
            self.unix = u
        if only in (None, "gps"):
            self.gps = gps_from_unix(u)
        if only is None:
            self.isot = np.array([isot_from_unix(t) for t in np.atleast_1d(u)]) if u.ndim else isot_from_unix(u)
        self._u, self._only = u, only

    def __len__(self):
        return len(self._u)

    def __getitem__(self, i):
        return Time(self._u[i], self._only)

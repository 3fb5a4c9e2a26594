"""``DataPack`` -- the array semantics of ionotomo.astro.real_data.DataPack (astro/real_data.py:16-498)
that sit either side of the ray-integral path, without astropy.

What the hot path consumes from a datapack is arrays: antenna positions, facet directions, times,
frequencies, and the ``phase / variance [Na,Nt,Nd,Nf]``, ``clock [Na,Nt]``, ``const [Na]`` slots with
index-set access, reference-antenna differencing and flagging.  Here the coordinate members are plain
arrays where the reference holds astropy objects:

    antennas   [Na,3]  ITRS metres           (reference: ac.ITRS SkyCoord)
    directions [Nd,2]  (ra, dec) radians     (reference: ac.ICRS SkyCoord)
    times      [Nt]    UTC unix seconds      (reference: at.Time); ``timestamps`` are the ISOT labels

Index conventions are the reference's (:145-260): ``-1`` selects a whole axis, index lists are used in
sorted order, and a getter returns the outer-product block of the selected indices.

Storage: ``save``/``load`` use ``.npz`` natively; a filename ending in ``.hdf5``/``.h5`` uses the
reference's ``datapack/*`` HDF5 layout (:43-117) through the pure-numpy reader / writer of utils/hdf5_lite.py, pinned by a file
the reference's own ``DataPack.save`` wrote with real h5py (tests/golden/datapack_reference_h5py.hdf5).  Otherwise
PARITY UNPINNED: the reference class cannot be instantiated without
astropy; the behaviour its own test pins (tests/test_astro.py:15-35) is restated in
tests/test_datapack.py.
"""
import time as _time

import numpy as np

from . import coords
from .radio_array import RadioArray, generate_example_radio_array
from .frames import geodetic_from_itrs, gmst_rad, itrs_direction_to_icrs, pointing_rotation

# slot name -> which of the (antenna, time, direction, frequency) axes it carries
_SLOT_AXES = {"phase": (0, 1, 2, 3), "variance": (0, 1, 2, 3), "prop": (0, 1, 2, 3), "clock": (0, 1), "const": (0,)}
_UNIX_MINUS_GPS = 315964800.0      # 1980-01-06T00:00:00 UTC in unix seconds (leap seconds ignored)


def isot_from_unix(t):
    t = float(t)
    return _time.strftime("%Y-%m-%dT%H:%M:%S", _time.gmtime(int(np.floor(t)))) + ".{:03d}".format(
        int(round((t - np.floor(t)) * 1000)) % 1000)


class DataPack(object):
    def __init__(self, data_dict=None, filename=None, ref_ant=None):
        self.ref_ant = None
        if data_dict is not None:
            self.add_data_dict(**data_dict)
        elif filename is not None:
            self.load(filename)
            return
        if ref_ant is not None:
            self.set_reference_antenna(ref_ant)

    def __repr__(self):
        return ("DataPack: num_antennas = {}, num_time = {}, num_directions = {}, num_freqs = {}\n"
                "Reference Antenna = {}").format(self.Na, self.Nt, self.Nd, self.Nf, self.ref_ant)

    # -- construction ------------------------------------------------------------------------------
    def add_data_dict(self, **args):
        for key in ("radio_array", "antennas", "antenna_labels", "times", "timestamps", "directions", "patch_names",
                    "freqs", "phase", "const", "clock", "prop", "variance"):
            setattr(self, key, args.get(key, None))
        # reference-typed members (ITRS / ICRS coordinates, Time: astro/real_data.py:124-131) are read by attribute into the plain
        # arrays this class holds (astro/coords.py; astropy is not imported)
        if self.antennas is not None and coords.is_coordinate(self.antennas):
            self.antennas = coords.itrs_metres(self.antennas).reshape(-1, 3)
        if self.directions is not None and coords.is_coordinate(self.directions):
            self.directions = coords.icrs_radec(self.directions).reshape(-1, 2)
        if self.times is not None and coords.is_time(self.times):
            if self.timestamps is None and hasattr(self.times, "isot"):
                self.timestamps = np.atleast_1d(np.asarray(self.times.isot)).astype(str)
            self.times = np.atleast_1d(coords.unix_seconds(self.times))
        # (plain float64 arrays that ALSO answer the attribute chains reference-side code reads off the astropy objects the reference
        #  holds here -- .cartesian.xyz, .ra / .dec, .unix / .gps / .isot: astro/coords.py)
        self.antennas = coords.ITRSArray(np.asarray(self.antennas, dtype=np.float64).reshape(-1, 3))
        self.times = coords.TimeArray(np.atleast_1d(np.asarray(self.times, dtype=np.float64)))
        self.directions = coords.ICRSArray(np.asarray(self.directions, dtype=np.float64).reshape(-1, 2))
        self.freqs = np.atleast_1d(np.asarray(self.freqs, dtype=np.float64))
        self.Na, self.Nt, self.Nd, self.Nf = len(self.antennas), len(self.times), len(self.directions), len(self.freqs)
        if self.timestamps is None:
            self.timestamps = [isot_from_unix(t) for t in self.times]
        self.antenna_labels = np.array(self.antenna_labels)
        self.patch_names = np.array(self.patch_names)
        self.timestamps = np.array(self.timestamps)
        full = (self.Na, self.Nt, self.Nd, self.Nf)
        for name, axes in _SLOT_AXES.items():
            arr = getattr(self, name)
            if arr is not None:
                want = tuple(full[a] for a in axes)
                assert np.shape(arr) == want, "Invalid shape {} {} for {}".format(np.shape(arr), want, name)
                setattr(self, name, np.array(arr, dtype=np.float64))
        self.ref_ant = None
        if args.get("ref_ant", None) is not None:
            self.set_reference_antenna(args["ref_ant"])

    def get_data_dict(self):
        return {"radio_array": self.radio_array, "antennas": self.antennas, "antenna_labels": self.antenna_labels,
                "times": self.times, "timestamps": self.timestamps, "directions": self.directions,
                "patch_names": self.patch_names, "freqs": self.freqs, "phase": self.phase, "ref_ant": self.ref_ant,
                "const": self.const, "clock": self.clock, "variance": self.variance}

    def clone(self):
        return DataPack(self.get_data_dict())

    # -- index-set access (astro/real_data.py:145-260) -----------------------------------------------
    def _axis_lengths(self):
        return (self.Na, self.Nt, self.Nd, self.Nf)

    def _block(self, param, indices):
        """np.ix_ selector of slot ``param`` for the 4-tuple ``indices`` (None = axis not addressed)."""
        assert isinstance(indices, (tuple, list)) and len(indices) == 4
        if param not in _SLOT_AXES or getattr(self, param, None) is None:
            raise ValueError("param does not exist {}".format(param))
        sel = []
        for axis, idx in enumerate(indices):
            if idx is None:
                continue
            if isinstance(idx, (int, np.integer)) and idx == -1:
                idx = np.arange(self._axis_lengths()[axis])
            sel.append(np.sort(np.asarray(idx, dtype=np.int64).reshape(-1)))
        assert len(sel) == len(_SLOT_AXES[param]), "{} takes {} index sets".format(param, len(_SLOT_AXES[param]))
        return np.ix_(*sel)

    def get_slot(self, param, indices):
        blk = self._block(param, indices)
        return getattr(self, param)[blk]

    def set_slot(self, param, A, indices, set_ref_ant=False):
        blk = self._block(param, indices)
        getattr(self, param)[blk] = A
        if set_ref_ant and self.ref_ant is not None:
            self.set_reference_antenna(self.ref_ant)

    def get_phase(self, ant_idx=[], time_idx=[], dir_idx=[], freq_idx=[]):
        return self.get_slot("phase", (ant_idx, time_idx, dir_idx, freq_idx))

    def set_phase(self, phase, ant_idx=[], time_idx=[], dir_idx=[], freq_idx=[], ref_ant=None):
        self.set_slot("phase", phase, (ant_idx, time_idx, dir_idx, freq_idx))
        self.set_reference_antenna(ref_ant)

    def get_variance(self, ant_idx=[], time_idx=[], dir_idx=[], freq_idx=[]):
        return self.get_slot("variance", (ant_idx, time_idx, dir_idx, freq_idx))

    def set_variance(self, variance, ant_idx=[], time_idx=[], dir_idx=[], freq_idx=[]):
        self.set_slot("variance", variance, (ant_idx, time_idx, dir_idx, freq_idx))

    def get_prop(self, ant_idx=[], time_idx=[], dir_idx=[], freq_idx=[]):
        return self.get_slot("prop", (ant_idx, time_idx, dir_idx, freq_idx))

    def set_prop(self, prop, ant_idx=[], time_idx=[], dir_idx=[], freq_idx=[], ref_ant=None):
        self.set_slot("prop", prop, (ant_idx, time_idx, dir_idx, freq_idx))
        self.set_reference_antenna(ref_ant)

    def get_clock(self, ant_idx=[], time_idx=[]):
        return self.get_slot("clock", (ant_idx, time_idx, None, None))

    def set_clock(self, clock, ant_idx=[], time_idx=[], ref_ant=None):
        self.set_slot("clock", clock, (ant_idx, time_idx, None, None))
        self.set_reference_antenna(ref_ant)

    def get_const(self, ant_idx=[]):
        return self.get_slot("const", (ant_idx, None, None, None))

    def set_const(self, const, ant_idx=[], ref_ant=None):
        self.set_slot("const", const, (ant_idx, None, None, None))
        self.set_reference_antenna(ref_ant)

    def _pick(self, idx, n):
        if isinstance(idx, (int, np.integer)) and idx == -1:
            idx = np.arange(n)
        return np.sort(np.asarray(idx, dtype=np.int64).reshape(-1))

    def get_antennas(self, ant_idx=[]):
        i = self._pick(ant_idx, self.Na)
        return self.antennas[i], self.antenna_labels[i]

    def get_times(self, time_idx=[]):
        i = self._pick(time_idx, self.Nt)
        return self.times[i], self.timestamps[i]

    def get_directions(self, dir_idx=[]):
        i = self._pick(dir_idx, self.Nd)
        return self.directions[i], self.patch_names[i]

    def get_freqs(self, freq_idx=[]):
        return self.freqs[self._pick(freq_idx, self.Nf)]

    def get_antenna_idx(self, ant):
        assert ant in self.antenna_labels, "{} not a valid label".format(ant)
        return int(np.where(self.antenna_labels == ant)[0][0])

    def get_center_direction(self):
        """Mean (ra, dec) [rad] of the facets (astro/real_data.py:380-385).  The right ascension is averaged on
        the circle, so a field straddling ra = 0 gets its true centre (the reference's plain ``np.mean`` of the
        wrapped angles does not)."""
        ra = self.directions[:, 0]
        ra = np.asarray(ra)
        return coords.ICRSArray([np.arctan2(np.mean(np.sin(ra)), np.mean(np.cos(ra))) % (2 * np.pi),
                                 float(np.mean(np.asarray(self.directions)[:, 1]))])

    # -- reference antenna (astro/real_data.py:366-378) -----------------------------------------------
    def set_reference_antenna(self, ref_ant):
        if ref_ant is None:
            return
        if ref_ant not in self.antenna_labels:
            raise ValueError("{} is not a valid antenna. Choose from {}".format(ref_ant, self.antenna_labels))
        i = self.get_antenna_idx(ref_ant)
        self.ref_ant = ref_ant
        self.phase -= self.phase[i, :, :, :].copy()
        if self.clock is not None:
            self.clock -= self.clock[i, :].copy()
        if self.const is not None:
            self.const -= self.const[i]

    # -- flagging (astro/real_data.py:387-482) ----------------------------------------------------------
    def find_flagged_antennas(self):
        """Labels of antennas whose phases are all zero (the reference antenna excepted)."""
        assert self.ref_ant is not None, "Set a ref_ant before finding flagged (zeroed) antennas"
        dead = np.sum(self.phase, axis=(1, 2, 3)) == 0
        return [str(lab) for lab, m in zip(self.antenna_labels, dead) if m and lab != self.ref_ant]

    def _drop(self, axis, keep):
        for name, axes in _SLOT_AXES.items():
            arr = getattr(self, name, None)
            if arr is not None and axis in axes:
                setattr(self, name, np.compress(keep, arr, axis=axes.index(axis)))

    @staticmethod
    def _as_list(x):
        return [x] if isinstance(x, str) or not hasattr(x, "__iter__") else list(x)

    def flag_antennas(self, antenna_labels):
        gone = set(self._as_list(antenna_labels))
        keep = np.array([lab not in gone for lab in self.antenna_labels], dtype=bool)
        assert keep.any(), "Must leave at least one antenna"
        if self.ref_ant in gone:
            self.ref_ant = None
        self.antenna_labels, self.antennas = self.antenna_labels[keep], self.antennas[keep]
        self._drop(0, keep)
        self.Na = len(self.antennas)

    def flag_times(self, timestamps):
        gone = set(self._as_list(timestamps))
        keep = np.array([t not in gone for t in self.timestamps], dtype=bool)
        assert keep.any(), "Must leave at least one time"
        self.timestamps, self.times = self.timestamps[keep], self.times[keep]
        self._drop(1, keep)
        self.Nt = len(self.times)

    def flag_directions(self, patch_names):
        gone = set(self._as_list(patch_names))
        keep = np.array([p not in gone for p in self.patch_names], dtype=bool)
        assert keep.any(), "Must leave at least one direction"
        self.patch_names, self.directions = self.patch_names[keep], self.directions[keep]
        self._drop(2, keep)
        self.Nd = len(self.directions)

    def flag_freqs(self, freq_idx=[]):
        keep = np.ones(self.Nf, dtype=bool)
        keep[np.asarray(list(freq_idx), dtype=np.int64)] = False
        assert keep.any(), "Must leave at least one frequency"
        self.freqs = self.freqs[keep]
        self._drop(3, keep)
        self.Nf = len(self.freqs)

    # -- storage -----------------------------------------------------------------------------------------
    def save(self, filename):
        if str(filename).endswith((".hdf5", ".h5")):
            return self._save_hdf5(filename)
        blob = dict(antenna_labels=self.antenna_labels.astype(str), antennas=self.antennas,
                    frequency=float(self.radio_array.frequency) if self.radio_array is not None else np.nan,
                    patch_names=self.patch_names.astype(str), directions=self.directions,
                    timestamps=self.timestamps.astype(str), times=self.times, freqs=self.freqs,
                    ref_ant=np.array("" if self.ref_ant is None else str(self.ref_ant)))
        for name in _SLOT_AXES:
            if getattr(self, name, None) is not None:
                blob[name] = getattr(self, name)
        with open(filename, "wb") as f:
            np.savez(f, **blob)

    def load(self, filename):
        if str(filename).endswith((".hdf5", ".h5")):
            return self._load_hdf5(filename)
        with np.load(filename, allow_pickle=False) as z:
            d = {k: z[k] for k in z.files}
        freq = float(d.pop("frequency"))
        ref = str(d.pop("ref_ant"))
        d["radio_array"] = RadioArray(antenna_pos=d["antennas"], frequency=120e6 if np.isnan(freq) else freq)
        self.add_data_dict(**d)
        # stored phases are already referenced; only restore the label
        self.ref_ant = ref or None

    def _save_hdf5(self, filename):
        """The reference's ``datapack/*`` layout (astro/real_data.py:43-79): same dataset names, dtypes (float64, variable-length
        UTF-8 strings) and attributes, written by utils/hdf5_lite.py (h5py / h5dump read it: tests/test_hdf5_lite.py)."""
        from ..utils import hdf5_lite
        dp = {"antennas": {"labels": np.array(self.antenna_labels, dtype=object), "locs": np.asarray(self.antennas, dtype=np.float64),
                           "@attrs": {"frequency": float(self.radio_array.frequency)}},
              "directions": {"patchnames": np.array(self.patch_names, dtype=object), "ra": np.rad2deg(self.directions[:, 0]),
                             "dec": np.rad2deg(self.directions[:, 1])},
              "times": {"timestamps": np.array(self.timestamps, dtype=object), "gps": np.asarray(self.times) - _UNIX_MINUS_GPS},
              "freqs": np.asarray(self.freqs, dtype=np.float64)}
        for name in ("phase", "variance", "clock", "const"):
            if getattr(self, name, None) is not None:
                dp[name] = np.asarray(getattr(self, name), dtype=np.float64)
        if "phase" in dp:
            dp["phase@attrs"] = {"ref_ant": str(self.ref_ant)}
        hdf5_lite.write(filename, {"datapack": dp})

    def _load_hdf5(self, filename):
        """A datapack written by the reference (h5py) or by ``_save_hdf5`` (astro/real_data.py:81-117)."""
        from ..utils import hdf5_lite
        f = hdf5_lite.read(filename)["datapack"]
        d = dict(antenna_labels=np.asarray(f["antennas"]["labels"]).astype(str), antennas=f["antennas"]["locs"],
                 patch_names=np.asarray(f["directions"]["patchnames"]).astype(str),
                 directions=np.deg2rad(np.stack([f["directions"]["ra"], f["directions"]["dec"]], -1)),
                 timestamps=np.asarray(f["times"]["timestamps"]).astype(str), times=f["times"]["gps"] + _UNIX_MINUS_GPS,
                 freqs=f["freqs"])
        na, nt, nd, nf = len(d["antennas"]), len(d["times"]), len(d["directions"]), len(d["freqs"])
        shapes = {"phase": (na, nt, nd, nf), "variance": (na, nt, nd, nf), "clock": (na, nt), "const": (na,)}
        for name, shp in shapes.items():
            d[name] = f[name] if name in f else np.zeros(shp)
        ref = f.get("phase@attrs", {}).get("ref_ant", None)
        d["radio_array"] = RadioArray(antenna_pos=d["antennas"], frequency=f["antennas"]["@attrs"]["frequency"])
        self.add_data_dict(**d)
        if ref is not None and str(ref) != "None":
            self.set_reference_antenna(str(ref))


def sky_from_pointing_dirs(dirs_uvw, centre_itrs_m, phase_radec, unix_time):
    """(ra, dec) [N,2] of unit vectors given in the Pointing frame about ``phase_radec`` at ``unix_time``
    (the inverse of astro/frames.py:model_frame_bundle_from_sky for directions)."""
    lon, _, _ = geodetic_from_itrs(centre_itrs_m)
    g = gmst_rad(unix_time)
    R = pointing_rotation(lon, g + lon, phase_radec[0], phase_radec[1])
    v = np.asarray(dirs_uvw, dtype=np.float64) @ R                 # rows of R are u, v, w in ITRS
    return itrs_direction_to_icrs(v, unix_time)


def generate_example_datapack(Nant=10, Ntime=1, Ndir=10, Nfreqs=4, fov=4., alt=90., az=0., time=None, radio_array=None,
                              seed=None):
    """A datapack for testing (astro/real_data.py:514-558): ``Ndir`` facets scattered in a ``fov``-degree
    cone about the pointing (alt, az) at the array centre, 8-s cadence, phases = const + 2 pi nu clock
    - 8.448e-7/nu TEC + 5-degree noise, referenced to the first antenna.  ``time``: unix seconds, an
    ISOT string or None (now)."""
    rng = np.random.default_rng(seed)
    if radio_array is None:
        radio_array = generate_example_radio_array(Nant=Nant, seed=seed)
    if time is None:
        t0 = float(int(_time.time()))
    elif isinstance(time, str):
        import calendar
        t0 = float(calendar.timegm(_time.strptime(time.split(".")[0], "%Y-%m-%dT%H:%M:%S")))
    else:
        t0 = float(time)
    antennas, labels = radio_array.get_antenna_locs(), radio_array.get_antenna_labels()
    Nant = len(antennas)
    times = t0 + 8.0 * np.arange(Ntime)
    centre = radio_array.get_center()
    lon, lat, _ = geodetic_from_itrs(centre)
    # pointing (alt, az) at the array centre -> (ra, dec) at t0 (az from north through east)
    a, z = np.deg2rad(alt), np.deg2rad(az)
    enu = np.array([np.cos(a) * np.sin(z), np.cos(a) * np.cos(z), np.sin(a)])
    from ..synthetic import enu_rotation
    v = enu @ enu_rotation(lon, lat)
    phase_centre = itrs_direction_to_icrs(v, t0)[0]
    phi = np.deg2rad(rng.uniform(-fov / 2., fov / 2., Ndir))
    theta = np.deg2rad(rng.uniform(0., 360., Ndir))
    uvw = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], -1)
    directions = sky_from_pointing_dirs(uvw, centre, phase_centre, t0)
    freqs = np.linspace(-0.5, 0.5, Nfreqs) * Nfreqs * 2e6 + radio_array.frequency
    tec = rng.normal(size=(Nant, Ntime, Ndir)) * 0.01 * 1e16
    clock = rng.normal(size=(Nant, Ntime)) * 5e-9
    const = rng.normal(size=Nant) * 2 * np.pi
    phase = (const[:, None, None, None] + 2 * np.pi * freqs[None, None, None, :] * clock[:, :, None, None]
             - 8.4480e-7 / freqs[None, None, None, :] * tec[..., None])
    phase = phase + rng.normal(size=phase.shape) * np.deg2rad(5.)
    dp = DataPack(data_dict=dict(radio_array=radio_array, antennas=antennas, antenna_labels=labels, times=times,
                                 timestamps=[isot_from_unix(t) for t in times], directions=directions,
                                 patch_names=np.array(["facet_patch_{}".format(i) for i in range(Ndir)]), freqs=freqs,
                                 phase=phase, clock=clock, const=const, variance=np.zeros_like(phase)))
    dp.set_reference_antenna(labels[0])
    return dp


def phase_screen_datapack(N, ant_idx=-1, time_idx=-1, dir_idx=-1, freq_idx=-1, Nant=10, Ntime=1, Nfreqs=1, fov=4., alt=90.,
                          az=0., time=None, radio_array=None, datapack=None):
    """Empty datapack whose N^2 directions tile the (ra, dec) bounding box of ``datapack``'s facets
    (astro/real_data.py:560-608)."""
    if datapack is None:
        datapack = generate_example_datapack(Nant=Nant, Ntime=Ntime, Ndir=1, Nfreqs=Nfreqs, fov=fov, alt=alt, az=az,
                                             time=time, radio_array=radio_array)
    antennas, labels = datapack.get_antennas(ant_idx=ant_idx)
    times, timestamps = datapack.get_times(time_idx=time_idx)
    freqs = datapack.get_freqs(freq_idx=freq_idx)
    d = datapack.directions
    ra, dec = np.meshgrid(np.linspace(d[:, 0].min(), d[:, 0].max(), N), np.linspace(d[:, 1].min(), d[:, 1].max(), N),
                          indexing="ij")
    dirs = np.stack([ra.ravel(), dec.ravel()], -1)
    na, nt, nf = len(antennas), len(times), len(freqs)
    dd = datapack.get_data_dict()
    dd.update(antennas=antennas, antenna_labels=labels, times=times, timestamps=timestamps, directions=dirs,
              patch_names=np.array(["facet_patch_{}".format(i) for i in range(len(dirs))]),
              phase=np.zeros((na, nt, len(dirs), nf)), clock=np.zeros((na, nt)), const=np.zeros(na), freqs=freqs,
              variance=np.zeros((na, nt, len(dirs), nf)), ref_ant=None)
    out = DataPack(data_dict=dd)
    out.set_reference_antenna(labels[0])
    return out

"""``RadioArray`` -- the API surface of ionotomo.astro.radio_array.RadioArray (astro/radio_array.py:11-139)
that the ray-integral path consumes, without astropy: positions are plain ITRS metres.

``get_antenna_locs()`` returns an ndarray [N,3] (ITRS m) where the reference returns an astropy
SkyCoord -- an ndarray SUBCLASS that also answers the attribute chains reference-side code reads off that
SkyCoord (``.cartesian.xyz.to(unit).value``, ``.earth_location``: astro/coords.py); ``get_center()`` the centroid [3], likewise; ``enu_km()`` gives the model-frame antenna positions
(local East/North/Up km about the centroid) that ``calc_rays`` takes.
"""
import os

import numpy as np

from .coords import ITRSArray, itrs_metres
from ..synthetic import itrs_to_enu_km, read_array_table, read_station_enu_csv

_ARRAYS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "arrays")


class RadioArray(object):
    lofar_array = os.path.join(_ARRAYS, 'lofar_hba_stations.csv')
    lofar_cycle0_array = os.path.join(_ARRAYS, 'lofar_cycle0_hba_stations.csv')
    gmrt_array = os.path.join(_ARRAYS, 'gmrt_stations.csv')

    def __init__(self, array_file=None, antenna_pos=None, name=None, msFile=None, num_antennas=0, earth_locs=None,
                 frequency=120e6, **kwargs):
        self.frequency = frequency
        self.Nantenna = 0
        if array_file is not None:
            self.array_file = array_file
            self.load_array_file(array_file)
        if antenna_pos is not None:
            self.load_pos_array(antenna_pos)
        elif earth_locs is not None:      # (astro/radio_array.py:22-23: EarthLocation-like, ``.x .y .z``)
            self.load_pos_array(earth_locs)

    def load_array_file(self, array_file):
        """Whitespace table ``X Y Z diameter label`` in ITRS metres, ``#`` comments
        (astro/radio_array.py:28-45)."""
        if str(array_file).endswith(".csv"):
            xyz, diam, labels = read_station_enu_csv(array_file)     # ENU-offset station table shipped here
        else:
            xyz, diam, labels = read_array_table(array_file)         # the reference's X Y Z diam label format
        self.locs = ITRSArray(xyz)
        if diam is not None and np.all(np.asarray(diam) < 0):
            diam = None                                # table without dish diameters (GMRT)
        self.diameters = diam
        self.labels = labels
        self.Nantenna = int(xyz.shape[0])
        self.calc_center()

    def load_pos_array(self, antenna_pos, antenna_labels=None):
        """``antenna_pos``: ITRS metres [N,3], or an ITRS coordinate object as the reference holds them (read by attribute,
        ``.cartesian.xyz``: astro/coords.py)."""
        self.locs = ITRSArray(np.ascontiguousarray(itrs_metres(antenna_pos), dtype=np.float64).reshape(-1, 3))
        self.Nantenna = self.locs.shape[0]
        if antenna_labels is not None:
            assert len(antenna_labels) == self.Nantenna
            self.labels = np.array([str(lab) for lab in antenna_labels])
        else:
            self.labels = np.array(["ant{:02d}".format(i) for i in range(self.Nantenna)])
        self.diameters = None
        self.calc_center()

    def get_antenna_locs(self):
        return self.locs

    def get_antenna_labels(self):
        return np.array(self.labels)

    def get_fov(self):
        return 4. * np.pi / 180.

    def calc_center(self):
        self.center = np.mean(self.locs, axis=0)
        return self.center

    def get_center(self):
        return self.center

    def get_antenna_idx(self, name):
        for i in range(self.Nantenna):
            if self.labels[i] == name:
                return i
        return None

    def enu_km(self):
        return itrs_to_enu_km(self.locs)

    def save_array_file(self, array_file):
        with open(array_file, 'w') as f:
            f.write('# ITRS(m)\n# X\tY\tZ\tdiameter\tlabels\n')
            for i in range(self.Nantenna):
                d = self.diameters[i] if self.diameters is not None else -1
                f.write('{0:1.9e}\t{1:1.9e}\t{2:1.9e}\t{3:1.4e}\t{4}\n'.format(self.locs[i, 0], self.locs[i, 1],
                                                                            self.locs[i, 2], d, self.labels[i]))

    def __repr__(self):
        return "Radio Array: {0:1.5e} MHz, {1} antennas".format(self.frequency, self.Nantenna)


def generate_example_radio_array(Nant=10, config=None, seed=None, **kwargs):
    """astro/radio_array.py:124-139: 'lofar' or Nant stations scattered N(0, (40 km)^2) about a point."""
    if config is not None:
        if config == 'lofar':
            return RadioArray(array_file=RadioArray.lofar_array, **kwargs)
        return None
    rng = np.random.default_rng(seed)
    lon, lat = rng.uniform(0, 2 * np.pi), rng.uniform(-np.pi / 3., np.pi / 3.)
    enu = np.zeros((Nant, 3))
    enu[:, :2] = rng.normal(scale=40e3, size=(Nant, 2))
    sl, cl, so, co = np.sin(lat), np.cos(lat), np.sin(lon), np.cos(lon)
    R = np.array([[-so, co, 0.0], [-sl * co, -sl * so, cl], [cl * co, cl * so, sl]])
    p0 = 6371e3 * np.array([cl * co, cl * so, sl])
    return RadioArray(antenna_pos=p0 + enu @ R, **kwargs)

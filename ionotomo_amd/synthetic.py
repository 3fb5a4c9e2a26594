"""Seeded synthetic inputs for the ray-integral path.

Nothing here is on the hot path: these are host-side numpy generators for the
benchmark / test workloads named in BASELINE.json ``configs``.  The formulas
restate the reference's own input generators so the workloads look like what
the reference feeds its forward model:

* stratified four-layer Chapman profile  -> ``ionosphere/iri.py:20-68``
  (``a_priori_model_``)
* Matern-5/2 Gaussian random field (spectral synthesis)
                                          -> ``ionosphere/simulation.py:51-112``
  (``IonosphereSimulation``), as used by ``create_turbulent_model``
  (``inversion/initial_model.py:75-84``: ``ne *= exp(dm)``, sigma = log(factor))
* facet directions  phi~U(-fov/2,fov/2), theta~U(0,360)
                                          -> ``astro/real_data.py:529-531``
* example antenna layout N(0,(40 km)^2)   -> ``astro/radio_array.py:133-135``
* 8 s time cadence                        -> ``astro/real_data.py:526``
* straight rays to the z = tmax plane, tmax = 1000 km
                                          -> ``inversion/inversion_pipeline.py:44``
"""
import os

import numpy as np

from .ionosphere.iri import chapman_profile      # noqa: F401  (the four-layer a-priori profile lives with the reference's module name)
from scipy.special import gamma

_HERE = os.path.dirname(os.path.abspath(__file__))
LOFAR_HBA_CSV = os.path.join(_HERE, "astro", "arrays", "lofar_hba_stations.csv")

EARTH_ROT_RATE = 7.2921e-5  # rad / s
TIME_CADENCE = 8.0  # s


def matern52_field(xvec, yvec, zvec, sigma, corr, seed):
    """Matern-5/2 Gaussian random field on a uniform grid, normalised so that
    ``std == sigma``.  Restates ``ionosphere/simulation.py:51-112`` including
    its legacy ``np.random.seed`` stream (``RandomState(seed)`` draws the same
    numbers), its half-spectrum frequency axes and the checkerboard sign flip."""
    nx, ny, nz = len(xvec), len(yvec), len(zvec)
    dx, dy, dz = xvec[1] - xvec[0], yvec[1] - yvec[0], zvec[1] - zvec[0]
    sx, sy, sz = 1.0 / (dx * nx), 1.0 / (dy * ny), 1.0 / (dz * nz)
    l = np.linspace(0, sx * nx / 2.0, nx)
    m = np.linspace(0, sy * ny / 2.0, ny)
    n = np.linspace(0, sz * nz / 2.0, nz)
    s2 = (l ** 2)[:, None, None] + (m ** 2)[None, :, None] + (n ** 2)[None, None, :]
    s2 = np.fft.ifftshift(s2)
    d, nu = 3.0, 2.5
    S = (sigma ** 2 * 2 ** d * np.pi ** (d / 2.0) * gamma(nu + d / 2.0) * (2 * nu) ** nu
         / gamma(nu) / corr ** (2 * nu)
         * (2 * nu / corr ** 2 + 4 * np.pi ** 2 * s2) ** (-nu - d / 2.0))
    S = np.sqrt(S)
    rs = np.random.RandomState(seed)
    Z = rs.normal(size=S.shape) + 1j * rs.normal(size=S.shape)
    B = np.fft.ifftn(S * Z, (nx, ny, nz), axes=(0, 1, 2)).real * (sx * nx) * (sy * ny) * (sz * nz)
    B[::2, :, :] *= -1
    B[:, ::2, :] *= -1
    B[:, :, ::2] *= -1
    B *= sigma / np.std(B)
    return B


def itrs_to_enu_km(xyz_m):
    """ITRS metres -> local East/North/Up km about the centroid (plain WGS-84
    geodetic latitude/longitude of the centroid, closed-form Bowring)."""
    xyz = np.asarray(xyz_m, dtype=np.float64)
    c = xyz.mean(axis=0)
    a, f = 6378137.0, 1.0 / 298.257223563
    b = a * (1 - f)
    e2, ep2 = 1 - (b / a) ** 2, (a / b) ** 2 - 1
    p = np.hypot(c[0], c[1])
    th = np.arctan2(c[2] * a, p * b)
    lat = np.arctan2(c[2] + ep2 * b * np.sin(th) ** 3, p - e2 * a * np.cos(th) ** 3)
    lon = np.arctan2(c[1], c[0])
    sl, cl, so, co = np.sin(lat), np.cos(lat), np.sin(lon), np.cos(lon)
    R = np.array([[-so, co, 0.0], [-sl * co, -sl * so, cl], [cl * co, cl * so, sl]])
    return (xyz - c) @ R.T / 1000.0


def read_array_table(path):
    """Whitespace table ``X Y Z diameter label`` with ``#`` comments
    (format of ``astro/arrays/lofar.hba.antenna.cfg``).  Returns
    (xyz[N,3] metres, diameters[N] or None, labels[N])."""
    xyz, diam, labels = [], [], []
    with open(path) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            tok = line.replace(",", " ").split()
            xyz.append([float(tok[0]), float(tok[1]), float(tok[2])])
            diam.append(float(tok[3]) if len(tok) > 3 else np.nan)
            labels.append(tok[4] if len(tok) > 4 else "ant{:02d}".format(len(labels)))
    diam = np.array(diam)
    return np.array(xyz), (None if np.all(np.isnan(diam)) else diam), np.array(labels)


def enu_rotation(lon, lat):
    """Rows (east, north, up) in ITRS axes at geodetic longitude / latitude (radians)."""
    sl, cl, so, co = np.sin(lat), np.cos(lat), np.sin(lon), np.cos(lon)
    return np.array([[-so, co, 0.0], [-sl * co, -sl * so, cl], [cl * co, cl * so, sl]])


def read_station_enu_csv(path):
    """Station table stored as East/North/Up offsets (metres) from a reference ITRS point given in the
    header (``# ref_itrs_m = X Y Z`` and ``# ref_lon_lat_rad = lon lat``).  Returns
    (itrs_xyz[N,3] metres, diameters[N], labels[N])."""
    ref, lonlat, rows = None, None, []
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith("#"):
                if "ref_itrs_m" in line:
                    ref = np.array([float(t) for t in line.split("=")[1].split()])
                if "ref_lon_lat_rad" in line:
                    lonlat = [float(t) for t in line.split("=")[1].split()]
                continue
            if not line or line.startswith("station"):
                continue
            rows.append(line.split(","))
    enu = np.array([[float(r[1]), float(r[2]), float(r[3])] for r in rows])
    xyz = ref + enu @ enu_rotation(*lonlat)
    return xyz, np.array([float(r[4]) for r in rows]), np.array([r[0] for r in rows])


def lofar_enu_km():
    """The 62 LOFAR-HBA stations (``astro/arrays/lofar.hba.antenna.cfg`` in the reference) as ENU km about
    their centroid."""
    xyz, _, _ = read_station_enu_csv(LOFAR_HBA_CSV)
    return itrs_to_enu_km(xyz)


def example_antennas_km(n, seed=0):
    rng = np.random.default_rng(seed)
    a = np.zeros((n, 3))
    a[:, :2] = rng.normal(scale=40.0, size=(n, 2))
    return a


def facet_directions(nd, fov_deg=4.0, seed=1):
    rng = np.random.default_rng(seed)
    phi = np.deg2rad(rng.uniform(-fov_deg / 2.0, fov_deg / 2.0, nd))
    theta = np.deg2rad(rng.uniform(0.0, 360.0, nd))
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], -1)


def rotate_about_pole(dirs, nt, colat_deg=37.0):
    """Directions [Nd,3] -> [Nt,Nd,3]: timestep t rotates every direction about the
    celestial-pole axis (tilted ``colat_deg`` from local up toward north) by
    t * 8 s * Earth rate."""
    k = np.array([0.0, np.sin(np.deg2rad(colat_deg)), np.cos(np.deg2rad(colat_deg))])
    out = np.empty((nt,) + dirs.shape)
    for t in range(nt):
        a = t * TIME_CADENCE * EARTH_ROT_RATE
        out[t] = (dirs * np.cos(a) + np.cross(k, dirs) * np.sin(a)
                  + np.outer(dirs @ k, k) * (1 - np.cos(a)))
    return out


def ray_bundle(antennas_km, dirs_tnd):
    """origins, directions as [Na,Nt,Nd,3] (``geometry/calc_rays.py:122-139`` layout)."""
    na, (nt, nd, _) = antennas_km.shape[0], dirs_tnd.shape
    origins = np.broadcast_to(antennas_km[:, None, None, :], (na, nt, nd, 3)).copy()
    directions = np.broadcast_to(dirs_tnd[None], (na, nt, nd, 3)).copy()
    return origins, directions


def domain_for(origins, directions, n, tmax=1000.0, margin_cells=4):
    """Uniform n^3 grid enclosing every straight ray up to z = tmax with a margin,
    in the spirit of ``inversion/initial_model.py:13-36``."""
    o = origins.reshape(-1, 3)
    d = directions.reshape(-1, 3)
    end = o + d * ((tmax - o[:, 2]) / d[:, 2])[:, None]
    lo = np.minimum(o.min(0), end.min(0))
    hi = np.maximum(o.max(0), end.max(0))
    vecs = []
    for a in range(3):
        span = hi[a] - lo[a]
        pad = span * margin_cells / (n - 1 - 2 * margin_cells)
        vecs.append(np.linspace(lo[a] - pad, hi[a] + pad, n))
    return vecs


def ne_model(xvec, yvec, zvec, seed=1234, factor=2.0, corr=20.0, turbulent=True):
    """ne [m^-3] = Chapman(z) * exp(Matern52 field, sigma = log(factor))."""
    prof = chapman_profile(np.maximum(zvec, 0.0), 45.0)
    ne = np.broadcast_to(prof[None, None, :], (len(xvec), len(yvec), len(zvec))).copy()
    if turbulent:
        ne *= np.exp(matern52_field(xvec, yvec, zvec, np.log(factor), corr, seed))
    return ne


CONFIGS = {
    # name: (antennas, Na, Nd, Nt, n)
    "cfg1": ("example", 8, 8, 1, 64),
    "cfg2": ("lofar", 62, 42, 1, 128),
    "cfg2b": ("lofar", 62, 42, 1, 256),
    "cfg4": ("lofar", 62, 100, 100, 256),
}


def make_workload(name=None, antennas="lofar", na=62, nd=42, nt=1, n=128, tmax=1000.0,
                  seed=1234, turbulent=True, margin_cells=4):
    """Everything the forward model needs, as plain numpy float64 arrays."""
    if name is not None:
        antennas, na, nd, nt, n = CONFIGS[name]
    ants = lofar_enu_km()[:na] if antennas == "lofar" else example_antennas_km(na, 0)
    dirs = rotate_about_pole(facet_directions(nd, 4.0, 1), nt)
    origins, directions = ray_bundle(ants, dirs)
    xvec, yvec, zvec = domain_for(origins, directions, n, tmax, margin_cells)
    ne = ne_model(xvec, yvec, zvec, seed=seed, turbulent=turbulent)
    K_ne = float(np.median(ne))
    return dict(xvec=xvec, yvec=yvec, zvec=zvec, ne=ne, K_ne=K_ne, m=np.log(ne / K_ne),
                origins=origins, directions=directions, tmax=float(tmax), Ns=n + 1, i0=0)

"""A-priori electron-density profiles (ionosphere/iri.py).  ``a_priori_model_`` is the reference's self-contained four-layer
Chapman model (:20-68), restated here; the IRI-2016 variant (``a_priori_model``, :8-16) needs the pyiri2016 Fortran package and
is not part of this build."""
import numpy as np


def chapman_profile(h, zenith=45.0, thin_f=False):
    """Electron density [m^-3] at heights ``h`` [km] for solar zenith angle
    ``zenith`` [deg]: D + E + F1 + F2 Chapman layers.
    Restates ``ionosphere/iri.py:20-68``."""
    h = np.asarray(h, dtype=np.float64)

    def peak_density(n0, dn, tau, b):
        y = zenith / tau
        return n0 + dn * np.exp(-y ** 2) / (1.0 + y ** (2 * b))

    def peak_height(z0, dz, rho, chi0):
        return z0 + dz / (1.0 + np.exp(-(zenith - chi0) / rho))

    def layer(nm, zm, H):
        y = (h - zm) / H
        return nm * np.exp(0.5 * (1.0 - y - np.exp(-y)))

    y = zenith / 58.0
    nm_d = 4e8 + 5.9e8 * np.exp(-y ** 2) if y < 1 else 4e8
    n = layer(nm_d, peak_height(81.0, 7.0, 7.46, 100.0), 8.0)
    n = n + layer(peak_density(1.6e9, 1.6e11, 87.0, 8.7), 110.0, 11.0)
    H_f1, H_f2 = (20.0, 27.5) if thin_f else (40.0, 55.0)
    n = n + layer(peak_density(2.0e11, 9.1e10, 54.0, 13.6), 185.0, H_f1)
    n = n + layer(peak_density(7.7e10, 4.4e11, 111.0, 4.8),
                  peak_height(242.0, 75.0, 7.46, 96.0), H_f2)
    return np.atleast_1d(n)


def a_priori_model_(h, zenith, thin_f=False):
    """Electron density [m^-3] at heights ``h`` [km] for solar zenith angle ``zenith`` [deg] (the reference's entry-point name)."""
    return chapman_profile(h, zenith, thin_f)


def a_priori_model(heights, hmax, lat, lon, time):
    raise NotImplementedError("IRI-2016 profiles need pyiri2016 (not available); use a_priori_model_(h, zenith)")

"""A-priori electron-density profiles (ionosphere/iri.py).  ``a_priori_model_`` is the reference's
self-contained four-layer Chapman model (:20-68); the IRI-2016 variant (``a_priori_model``, :8-16) needs
the pyiri2016 Fortran package and is not part of this build."""
from ..synthetic import chapman_profile


def a_priori_model_(h, zenith, thin_f=False):
    """Electron density [m^-3] at heights ``h`` [km] for solar zenith angle ``zenith`` [deg]."""
    return chapman_profile(h, zenith, thin_f)


def a_priori_model(heights, hmax, lat, lon, time):
    raise NotImplementedError("IRI-2016 profiles need pyiri2016 (not available); use a_priori_model_(h, zenith)")

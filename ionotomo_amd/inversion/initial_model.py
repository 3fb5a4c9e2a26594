"""A-priori model builders around a DataPack -- ionotomo.inversion.initial_model (inversion/initial_model.py:13-84)
without astropy / pyiri2016.  Host-side set-up that runs once per inversion: it sizes the box with
``determine_inversion_domain``, fills it with the reference's self-contained Chapman-layer profile
``a_priori_model_`` evaluated at WGS-84 geodetic height (the IRI-2016 profile the reference calls at :66 needs the
pyiri2016 Fortran package), and optionally multiplies by a log-normal Matern-5/2 realisation (:75-84).
"""
import numpy as np

from ..astro.frames import (determine_inversion_domain, geodetic_from_itrs, gmst_rad, icrs_direction_in_itrs,
                            itrs_to_pointing_km, pointing_rotation)
from ..geometry.tri_cubic import TriCubic
from ..ionosphere.iri import a_priori_model_
from ..ionosphere.simulation import turbulent_perturbation

__all__ = ["determine_inversion_domain", "turbulent_perturbation", "create_initial_model", "create_turbulent_model",
           "model_frame_of"]


def model_frame_of(datapack, time_idx=-1):
    """(centre_itrs_m, phase_radec, fixtime_unix, R) of the datapack's model frame: the Pointing axes at the
    middle selected time (inversion/initial_model.py:52-56)."""
    times, _ = datapack.get_times(time_idx=time_idx)
    fixtime = float(times[len(times) >> 1])
    centre = np.asarray(datapack.radio_array.get_center(), dtype=np.float64)
    phase = datapack.get_center_direction()
    lon, _, _ = geodetic_from_itrs(centre)
    R = pointing_rotation(lon, gmst_rad(fixtime) + lon, phase[0], phase[1])
    return centre, phase, fixtime, R


def create_initial_model(datapack, ant_idx=-1, time_idx=-1, dir_idx=-1, zmax=1000., spacing=5., padding=20, zenith=45.,
                         thin_f=False):
    """TriCubic of the a-priori electron density [m^-3] on the inversion box (inversion/initial_model.py:43-73).
    ``zenith`` is the solar zenith angle [deg] handed to the Chapman profile (the reference derives it from
    astropy's sun position, :58)."""
    antennas, _ = datapack.get_antennas(ant_idx=ant_idx)
    patches, _ = datapack.get_directions(dir_idx=dir_idx)
    centre, phase, fixtime, R = model_frame_of(datapack, time_idx)
    ants_km = itrs_to_pointing_km(antennas, centre, R)
    dirs = icrs_direction_in_itrs(patches[:, 0], patches[:, 1], fixtime) @ R.T
    xvec, yvec, zvec = determine_inversion_domain(spacing, ants_km, dirs, zmax, padding=padding)
    X, Y, Z = np.meshgrid(xvec, yvec, zvec, indexing='ij')
    itrs = np.stack([X.ravel(), Y.ravel(), Z.ravel()], -1) * 1000.0 @ R + centre
    _, _, h = geodetic_from_itrs(itrs.T)
    ne = a_priori_model_(h / 1000.0, zenith, thin_f=thin_f).reshape(X.shape)
    return TriCubic(xvec, yvec, zvec, ne)


def create_turbulent_model(datapack, factor=2., corr=20., seed=None, **initial_model_kwargs):
    """A-priori model times exp(dm), dm a Matern-5/2 field with std log(factor) (inversion/initial_model.py:75-84)."""
    ne_tci = create_initial_model(datapack, **initial_model_kwargs)
    dm = turbulent_perturbation(ne_tci, sigma=np.log(factor), corr=corr, seed=seed)
    ne_tci.M = ne_tci.M * np.exp(dm)
    return ne_tci

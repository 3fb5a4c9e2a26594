"""ionotomo_amd -- MI355X-native ray-integral engine with IonoTomo's hot-path API surface.

Mirrors the names ``ionotomo/__init__.py:1-29`` re-exports for this path.  Importing the package
never touches the GPU; the first numeric call loads ``libionotomo_hip.so`` and raises if it (or a
GPU) is missing -- there is no CPU fallback.
"""
from .geometry.tri_cubic import TriCubic, bisection
from .geometry.calc_rays import calc_rays, calc_rays_dask, cast_ray
from .inversion.fermat import Fermat
from .inversion.forward_equation import forward_equation, forward_equation_dask, do_forward_equation
from .inversion.gradient import compute_gradient, compute_gradient_dask
from .astro.radio_array import RadioArray, generate_example_radio_array
from .astro.real_data import DataPack, generate_example_datapack, phase_screen_datapack
from .astro.simulate_observables import simulate_phase
from .inversion.initial_model import create_initial_model, create_turbulent_model, determine_inversion_domain
from .tomography.linear_operators import RayOp, TECForwardEquation
from .ionosphere.covariance import Covariance
from .ionosphere.simulation import IonosphereSimulation
from .ionosphere.iri import a_priori_model_
from .utils.timer import clock
from ._lib import Context, default_context

__all__ = ["TriCubic", "bisection", "calc_rays", "calc_rays_dask", "cast_ray", "Fermat", "forward_equation",
           "forward_equation_dask", "do_forward_equation", "compute_gradient", "compute_gradient_dask", "RadioArray",
           "generate_example_radio_array", "DataPack", "generate_example_datapack",
           "phase_screen_datapack", "simulate_phase", "create_initial_model", "create_turbulent_model",
           "determine_inversion_domain", "RayOp", "TECForwardEquation", "Covariance", "IonosphereSimulation",
           "a_priori_model_", "clock", "Context", "default_context"]

"""``IonosphereSimulation`` -- Matern-5/2 Gaussian random field used to perturb the a-priori model
(ionosphere/simulation.py:45-112; used by inversion/initial_model.py:37-84).  Host-side input
generation (numpy FFT), pinned to the reference's own realisation in tests/test_oracle_golden.py."""
import numpy as np

from ..synthetic import matern52_field


class IonosphereSimulation(object):
    def __init__(self, xvec, yvec, zvec, sigma, corr, type='m52'):
        assert type in ['m52']
        self.xvec, self.yvec, self.zvec = np.asarray(xvec), np.asarray(yvec), np.asarray(zvec)
        self.nx, self.ny, self.nz = self.xvec.size, self.yvec.size, self.zvec.size
        self.sigma, self.corr, self.type = sigma, corr, type

    def realization(self, seed=None):
        """Gaussian random field with std == sigma (same numbers as the reference for the same seed)."""
        if seed is None:
            seed = np.random.randint(0, 2 ** 31 - 1)
        return matern52_field(self.xvec, self.yvec, self.zvec, self.sigma, self.corr, seed)


def turbulent_perturbation(tci, sigma=3., corr=20., seed=None):
    """inversion/initial_model.py:37-40."""
    return IonosphereSimulation(tci.xvec, tci.yvec, tci.zvec, sigma, corr, type='m52').realization(seed=seed)

"""``IonosphereSimulation`` -- Matern-5/2 Gaussian random field used to perturb the a-priori model
(ionosphere/simulation.py:45-112; used by inversion/initial_model.py:37-84).  Host-side input
generation (numpy FFT), pinned to the reference's own realisation in tests/test_oracle_golden.py.
``realization_device`` is the same construction with the spectrum shaping, the inverse FFT (hipFFT through
torch.fft), the checkerboard sign and the normalisation on the GPU, returning a device tensor ready for
``RayEngine.set_values`` -- for cfg4-scale synthetic inputs."""
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

    def realization_device(self, seed=None, device=0, host_draws=True):
        """The same field as ``realization(seed)`` as a float64 tensor on ``cuda:device``.  With
        ``host_draws`` the white noise comes from numpy's legacy stream (identical numbers to the reference for
        the same seed); without, from torch's device generator (a different, much faster stream)."""
        import torch
        from scipy.special import gamma
        dev = torch.device("cuda", device)
        if not torch.cuda.is_available():
            raise RuntimeError("realization_device needs a GPU (use realization() for the host version)")
        if seed is None:
            seed = np.random.randint(0, 2 ** 31 - 1)
        n = (self.nx, self.ny, self.nz)
        step = [float(v[1] - v[0]) for v in (self.xvec, self.yvec, self.zvec)]
        axes = [torch.linspace(0.0, 0.5 / step[a], n[a], dtype=torch.float64, device=dev) for a in range(3)]
        s2 = (axes[0] ** 2)[:, None, None] + (axes[1] ** 2)[None, :, None] + (axes[2] ** 2)[None, None, :]
        s2 = torch.fft.ifftshift(s2)
        d, nu, corr, sigma = 3.0, 2.5, float(self.corr), float(self.sigma)
        amp = (sigma ** 2 * 2 ** d * np.pi ** (d / 2.0) * gamma(nu + d / 2.0) * (2 * nu) ** nu / gamma(nu)
               / corr ** (2 * nu))
        S = torch.sqrt(amp * (2 * nu / corr ** 2 + 4 * np.pi ** 2 * s2) ** (-nu - d / 2.0))
        if host_draws:
            rs = np.random.RandomState(seed)
            re = torch.from_numpy(rs.normal(size=n)).to(dev)
            im = torch.from_numpy(rs.normal(size=n)).to(dev)
        else:
            gen = torch.Generator(device=dev).manual_seed(int(seed))
            re = torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
            im = torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
        B = torch.fft.ifftn(torch.complex(S * re, S * im), dim=(0, 1, 2)).real
        sign = [1.0 - 2.0 * ((torch.arange(n[a], device=dev) + 1) % 2).to(torch.float64) for a in range(3)]
        B = B * sign[0][:, None, None] * sign[1][None, :, None] * sign[2][None, None, :]
        return B * (sigma / torch.std(B, unbiased=False))


def turbulent_perturbation(tci, sigma=3., corr=20., seed=None):
    """inversion/initial_model.py:37-40."""
    return IonosphereSimulation(tci.xvec, tci.yvec, tci.zvec, sigma, corr, type='m52').realization(seed=seed)

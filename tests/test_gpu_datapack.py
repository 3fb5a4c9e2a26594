"""Callers either side of the path: simulate_phase over a DataPack, the full objective
with prior terms, C_m^{-1}.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def small_datapack():
    from ionotomo_amd import generate_example_datapack
    return generate_example_datapack(Nant=7, Ntime=3, Ndir=5, Nfreqs=3, fov=4., alt=80., az=20., time=1.49e9, seed=11)


def test_simulate_phase_matches_oracle(O, tmp_path, monkeypatch):
    """astro/simulate_observables.py:22-66 end to end: turbulent model -> Pointing-frame rays -> phase forward
    -> datapack, against the oracle's phase model on the same rays and the same random draws."""
    import ionotomo_amd as it
    from ionotomo_amd.inversion.initial_model import model_frame_of
    monkeypatch.chdir(tmp_path)
    dp = small_datapack()
    tci = it.create_turbulent_model(dp, factor=2., corr=20., seed=5, spacing=10., padding=8)
    assert tci.M.min() > 0 and tci.nz > 100
    out = it.simulate_phase(dp.clone(), ne_tci=tci.copy(), datafolder="sim", seed=7)
    saved = it.TriCubic(filename=str(tmp_path / "sim" / "turbulent_ne.npz"))
    assert np.array_equal(saved.M, tci.M) and np.array_equal(saved.zvec, tci.zvec)
    # oracle replay
    centre, phase, fixtime, _ = model_frame_of(dp)
    rays = it.calc_rays(dp.antennas, dp.directions, dp.times, centre, fixtime, phase, tci, dp.freqs[1], True, 1000., None)
    assert rays.shape == (7, 3, 5, 4, tci.nz)
    rng = np.random.default_rng(7)
    clock, const = 5e-9 * rng.normal(size=[7, 3]), np.pi / 2. * np.pi * rng.normal(size=7)
    ref = O.forward_phase(np.log(tci.M / 1e11), clock, const, tci.xvec, tci.yvec, tci.zvec, rays, dp.freqs, K=1e11, i0=0)
    ref = ref - ref[0]
    assert out.ref_ant == dp.antenna_labels[0]
    assert np.max(np.abs(out.phase - ref)) < 1e-9 * np.max(np.abs(ref))
    # sub-selection + flag_remaining keeps only what was simulated
    sub = it.simulate_phase(dp.clone(), ne_tci=tci.copy(), ant_idx=[0, 2, 5], time_idx=[1], dir_idx=[0, 4], freq_idx=[2],
                            flag_remaining=True, seed=7)
    assert (sub.Na, sub.Nt, sub.Nd, sub.Nf) == (3, 1, 2, 1)
    assert np.all(np.isfinite(sub.phase)) and np.all(sub.phase[0] == 0) and np.any(sub.phase[1:] != 0)


def test_device_realisation_equals_host_realisation():
    """ionosphere/simulation.py:94-112 on the GPU (hipFFT): same numbers as the host construction that the
    golden fixture pins to the reference."""
    import torch
    from ionotomo_amd import IonosphereSimulation
    sim = IonosphereSimulation(np.linspace(-50, 50, 24), np.linspace(-40, 45, 30), np.linspace(0, 300, 45), 0.7, 20.)
    host = sim.realization(seed=123)
    dev = sim.realization_device(seed=123)
    assert dev.is_cuda and dev.dtype == torch.float64
    assert np.max(np.abs(dev.cpu().numpy() - host)) < 1e-10 * np.max(np.abs(host))
    fast = sim.realization_device(seed=123, host_draws=False)
    assert abs(float(fast.std(unbiased=False)) - 0.7) < 1e-12 and not torch.allclose(fast, dev)

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


def test_phase_observable_on_the_device_and_its_adjoint():
    """iono_forward_phase_straight_dev / iono_adjoint_phase_straight_dev (samples generated in-kernel, the reference's
    rays[Na,Nt,Nd,4,Ns] never exists) against oracle.forward_phase / oracle.gradient_phase
    (inversion/iterative_newton.py:86-127): ideal-uniform grid (LDS-tiled transpose) and a non-uniform one (general
    tier), 3 and 10 frequencies (two passes of 8), with / without walk order, plus a finite-difference check."""
    import torch
    from oracle import oracle as O
    from ionotomo_amd.engine import RayEngine
    from ionotomo_amd import synthetic as syn
    rng = np.random.default_rng(0)
    w = syn.make_workload(antennas="lofar", na=12, nd=5, nt=3, n=40)
    na, nt, nd = 12, 3, 5
    o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
    xv_nu = w["xvec"] + 0.2 * (w["xvec"][1] - w["xvec"][0]) * np.sin(np.arange(40))        # a non-uniform variant of the x axis
    for xv, Ns in ((w["xvec"], 65), (w["xvec"], 70), (xv_nu, 33)):
        # (the last set reaches ne / n_p = 0.1: the kernels' binomial series for 1 - sqrt(1 - x) hands over to the square root)
        for freqs in (np.array([120e6, 150e6, 180e6]), np.linspace(110e6, 190e6, 10), np.array([25e6, 30e6, 40e6, 60e6, 150e6])):
            eng = RayEngine(0)
            eng.set_grid(xv, w["yvec"], w["zvec"])
            mu = np.log(w["ne"] / 1e11) + 0.05 * rng.normal(size=w["ne"].shape)
            eng.set_log_model(eng.tensor(mu), 1e11)
            clock, const = rng.normal(size=(na, nt)) * 1e-9, rng.normal(size=na)
            ot, dt = eng.tensor(o), eng.tensor(d)
            g = eng.forward_phase(ot, dt, na, nt, nd, w["tmax"], Ns, freqs, eng.tensor(clock), eng.tensor(const), 2).cpu().numpy()
            assert not eng.check_oob()
            rays = O.straight_rays(w["origins"], w["directions"], w["tmax"], Ns)
            ref = O.forward_phase(mu, clock, const, xv, w["yvec"], w["zvec"], rays, freqs, K=1e11, i0=2)
            assert np.max(np.abs(g - ref)) < 1e-11 * np.max(np.abs(ref))
            y = rng.normal(size=ref.shape)
            gref = O.gradient_phase(mu, xv, w["yvec"], w["zvec"], rays, freqs, y, K=1e11, i0=2)
            yt = eng.tensor(y.reshape(na, nt * nd, freqs.size))
            order = eng.locality_order(ot, dt, w["tmax"])
            for ordr in (None, order):
                gg = eng.adjoint_phase(ot, dt, yt, na, w["tmax"], Ns, freqs, 2, order=ordr).cpu().numpy()
                assert np.max(np.abs(gg - gref)) < 1e-10 * np.max(np.abs(gref))
            gne = eng.adjoint_phase(ot, dt, yt, na, w["tmax"], Ns, freqs, 2, wrt_log_model=False).cpu().numpy()
            assert np.max(np.abs(gne * np.exp(mu) * 1e11 - gref)) < 1e-10 * np.max(np.abs(gref))
            # the node-stationary (box-binned) transpose once the geometry is planned: 1, 2, 4 and 8 frequencies per pass
            if eng.plan_adjoint(ot, dt, w["tmax"], Ns)[0] > 0:
                gp = eng.adjoint_phase(ot, dt, yt, na, w["tmax"], Ns, freqs, 2).cpu().numpy()
                assert np.max(np.abs(gp - gref)) < 1e-10 * np.max(np.abs(gref))
                for nf in sorted({1, 2, min(5, freqs.size)}):
                    fs, ys = freqs[:nf], np.ascontiguousarray(y[..., :nf])
                    gr = O.gradient_phase(mu, xv, w["yvec"], w["zvec"], rays, fs, ys, K=1e11, i0=2)
                    gq = eng.adjoint_phase(ot, dt, eng.tensor(ys.reshape(na, nt * nd, nf)), na, w["tmax"], Ns, fs, 2).cpu().numpy()
                    assert np.max(np.abs(gq - gr)) < 1e-10 * np.max(np.abs(gr)), nf
                    gf = eng.forward_phase(ot, dt, na, nt, nd, w["tmax"], Ns, fs, eng.tensor(clock), eng.tensor(const), 2).cpu().numpy()
                    assert np.max(np.abs(gf - ref[..., :nf])) < 1e-11 * np.max(np.abs(ref))
                eng.clear_adjoint_plan()
            else:
                assert xv is xv_nu
    # finite difference of S = 1/2 sum (g - dobs)^2 / CdCt through the device path
    dobs, CdCt = ref + rng.normal(size=ref.shape) * 0.1, rng.uniform(0.5, 2.0, size=ref.shape) * 0.01
    ct, kt = eng.tensor(clock), eng.tensor(const)

    def S(m):
        eng.set_log_model(eng.tensor(m), 1e11)
        gm = eng.forward_phase(ot, dt, na, nt, nd, w["tmax"], Ns, freqs, ct, kt, 2).cpu().numpy()
        return 0.5 * np.sum((gm - dobs) ** 2 / CdCt), gm
    S0, g0 = S(mu)
    grad = eng.adjoint_phase(ot, dt, eng.tensor(((g0 - dobs) / CdCt).reshape(na, nt * nd, freqs.size)), na, w["tmax"], Ns, freqs,
                             2).cpu().numpy()
    for f in np.argsort(-np.abs(grad.ravel()))[:4]:
        e = np.zeros(mu.size)
        e[f] = 1e-5
        fd = (S(mu + e.reshape(mu.shape))[0] - S(mu - e.reshape(mu.shape))[0]) / 2e-5
        assert abs(fd - grad.ravel()[f]) < 1e-4 * abs(grad.ravel()[f]) + 1e-8


def test_phase_inversion_descends():
    """solvers.steepest_descent_phase: the reference's objective on the phase observable falls monotonically from the
    prior towards a perturbed truth, every iterate obtained on the device (no rays array)."""
    import torch
    from ionotomo_amd import solvers, synthetic as syn
    from ionotomo_amd.engine import RayEngine
    w = syn.make_workload(antennas="lofar", na=16, nd=6, nt=3, n=32)
    na, nt, nd, Ns = 16, 3, 6, 33
    eng = RayEngine(0)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    X, Y, Z = np.meshgrid(w["xvec"], w["yvec"], w["zvec"], indexing="ij")
    mu_prior = np.log(w["ne"] / 1e11)
    mu_true = mu_prior + 0.3 * np.exp(-(X ** 2 + Y ** 2) / 20.0 ** 2 - ((Z - 300) / 100.0) ** 2)
    freqs = np.array([120e6, 140e6, 160e6, 180e6])
    o, d = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
    clock, const = torch.zeros(na, nt, dtype=torch.float64, device=eng.device), torch.zeros(na, dtype=torch.float64, device=eng.device)
    eng.set_log_model(eng.tensor(mu_true), 1e11)
    dobs = eng.forward_phase(o, d, na, nt, nd, w["tmax"], Ns, freqs, clock, const, 0).clone()
    CdCt = torch.full_like(dobs, 1e-4)
    mu, hist = solvers.steepest_descent_phase(eng, o, d, na, nt, nd, w["tmax"], Ns, freqs, clock, const, dobs, CdCt,
                                              eng.tensor(mu_prior), K=1e11, i0=0, max_iter=12)
    # (the reference rule stops after its 5 mandatory updates here: steps fall below pgtol = 1e-2)
    assert len(hist) >= 6 and all(b <= a * (1 + 1e-12) for a, b in zip(hist, hist[1:])) and hist[-1] < 0.5 * hist[0]
    assert not eng.check_oob()


def test_simulate_phase_on_a_datapack_built_from_reference_typed_members(tmp_path, monkeypatch):
    """The reference builds a DataPack from astropy objects (astro/real_data.py:124-131: ITRS antennas, ICRS directions, Time) and
    hands it to simulate_phase (astro/simulate_observables.py:22-66).  A DataPack made from stand-ins with exactly those attributes
    (tests/astropy_standins.py) must simulate the same phases, bit for bit, as the one made from plain arrays."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from astropy_standins import ICRSCoord, ITRSCoord, Time
    import ionotomo_amd as it
    from ionotomo_amd.astro.real_data import DataPack
    monkeypatch.chdir(tmp_path)
    dp = small_datapack()
    d = dp.get_data_dict()
    typed = DataPack(dict(d, antennas=ITRSCoord(dp.antennas), directions=ICRSCoord(dp.directions[:, 0], dp.directions[:, 1]),
                          times=Time(dp.times), timestamps=None))
    assert list(typed.timestamps) == list(dp.timestamps)
    tci = it.create_turbulent_model(dp, factor=2., corr=20., seed=5, spacing=10., padding=8)
    tci2 = it.create_turbulent_model(typed, factor=2., corr=20., seed=5, spacing=10., padding=8)
    assert np.array_equal(tci.M, tci2.M) and np.array_equal(tci.xvec, tci2.xvec)
    a = it.simulate_phase(dp.clone(), ne_tci=tci.copy(), seed=7)
    b = it.simulate_phase(typed, ne_tci=tci2.copy(), seed=7)
    assert np.array_equal(a.phase, b.phase) and a.ref_ant == b.ref_ant

"""``simulate_phase`` -- forward-simulate a DataPack through the GPU ray-integral path
(astro/simulate_observables.py:22-98; SURVEY.md section 3, call stack B).

rays = calc_rays(...) (one launch of the tracer), phase = iterative_newton.forward_equation(...) (one launch
of the phase kernel), written back into the datapack and referenced to the first selected antenna.
"""
import os

import numpy as np

from ..geometry.calc_rays import calc_rays
from ..inversion.initial_model import create_turbulent_model, model_frame_of
from ..inversion.iterative_newton import forward_equation


def simulate_phase(datapack, ne_tci=None, num_threads=1, datafolder=None, ant_idx=-1, time_idx=-1, dir_idx=-1,
                   freq_idx=-1, do_plot_datapack=False, flag_remaining=False, seed=None):
    """Same arguments as the reference; ``num_threads`` and ``do_plot_datapack`` are accepted and ignored
    (one GPU launch replaces the thread pool; plotting is outside this build).  ``seed`` makes the clock /
    constant draws and the turbulent model reproducible."""
    rng = np.random.default_rng(seed)
    if ne_tci is None:
        ne_tci = create_turbulent_model(datapack, factor=2., corr=20., seed=seed)
    if datafolder is not None:
        os.makedirs(os.path.join(os.getcwd(), datafolder), exist_ok=True)
        ne_tci.save(os.path.join(os.getcwd(), datafolder, "turbulent_ne.npz"))
    antennas, antenna_labels = datapack.get_antennas(ant_idx=ant_idx)
    patches, patch_names = datapack.get_directions(dir_idx=dir_idx)
    times, timestamps = datapack.get_times(time_idx=time_idx)
    freqs = datapack.get_freqs(freq_idx=freq_idx)
    Na, Nt, Nf = len(antennas), len(times), len(freqs)
    centre, phase, fixtime, _ = model_frame_of(datapack, time_idx)
    rays = calc_rays(antennas, patches, times, centre, fixtime, phase, ne_tci, freqs[Nf >> 1], True, 1000., None)
    model = (np.log(ne_tci.M / 1e11), 5e-9 * rng.normal(size=[Na, Nt]), np.pi / 2. * np.pi * rng.normal(size=Na))
    dobs = forward_equation(model, ne_tci, rays, freqs, K=1e11, i0=0)
    datapack.set_phase(dobs, ant_idx=ant_idx, time_idx=time_idx, dir_idx=dir_idx, freq_idx=freq_idx)
    datapack.set_reference_antenna(antenna_labels[0])
    if flag_remaining:
        all_ants, all_dirs, all_times = datapack.antenna_labels, datapack.patch_names, datapack.timestamps
        drop_f = [l for l, f in enumerate(datapack.freqs) if f not in freqs]
        datapack.flag_antennas([a for a in all_ants if a not in antenna_labels])
        datapack.flag_times([t for t in all_times if t not in timestamps])
        datapack.flag_directions([d for d in all_dirs if d not in patch_names])
        datapack.flag_freqs(drop_f)
    return datapack

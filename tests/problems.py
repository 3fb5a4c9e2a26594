"""Small seeded inversion problems shared by the solver / distributed tests."""
import numpy as np

from ionotomo_amd import synthetic as syn


def small_problem(na=5, nd=4, nt=3, n=14, Ns=15, seed=0, i0=1):
    w = syn.make_workload(antennas="example", na=na, nd=nd, nt=nt, n=n)
    o = w["origins"].reshape(na, nt * nd, 3)
    d = w["directions"].reshape(na, nt * nd, 3)
    rng = np.random.default_rng(seed)
    x_true = w["ne"] / 1e13 * np.exp(0.2 * rng.normal(size=w["ne"].shape))
    x0 = w["ne"] / 1e13
    return dict(w=w, o=o, d=d, x_true=x_true, x0=x0, Ns=Ns, i0=i0, tmax=w["tmax"], rng=rng, na=na, P=nt * nd)

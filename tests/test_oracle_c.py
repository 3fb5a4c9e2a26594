"""The C/OpenMP restatement (CPU baseline) against the numpy oracle, which is pinned to the
reference's golden vectors."""
import numpy as np

from oracle import oracle as O
from oracle import oracle_c as OC
from ionotomo_amd import synthetic as syn


def test_c_forward_and_adjoint_match_numpy_oracle(golden):
    g = golden("forward_tec")
    w = syn.make_workload("cfg1")
    ne = O.ne_from_log_model(w["m"], w["K_ne"])
    for Ns in (65, 64, 33):
        rays = O.straight_rays(w["origins"], w["directions"], w["tmax"], Ns)
        ref = O.forward_tec(rays, w["xvec"], w["yvec"], w["zvec"], ne, O.QUAD_SIMPSON_AVG)
        for nt in (1, 4):
            tec = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], ne, w["origins"], w["directions"], w["tmax"], Ns, nt)
            assert np.max(np.abs(tec - ref)) < 1e-13 * np.max(np.abs(ref))
        if Ns == 65:
            assert np.max(np.abs(tec - g["tec65"])) < 1e-13 * np.max(np.abs(g["tec65"]))     # the reference itself
    y = np.random.default_rng(0).normal(size=(8, 1, 8))
    rays = O.straight_rays(w["origins"], w["directions"], w["tmax"], 33)
    grad = OC.adjoint_straight(w["xvec"], w["yvec"], w["zvec"], w["origins"], w["directions"], y, w["tmax"], 33)
    ref = O.adjoint_tec(rays, w["xvec"], w["yvec"], w["zvec"], y)
    assert np.max(np.abs(grad - ref)) < 1e-12 * np.max(np.abs(ref))

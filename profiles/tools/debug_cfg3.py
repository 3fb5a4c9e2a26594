"""Debug: cfg3 tricubic bending tracer vs oracle, per lane mapping."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ionotomo_amd import _lib, synthetic as syn
from oracle import oracle as O
w = syn.make_workload("cfg2", margin_cells=16)
o, d = w["origins"].reshape(-1, 3), w["directions"].reshape(-1, 3)
idx = np.sort(np.random.default_rng(0).choice(len(o), 208, replace=False))
nM = O.ne_to_n(w["ne"], 120e6)
ref = O.fermat_trace(o[idx], d[idx], w["tmax"], w["Ns"], O.n_field_tricubic(w["xvec"], w["yvec"], w["zvec"], nM), bend=True, substeps=4)
for env in ({}, {"IONOTOMO_VARIANT": "3"}, {"IONOTOMO_FERMAT_COOP_RPW": "1"}, {"IONOTOMO_FERMAT_COOP_RPW": "4"}):
    for k in ("IONOTOMO_VARIANT", "IONOTOMO_FERMAT_COOP_RPW"):
        os.environ.pop(k, None)
    os.environ.update(env)
    c = _lib.Context(0)
    c.set_grid(w["xvec"], w["yvec"], w["zvec"], w["ne"])
    for sel, name in ((slice(None), "all"), (idx, "subset")):
        rays = c.trace_fermat(o[sel], d[sel], w["tmax"], w["Ns"], 120e6, bend=True, kind="cubic", substeps=4)
        got = rays[idx] if name == "all" else rays
        err = np.abs(got - ref).max(axis=(1, 2))
        wr = int(np.argmax(err))
        print(env, name, "max err %.3e median %.3e worst ray %d" % (err.max(), np.median(err), wr),
              "profile", np.abs(got[wr, 0] - ref[wr, 0])[::16])
    c.close()

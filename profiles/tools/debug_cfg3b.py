import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ionotomo_amd import _lib, synthetic as syn
w = syn.make_workload("cfg2", margin_cells=16)
o, d = w["origins"].reshape(-1, 3)[::13], w["directions"].reshape(-1, 3)[::13]
X, Y, Z = np.meshgrid(w["xvec"], w["yvec"], w["zvec"], indexing="ij")
smooth = 1e12 * np.exp(-((Z - 300) / 120.0) ** 2) * (1 + 0.5 * np.exp(-((X - 5) ** 2 + (Y + 3) ** 2) / 30.0 ** 2))
os.environ["IONOTOMO_VARIANT"] = "3"
cg = _lib.Context(0)
del os.environ["IONOTOMO_VARIANT"]
cc = _lib.Context(0)
for name, ne in (("turbulent", w["ne"]), ("smooth", smooth), ("const", np.full_like(smooth, 3e11)), ("linz", 1e9 * (Z + 100))):
    for c in (cg, cc):
        c.set_grid(w["xvec"], w["yvec"], w["zvec"], ne)
    for bend in (False, True):
        for sub in (1, 4):
            a = cg.trace_fermat(o, d, w["tmax"], w["Ns"], 120e6, bend=bend, kind="cubic", substeps=sub)
            b = cc.trace_fermat(o, d, w["tmax"], w["Ns"], 120e6, bend=bend, kind="cubic", substeps=sub)
            e = np.abs(a - b)
            print(name, "bend", bend, "sub", sub, "max |coop - generic| x,y,z,s:", e[:, 0].max(), e[:, 1].max(), e[:, 2].max(), e[:, 3].max())

import sys; sys.path.insert(0,'.')
import numpy as np, torch
from ionotomo_amd import synthetic as syn
from ionotomo_amd.engine import RayEngine
w = syn.make_workload("cfg2")
eng = RayEngine(0, interp="linear")
eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
eng.set_values(eng.tensor(w["ne"]))
o, d = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
R=o.shape[0]
for kind in ("cubic","linear"):
    rays = eng.trace_fermat(o, d, w["tmax"], w["Ns"], 120e6, bend=True, kind=kind, substeps=4)
    print(kind, 'trace oob', eng.check_oob(), 'nan', bool(torch.isnan(rays).any()))
    r=rays.cpu().numpy()
    print(' x', r[:,0].min(), r[:,0].max(), w["xvec"][[0,-1]], ' y', r[:,1].min(), r[:,1].max(), w["yvec"][[0,-1]], ' z', r[:,2].min(), r[:,2].max(), w["zvec"][[0,-1]])
    t = eng.forward_rays(rays); print(' fwd_rays oob', eng.check_oob())
t=eng.forward(o,d,w["tmax"],w["Ns"]); print('fwd oob', eng.check_oob())

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from ionotomo_amd import parallel, solvers
from oracle import oracle as Or, solvers as OS
from problems import small_problem
from test_gpu_engine import make_engine
pb = small_problem(na=6, nd=6, nt=4, n=18, Ns=19)
w = pb["w"]
rays = Or.straight_rays(pb["o"], pb["d"], pb["tmax"], pb["Ns"])
G, A = OS.dense_operator(rays, w["xvec"], w["yvec"], w["zvec"], pb["i0"])
d = A @ pb["x_true"].ravel() + pb["rng"].normal(size=A.shape[0]) * 1e-3
cd = np.full(A.shape[0], 1e-6)
for det in (False, True):
    eng = make_engine(w)
    eng.set_deterministic(det)
    prob = parallel.ShardedRays(eng, pb["o"], pb["d"], pb["tmax"], pb["Ns"], dobs=d.reshape(pb["na"], pb["P"]), cdct=cd.reshape(pb["na"], pb["P"]), i0=pb["i0"])
    x0 = eng.tensor(pb["x0"])
    for n_iter in (10, 20, 30, 50):
        xc, hc = solvers.cgls(prob, x0, n_iter=n_iter)
        xr, hr = OS.cgls(A, d, cd, pb["x0"].ravel(), n_iter)
        print(det, n_iter, "x err", np.max(np.abs(xc.cpu().numpy().ravel() - xr)) / np.max(np.abs(xr)), "hist err", np.max(np.abs(np.array(hc) - np.array(hr))) / hr[0])

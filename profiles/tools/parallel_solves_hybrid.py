#!/usr/bin/env python3
"""Stacked single-time-step solves (parallel_solves.py): the forward against the forced per-bundle threshold IONOTOMO_HYBRID_MIN
(0 = the plan's own choice), to check the plan's cost model on this geometry.  python profiles/tools/parallel_solves_hybrid.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from ionotomo_amd import synthetic as syn
from ionotomo_amd.inversion.parallel_solves import StackedSolves

n, tmax, Ns = 128, 1000.0, 129
ants = syn.lofar_enu_km()
Bs = [8, 16, 24, 32]
dirs = syn.rotate_about_pole(syn.facet_directions(42, 4.0, 1), max(Bs))
o_all, d_all = syn.ray_bundle(ants, dirs)
grid = syn.domain_for(o_all, d_all, n, tmax, 4)
ne0 = syn.ne_model(*grid, seed=7, corr=30.0) / 1e11
out = []
for B in Bs:
    for hmin in (0, 65, 32, 24, 16, 8, 1):
        if hmin:
            os.environ["IONOTOMO_HYBRID_MIN"] = str(hmin)
        else:
            os.environ.pop("IONOTOMO_HYBRID_MIN", None)
        st = StackedSolves(tuple(grid), count=B)
        o, d = st.rays([o_all[:, b] for b in range(B)], [d_all[:, b] for b in range(B)], tmax)
        eng = st.engine
        eng.set_values(st.stack_grids([torch.as_tensor(ne0)] * B).reshape(-1))
        ot, dt = eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3))
        tec = torch.empty(ot.shape[0], dtype=torch.float64, device=eng.device)
        eng.plan_forward(ot, dt, tmax, Ns)
        sp = eng.forward_plan_split()
        order = None if sp["bundles_served"] else eng.coherent_order(ot, dt)
        fn = lambda: eng.forward(ot, dt, tmax, Ns, out=tec, order=order)
        bench.SETTLE_MS = 30.0
        ks = sorted(bench.time_steps(fn, 50, 3, torch, None, 1)[1] for _ in range(3))
        rec = {"B": B, "hybrid_min": hmin, "us": ks[1] * 1e6, "served": sp["bundles_served"], "tail": sp["rays_tail"], "T": sp["min_rays_per_served_bundle"],
               "model": sp["model_us"]}
        out.append(rec)
        print(json.dumps(rec), flush=True)
        del st, eng
os.environ.pop("IONOTOMO_HYBRID_MIN", None)

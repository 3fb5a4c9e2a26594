#!/usr/bin/env python3
"""Pipeline-sized forward launches (62 x 42 x Nt rays) against the number of workgroups per CU the lanes = samples kernel is given
(IONOTOMO_BLOCKS_PER_CU): is a small launch bound by its waves' chains (fewer rays per wave = faster) or by starting workgroups?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from ionotomo_amd import synthetic as syn
w = bench.build_workload(0)
out = []
for nt in (1, 4, 8):
    o, d = syn.ray_bundle(syn.lofar_enu_km(), syn.rotate_about_pole(syn.facet_directions(bench.ND, 4.0, 1), nt))
    for ns in (257, 129, 65):
        for bpc in (0, 1, 2, 3, 4, 6, 8):
            e = bench.engine_with_env({"IONOTOMO_BLOCKS_PER_CU": bpc, "IONOTOMO_HYBRID_MIN": 65}, 0)
            e.set_grid(w["xvec"], w["yvec"], w["zvec"])
            e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
            ot, dt = e.tensor(o.reshape(-1, 3)), e.tensor(d.reshape(-1, 3))
            t = torch.empty(ot.shape[0], dtype=torch.float64, device=e.device)
            fn = e.forward_launcher(ot, dt, bench.TMAX, ns, t)
            bench.SETTLE_MS = 20.0
            ks = [bench.time_steps(fn, 100, 5, torch, None, 1)[1] for _ in range(3)]
            out.append({"Nt": nt, "rays": int(ot.shape[0]), "Ns": ns, "blocks_per_cu": bpc, "us": sorted(ks)[1] * 1e6})
            print(json.dumps(out[-1]), file=sys.stderr, flush=True)
            del e
print(json.dumps(out))

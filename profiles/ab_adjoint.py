#!/usr/bin/env python3
"""Interleaved timing of adjoint variants in one process.
    python profiles/ab_adjoint.py "ORDER=1" "ORDER=0" "ORDER=1,ACCUM=f32" "VARIANT=2,ORDER=0" ...
Under rocprofv3 --pmc pass REPS=1 so that dispatch k of k_adjoint_* corresponds to spec k."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ionotomo_amd.engine import RayEngine  # noqa: E402

specs = [a for a in sys.argv[1:] if not a.startswith("REPS=")] or ["ORDER=1", "ORDER=0"]
reps = int(([a for a in sys.argv[1:] if a.startswith("REPS=")] or ["REPS=5"])[0][5:])
w = bench.build_workload(0)
R = w["origins"].shape[0]
engines = []
for spec in specs:
    kv = dict(x.split("=") for x in spec.split(",") if x)
    for k in list(os.environ):
        if k.startswith("IONOTOMO_"):
            del os.environ[k]
    use_order = kv.pop("ORDER", "0") == "1"
    accum = torch.float32 if kv.pop("ACCUM", "f64") == "f32" else torch.float64
    for k, v in kv.items():
        os.environ["IONOTOMO_" + k] = v
    e = RayEngine(0)
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
    engines.append((spec, e, use_order, accum))
e0 = engines[0][1]
o_t, d_t = e0.tensor(w["origins"]), e0.tensor(w["directions"])
order = e0.locality_order(o_t, d_t, bench.TMAX)
y = e0.tensor(np.random.default_rng(0).normal(size=R))
times = {s: [] for s, _, _, _ in engines}
ref = None
for rnd in range(reps):
    for spec, e, uo, accum in engines:
        g = torch.zeros(e.shape, dtype=accum, device="cuda")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        e.adjoint(o_t, d_t, y, bench.TMAX, bench.NS, out=g, order=order if uo else None)
        b.record()
        torch.cuda.synchronize()
        times[spec].append(a.elapsed_time(b))
        if ref is None:
            ref = g.double().clone()
        else:
            err = float((g.double() - ref).abs().max() / ref.abs().max())
            assert "WALK" in spec or err < (1e-4 if accum == torch.float32 else 1e-10), (spec, err)
for spec in times:
    t = np.array(times[spec])
    print("%-40s median %.4f ms  min %.4f ms" % (spec, np.median(t), t.min()), flush=True)

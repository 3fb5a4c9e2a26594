#!/usr/bin/env python3
"""Interleaved A/B timing of forward-kernel variants in ONE process (guide rule 24).
    python profiles/ab_forward.py "VARIANT=0" "VARIANT=1" "VARIANT=0,BLOCKS_PER_CU=4" "PLAN=1" ...
ORDER=0..3: walk order handed to the kernel (none / locality / antenna-direction-time / coherent); PLAN=1: bundle plan.
Each spec sets IONOTOMO_<KEY> env vars before creating its own context/engine."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ionotomo_amd.engine import RayEngine  # noqa: E402

specs = sys.argv[1:] or ["VARIANT=0", "VARIANT=1"]
w = bench.build_workload(0)
R = w["origins"].shape[0]
engines = []
for spec in specs:
    kv = dict(x.split("=") for x in spec.split(",") if x)
    for k in list(os.environ):
        if k.startswith("IONOTOMO_"):
            del os.environ[k]
    storage = kv.pop("STORAGE", "f64")
    use_order = int(kv.pop("ORDER", "0"))
    use_plan = int(kv.pop("PLAN", "0"))
    for k, v in kv.items():
        os.environ["IONOTOMO_" + k] = v
    e = RayEngine(0, storage=storage)
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
    engines.append((spec, e, use_order, use_plan))
o_t, d_t = engines[0][1].tensor(w["origins"]), engines[0][1].tensor(w["directions"])
for spec, e, _, use_plan in engines:
    if use_plan:
        import time
        t0 = time.perf_counter()
        info = e.plan_forward(o_t, d_t, bench.TMAX, bench.NS)
        print("%s: forward plan %s built in %.1f ms" % (spec, info, (time.perf_counter() - t0) * 1e3), flush=True)
engines = [x[:3] for x in engines]
order1 = engines[0][1].locality_order(o_t, d_t, bench.TMAX)
# ORDER=2: (antenna, direction, time) -- consecutive rays are the same line of sight 8 s apart
idx = torch.arange(R, device="cuda").reshape(bench.NA, bench.NT, bench.ND)
order2 = idx.permute(0, 2, 1).reshape(-1).to(torch.int32).contiguous()
orders = {0: None, 1: order1, 2: order2, 3: RayEngine.coherent_order(o_t, d_t)}     # 3: generic (antenna, direction-Morton)
out = torch.empty(R, dtype=torch.float64, device="cuda")
times = {s: [] for s, _, _ in engines}
ref = None
for rnd in range(7):
    for spec, e, uo in engines:
        for _ in range(2):
            e.forward(o_t, d_t, bench.TMAX, bench.NS, out=out, order=orders[uo])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(10):
            e.forward(o_t, d_t, bench.TMAX, bench.NS, out=out, order=orders[uo])
        b.record()
        torch.cuda.synchronize()
        times[spec].append(a.elapsed_time(b) / 10)
        if ref is None:
            ref = out.clone()
        elif "f32" not in spec and not os.environ.get("IONO_AB_NOCHECK"):
            assert float((out - ref).abs().max() / ref.abs().max()) < 1e-12, spec
for spec in times:
    t = np.array(times[spec])
    print("%-40s median %.4f ms  min %.4f ms  -> %.3e rays/s" % (spec, np.median(t), t.min(), R / np.median(t) * 1e3), flush=True)

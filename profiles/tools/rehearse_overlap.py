#!/usr/bin/env python3
"""2 ranks on ONE GPU over gloo: the multi-GPU SIRT / CGLS iteration with the exchange of the back-projected update run back to back
(exchange="compact") and hidden behind the back-projection slab by slab (exchange="overlap").  A control-flow rehearsal with a measured
time: gloo carries device tensors through host memory, so the exchange is far slower than RCCL over xGMI will be -- what the numbers show is
whether the overlapped iteration is shorter than the sequential one by (about) the back-projection time it hides.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 profiles/tools/rehearse_overlap.py"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ionotomo_amd import parallel, solvers  # noqa: E402
from ionotomo_amd.engine import RayEngine  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
w = bench.build_workload(0)                     # the same 260 400 rays on both ranks, split by (time, direction) pair
NA = bench.NA
o, d = w["origins"].reshape(NA, -1, 3), w["directions"].reshape(NA, -1, 3)
out = {}
for mode in ("compact", "overlap"):
    eng = RayEngine(0)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    x0 = torch.exp(eng.tensor(w["m"])).mul_(w["K_ne"] / 1e13).reshape(eng.shape)
    P = o.shape[1]
    prob = parallel.ShardedRays(eng, o, d, bench.TMAX, bench.NS, dobs=np.zeros((NA, P)), cdct=np.full((NA, P), 1e-6), i0=0,
                                exchange=mode, reduce_dtype=torch.float32)
    eng.set_values((x0 * 1.1).reshape(-1))
    prob.dobs = prob.forward().clone()
    for name in ("sirt", "cgls"):
        fn = getattr(solvers, name)
        fn(prob, x0, n_iter=2)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        x, hist = fn(prob, x0, n_iter=12)
        torch.cuda.synchronize()
        dist.barrier()
        out["%s_%s_ms_per_iteration" % (name, mode)] = (time.perf_counter() - t0) / 12 * 1e3
        out["%s_%s_last_objective" % (name, mode)] = hist[-1]
    out["%s_overlapped" % mode] = bool(prob.overlapped())
    out["%s_slab_ranges" % mode] = prob.slab_ranges
    # the pieces, by themselves: the whole back-projection and ONE all-reduce of the compact float32 update
    idx = prob.active_index()
    s_full = torch.zeros(eng.shape, dtype=torch.float64, device=eng.device)
    y = torch.randn(prob.R_local, dtype=torch.float64, device=eng.device)
    buf = torch.zeros(idx.numel(), dtype=torch.float32, device=eng.device)
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    for _ in range(5):
        eng.adjoint(prob.origins, prob.dirs, y, bench.TMAX, bench.NS, out=s_full)
    torch.cuda.synchronize(); out["%s_backprojection_ms" % mode] = (time.perf_counter() - t0) / 5 * 1e3
    dist.barrier(); t0 = time.perf_counter()
    for _ in range(3):
        dist.all_reduce(buf)
    torch.cuda.synchronize(); out["%s_allreduce_compact_f32_ms" % mode] = (time.perf_counter() - t0) / 3 * 1e3
    out["compact_f32_bytes"] = int(buf.numel() * 4)
    del prob, eng
if rank == 0:
    out["note"] = "2 gloo ranks sharing one MI355X; rays %d per rank" % (o.shape[0] * o.shape[1] // world)
    print(json.dumps(out))
dist.destroy_process_group()

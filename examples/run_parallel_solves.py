#!/usr/bin/env python3
"""The pipeline's parallel solves on one GPU: every time step is its own tomographic solve (its own rays, its own model:
inversion/inversion_pipeline.py:131-216 of the reference), and `num_parallel_solves` of them run at once -- here as ONE stacked
problem (ionotomo_amd/inversion/parallel_solves.py), SIRT, with each solve's own objective reported.

    python examples/run_parallel_solves.py --solves 32 --iters 30                  # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        examples/run_parallel_solves.py --solves 256                               # 8 GPUs: 32 solves each, no exchange

Synthetic problem per time step: a Chapman ionosphere with Matern turbulence as the a-priori model, the "truth" 5 % denser with
its own turbulence; data = differential TEC of the truth.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ionotomo_amd import parallel, solvers, synthetic as syn  # noqa: E402
from ionotomo_amd.inversion.parallel_solves import StackedSolves, solve_share  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--solves", type=int, default=16, help="time steps = independent solves (all ranks together)")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--grid", type=int, default=96, help="nodes per axis of every solve's grid")
    ap.add_argument("--backend", default=os.environ.get("IONO_BENCH_BACKEND", "nccl"))
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    mine = list(solve_share(args.solves, world, rank))             # this rank's time steps; nothing is exchanged between ranks:
    parallel.INDEPENDENT_RANKS = True                              # ... the rank's stacked problem is whole, not a shard
    n, tmax, Ns = args.grid, 1000.0, args.grid + 1
    ants = syn.lofar_enu_km()
    dirs = syn.rotate_about_pole(syn.facet_directions(42, 4.0, 1), args.solves)            # the field at every time step
    o_all, d_all = syn.ray_bundle(ants, dirs)                                              # [Na, Nt, Nd, 3]
    grid = syn.domain_for(o_all, d_all, n, tmax, 4)
    Na = o_all.shape[0]
    result = {"rank": rank, "solves": mine}
    if mine:
        st = StackedSolves(tuple(grid), count=len(mine), device=local)
        o, d = st.rays([o_all[:, t] for t in mine], [d_all[:, t] for t in mine], tmax)
        eng = st.engine
        prior = [torch.as_tensor(syn.ne_model(*grid, seed=1000 + t, corr=30.0) / 1e11) for t in mine]
        truth = [torch.as_tensor(1.05 * syn.ne_model(*grid, seed=2000 + t, corr=30.0) / 1e11) for t in mine]
        eng.set_values(st.stack_grids(truth).reshape(-1))
        tec = eng.forward(eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3)), tmax, Ns).reshape(Na, -1)
        dobs = (tec - tec[0:1]).cpu().numpy()
        cdct = np.full(dobs.shape, 1e-4)
        prob = parallel.ShardedRays(eng, o, d, tmax, Ns, dobs=dobs, cdct=cdct, i0=0, tune=False)
        x0 = st.stack_grids(prior)

        def objectives(x):                                         # 1/2 sum r^2 / CdCt of every solve
            eng.set_values(x.reshape(-1).contiguous())
            r = prob.forward().reshape(-1) - prob.dobs.reshape(-1)
            return (0.5 * st.per_solve_sum(r * r / prob.cdct.reshape(-1), Na)).cpu().numpy()

        S0 = objectives(x0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, hist = solvers.sirt(prob, x0, n_iter=args.iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        S1 = objectives(x)
        result.update({"rays": int(o.shape[0] * o.shape[1]), "seconds": dt, "us_per_solve_iteration": dt / args.iters / len(mine) * 1e6,
                       "objective_before": S0.tolist(), "objective_after": S1.tolist(),
                       "stacked_objective_history_first_last": [float(hist[0]), float(hist[-1])]})
        assert (S1 < S0).all(), "every solve must have reduced its own objective"
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

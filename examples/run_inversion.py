#!/usr/bin/env python3
"""Sharded tomographic inversion (BASELINE.json configs 4 / 5): rays split over the ranks by
(time, direction) pair, grid replicated, one all-reduce of the back-projected update per iteration.

    python examples/run_inversion.py --size small --solver cgls --iters 20                  # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        examples/run_inversion.py --size cfg5 --solver sirt --iters 50                      # 8 GPUs (RCCL)

Synthetic problem: the a-priori model is a Chapman ionosphere with Matern turbulence; the "truth" adds a
travelling-disturbance-like blob; data = differential TEC of the truth + noise.
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
from ionotomo_amd.engine import RayEngine  # noqa: E402
from ionotomo_amd.ionosphere.covariance import Covariance  # noqa: E402

SIZES = {"small": dict(na=16, nd=12, nt=4, n=64), "cfg2": dict(na=62, nd=42, nt=1, n=128),
         "cfg5": dict(na=62, nd=100, nt=100, n=256)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", choices=sorted(SIZES), default="small")
    ap.add_argument("--solver", choices=["cgls", "sirt", "sd"], default="cgls")
    ap.add_argument("--iters", type=int, default=20)
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
    sz = SIZES[args.size]
    w = syn.make_workload(antennas="lofar", **sz)
    na, P = sz["na"], sz["nt"] * sz["nd"]
    o, d = w["origins"].reshape(na, P, 3), w["directions"].reshape(na, P, 3)
    x0 = w["ne"] / 1e13                                                       # prior, TECU / km
    X, Y, Z = np.meshgrid(w["xvec"], w["yvec"], w["zvec"], indexing="ij")
    x_true = x0 * (1.0 + 0.3 * np.exp(-((X - 5) ** 2 + (Y + 8) ** 2) / 15.0 ** 2 - ((Z - 300) / 80.0) ** 2))
    eng = RayEngine(local)
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    sigma = 1e-3
    prob = parallel.ShardedRays(eng, o, d, w["tmax"], w["Ns"], dobs=np.zeros((na, P)), cdct=np.full((na, P), sigma ** 2), i0=0)
    eng.set_values(eng.tensor(x_true))
    noise = np.random.default_rng(7).normal(size=(na, P)) * sigma                 # same on every rank, sliced
    prob.dobs = prob.forward() + prob.slice(noise)
    x0_t = eng.tensor(x0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    if args.solver == "cgls":
        x, hist = solvers.cgls(prob, x0_t, n_iter=args.iters)
    elif args.solver == "sirt":
        x, hist = solvers.sirt(prob, x0_t, n_iter=args.iters)
    else:
        K = float(np.median(x0))
        cov = Covariance(dx=w["xvec"][1] - w["xvec"][0], dy=w["yvec"][1] - w["yvec"][0], dz=w["zvec"][1] - w["zvec"][0])
        m, hist = solvers.steepest_descent_log_model(prob, eng.tensor(np.log(x0 / K)), K, max_iter=args.iters, covariance=cov)
        x = K * torch.exp(m)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert not eng.check_oob()
    if rank == 0:
        print(json.dumps({"size": args.size, "solver": args.solver, "ranks": world, "rays_total": na * P,
                          "rays_this_rank": prob.R_local, "grid": list(eng.shape), "iterations": len(hist),
                          "seconds": dt, "ms_per_iteration": dt / max(len(hist), 1) * 1e3,
                          "objective_first": hist[0], "objective_last": hist[-1],
                          "chi2_per_datum_last": 2 * hist[-1] / (na * P),
                          "model_rms_change": float((x - x0_t).norm() / x0_t.norm())}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

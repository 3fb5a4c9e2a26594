#!/usr/bin/env python3
"""What a 1-GPU box can say about the RCCL side of the multi-GPU path: a process group of ONE rank on torch's
`nccl` backend (= RCCL on ROCm) and `parallel.FORCE_COLLECTIVES`, so that every collective of the sharded code paths is issued
although a one-rank sum is the identity.  Measured: launch + local cost per collective at the sizes the solvers exchange (14 MB
compact float32, 28 MB compact float64, 128 MiB dense float64), reduce-scatter + all-gather, the cost of `async_op=True`, and a
CGLS / SIRT iteration in every exchange mode against the same iteration without a group.  Under `rocprofv3 --kernel-trace` the
kernel trace shows which RCCL kernels ran and whether they overlapped k_adjoint_binned (profiles/tools/trace_overlap.py).

    python profiles/tools/nccl_1rank.py [--iters 30] > profiles/r06_nccl_1rank.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def timed(fn, torch, n):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    host_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return {"device_us": a.elapsed_time(b) * 1e3 / n, "host_issue_us": host_issue * 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--no-solvers", action="store_true")
    args = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    import torch
    import torch.distributed as dist
    import bench
    from ionotomo_amd import parallel, solvers
    from ionotomo_amd.engine import RayEngine
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
           "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()),
           "env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_"))}, "csrc_sha": bench.csrc_sha()}
    dev = torch.device("cuda", 0)
    coll = {}
    for name, dtype, n in (("compact_f32_14MB", torch.float32, 3_500_000), ("compact_f64_28MB", torch.float64, 3_500_000),
                           ("dense_f64_128MiB", torch.float64, 256 ** 3), ("scalar_f64", torch.float64, 1)):
        buf = torch.ones(n, dtype=dtype, device=dev)
        outb = torch.empty_like(buf)
        rec = {"bytes": buf.numel() * buf.element_size()}
        rec["all_reduce_in_place"] = timed(lambda: dist.all_reduce(buf), torch, args.iters)
        rec["all_reduce_async_wait"] = timed(lambda: dist.all_reduce(buf, async_op=True).wait(), torch, args.iters)
        rec["reduce_scatter"] = timed(lambda: dist.reduce_scatter_tensor(outb, buf), torch, args.iters)
        rec["all_gather"] = timed(lambda: dist.all_gather_into_tensor(outb, buf), torch, args.iters)
        rec["device_copy_same_bytes"] = timed(lambda: outb.copy_(buf), torch, args.iters)
        assert bool((buf == 1).all())                                   # a one-rank sum is the identity
        coll[name] = rec
        del buf, outb
    out["collectives"] = coll
    if not args.no_solvers:
        w = bench.build_workload(0)
        eng = RayEngine(0, storage="f64")
        eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
        m_t = eng.tensor(w["m"])
        x0 = torch.exp(m_t).mul_(w["K_ne"] / 1e13).reshape(eng.shape)
        oo, dd = w["origins"].reshape(bench.NA, -1, 3), w["directions"].reshape(bench.NA, -1, 3)
        P = oo.shape[1]
        res = {}
        ref = {}
        for forced, exchange, rd in ((False, "compact", None), (True, "compact", None), (True, "compact", torch.float32), (True, "dense", None),
                                     (True, "sharded", None), (True, "overlap", None), (True, "overlap", torch.float32)):
            parallel.FORCE_COLLECTIVES = forced
            prob = parallel.ShardedRays(eng, oo, dd, bench.TMAX, bench.NS, dobs=np.zeros((bench.NA, P)), cdct=np.full((bench.NA, P), 1e-6), i0=0,
                                        exchange=exchange, reduce_dtype=rd, tune=False)
            eng.set_values((x0 * 1.1).reshape(-1))
            prob.dobs = prob.forward().clone()
            tag = ("forced_" if forced else "no_group_") + exchange + ("_f32" if rd is not None else "")
            rec = {"multi": bool(prob.multi), "overlapped": bool(prob.overlapped())}
            for name in ("cgls", "sirt"):
                fn = getattr(solvers, name)
                eng.set_deterministic(True)             # fixed-point back-projection: two runs CAN agree bit for bit
                x, hist = fn(prob, x0, n_iter=6)
                eng.set_deterministic(False)
                if not forced:
                    ref[name] = (x.clone(), list(hist))
                elif rd is None:
                    rec[name + "_bit_equal_to_no_group"] = bool(torch.equal(x, ref[name][0])) and list(hist) == ref[name][1]
                else:
                    rec[name + "_max_rel_dev_vs_no_group"] = float(((x - ref[name][0]).abs().max() / ref[name][0].abs().max()))
                fn(prob, x0, n_iter=10)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    fn(prob, x0, n_iter=30)
                torch.cuda.synchronize()
                t30 = (time.perf_counter() - t0) / 3
                t0 = time.perf_counter()
                for _ in range(3):
                    fn(prob, x0, n_iter=10)
                torch.cuda.synchronize()
                t10 = (time.perf_counter() - t0) / 3
                rec[name + "_ms_per_iteration_marginal"] = (t30 - t10) / 20 * 1e3
            res[tag] = rec
            del prob
        parallel.FORCE_COLLECTIVES = False
        out["solver_iterations_bench_shape"] = res
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

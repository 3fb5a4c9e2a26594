#!/usr/bin/env python3
"""bench.py -- ray-integrals/s through a 256^3 ne grid (BASELINE.json metric) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 100 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the forward ray-integral kernel (straight rays generated in-kernel,
trilinear interpolation, Simpson quadrature, float64) over this rank's batch of synthetic rays:
62 LOFAR-HBA stations x 42 directions x 100 timesteps = 260,400 rays, Ns = 257 samples each,
through a 256^3 electron-density grid resident in HBM.  Weak scaling: every rank owns its own
(time, direction) block of rays and a replica of the grid; the forward needs no collective.
Rank 0 prints ONE JSON line.  Extra keys report the adjoint, a full forward+adjoint+all-reduce
iteration, float32 grid storage and the single-timestep (2,604-ray) launch.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NA, ND, NT, NGRID, NS, TMAX = 62, 42, 100, 256, 257, 1000.0
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MAX_RANKS_FOR_DOMAIN = 8


def algorithmic_bytes_per_ray(ns, grid_elem_bytes, corners=8):
    """Ns * C * sizeof(T_grid) + 56 (origin + direction in, TEC out)."""
    return ns * corners * grid_elem_bytes + 56


def build_workload(rank):
    """Per-rank rays + the (rank-independent) grid.  Rank r observes its own set of 42 facet
    directions (seed 1 + r) over 100 timesteps; the grid box is the bounding box of the cones of
    all MAX_RANKS_FOR_DOMAIN possible ranks, so the grid is identical for every N."""
    from ionotomo_amd import synthetic as syn
    ants = syn.lofar_enu_km()
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    mine = None
    for r in range(MAX_RANKS_FOR_DOMAIN):
        dirs = syn.rotate_about_pole(syn.facet_directions(ND, 4.0, 1 + r), NT)      # [Nt,Nd,3]
        if r == rank:
            mine = dirs
        slope = dirs[..., :2] / dirs[..., 2:3]
        for a in (ants[np.argmin(ants[:, 0])], ants[np.argmax(ants[:, 0])], ants[np.argmin(ants[:, 1])],
                  ants[np.argmax(ants[:, 1])], ants[np.argmin(ants[:, 2])], ants[np.argmax(ants[:, 2])]):
            end = a[:2] + slope.reshape(-1, 2) * (TMAX - a[2])
            lo[:2] = np.minimum(lo[:2], np.minimum(end.min(0), ants[:, :2].min(0)))
            hi[:2] = np.maximum(hi[:2], np.maximum(end.max(0), ants[:, :2].max(0)))
    lo[2], hi[2] = ants[:, 2].min(), TMAX
    vecs = []
    for a in range(3):
        span = hi[a] - lo[a]
        pad = span * 4 / (NGRID - 1 - 8) + 1e-3 * span
        vecs.append(np.linspace(lo[a] - pad, hi[a] + pad, NGRID))
    origins, directions = syn.ray_bundle(ants, mine)                                   # [Na,Nt,Nd,3]
    ne = syn.ne_model(vecs[0], vecs[1], vecs[2], seed=1234)
    K_ne = float(np.median(ne))
    return dict(xvec=vecs[0], yvec=vecs[1], zvec=vecs[2], m=np.log(ne / K_ne), K_ne=K_ne,
                origins=origins.reshape(-1, 3), directions=directions.reshape(-1, 3))


def time_steps(fn, steps, warmup, torch, dist, world):
    """W warmups, then EXACTLY K steps bracketed by barrier + synchronize; per-step HIP-event
    times on the launch stream (torch's current stream == the ctx stream).  Returns
    (wall seconds MAX over ranks, mean kernel seconds per step on this rank)."""
    for _ in range(warmup):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)        # MAX over ranks
        wall = float(t.item())
    kern = float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e-3
    return wall, kern


def cpu_baseline(w, tec_gpu):
    """The oracle's C/OpenMP restatement on this box's host cores, same workload (rank 0, N=1):
    reported next to the GPU number, and used as the in-run parity gate."""
    from oracle import oracle as O
    from oracle import oracle_c as OC
    ne = O.ne_from_log_model(w["m"], w["K_ne"])
    R = w["origins"].shape[0]
    threads = OC.num_threads()
    OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], ne, w["origins"][:2604], w["directions"][:2604], TMAX, NS)
    reps, t0, tec = 0, time.perf_counter(), None
    while reps < 6 and (time.perf_counter() - t0 < 8.0 or reps == 0):
        tec = OC.forward_tec_straight(w["xvec"], w["yvec"], w["zvec"], ne, w["origins"], w["directions"], TMAX, NS)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    rel = float(np.max(np.abs(tec_gpu - tec) / np.abs(tec)))
    # single-thread numpy port in the shape of the reference's per-ray loop, on one timestep's worth of rays
    sub = slice(0, 2604 * 4, 4 * 25)
    t1 = time.perf_counter()
    rays = O.straight_rays(w["origins"][sub], w["directions"][sub], TMAX, NS)
    O.forward_tec_loop(rays, w["xvec"], w["yvec"], w["zvec"], ne)
    numpy_rate = rays.shape[0] / (time.perf_counter() - t1)
    return dict(value=R / dt, unit="ray-integrals/s", cores=threads, kind="port",
                sample="full per-GPU batch (%d rays x %d samples, 256^3 f64 grid) x %d repetitions, "
                       "oracle/oracle_c.c with OpenMP on %d threads; numpy per-ray-loop port (1 thread, "
                       "%d rays): %.3g ray-integrals/s" % (R, NS, reps, threads, rays.shape[0], numpy_rate)), rel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-order", dest="order", action="store_false", help="walk rays in [Na][Nt][Nd] memory order")
    ap.add_argument("--main-only", action="store_true",
                    help="only the timed headline launches (clean rocprofv3 --stats averages); implies --no-cpu")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # one process per GPU; IONO_BENCH_BACKEND=gloo + several ranks on one card is only for rehearsing the
    # multi-rank control flow on a 1-GPU box
    backend = os.environ.get("IONO_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from ionotomo_amd.engine import RayEngine
    w = build_workload(rank)
    R = w["origins"].shape[0]
    eng = RayEngine(local, storage="f64")
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    m_t = eng.tensor(w["m"])
    eng.set_log_model(m_t, w["K_ne"] / 1e13)
    o_t, d_t = eng.tensor(w["origins"]), eng.tensor(w["directions"])
    tec_t = torch.empty(R, dtype=torch.float64, device=eng.device)

    # walk order: rays whose paths nearly coincide run back to back (geometry only, computed once,
    # reused by every launch of an inversion; results are independent of it)
    order_t = eng.locality_order(o_t, d_t, TMAX) if args.order else None

    def fwd():
        eng.forward(o_t, d_t, TMAX, NS, out=tec_t)          # the forward gains nothing from the order (measured)

    wall, kern = time_steps(fwd, args.steps, args.warmup, torch, dist, world)
    assert not eng.check_oob(), "rays left the grid"
    value = world * R * args.steps / wall
    bytes_ray = algorithmic_bytes_per_ray(NS, 8)
    achieved = R * bytes_ray / kern / 1e9
    tec_gpu = tec_t.cpu().numpy()

    extra = {}
    if args.main_only:
        if rank == 0:
            print(json.dumps({"metric": "ray-integrals/sec through 256^3 ne grid", "value": value, "n_gpus": world,
                              "steps": args.steps, "kernel_ms": kern * 1e3, "main_only": True}))
        if world > 1:
            dist.destroy_process_group()
        return
    # everything below is reported next to the headline number; a failure there (e.g. in the collective of the
    # iteration leg) must not lose the headline line
    try:
        # ---- adjoint + one full iteration (forward, fused residual adjoint, all-reduce of the update)
        rng = np.random.default_rng(2 + rank)
        dobs_t = eng.tensor(tec_gpu.reshape(NA, -1) - tec_gpu.reshape(NA, -1)[0] + rng.normal(size=(NA, R // NA)) * 1e-3)
        cdct_t = torch.full((R,), 1e-6, dtype=torch.float64, device=eng.device)
        grad_t = torch.zeros(eng.shape, dtype=torch.float64, device=eng.device)

        def adj():
            grad_t.zero_()
            eng.adjoint_residual(o_t, d_t, tec_t, dobs_t, cdct_t, NA, 0, TMAX, NS, out=grad_t, order=order_t)

        def iteration():
            fwd()
            adj()
            if world > 1:
                dist.all_reduce(grad_t)

        k2 = max(3, min(25, args.steps // 4))
        # work partition of the back-projection balanced by measured cost: like the walk order it depends on the ray
        # geometry only, is computed once per inversion and never changes results (engine.tune_adjoint_partition)
        extra["adjoint_partition"] = eng.tune_adjoint_partition(adj, R) if args.order else None
        awall, akern = time_steps(adj, k2, 1, torch, dist, world)
        iwall, _ = time_steps(iteration, k2, 1, torch, dist, world)
        extra["adjoint_ray_integrals_per_s"] = world * R * k2 / awall
        extra["adjoint_ms"] = akern * 1e3
        extra["iteration_ms_fwd_adj_allreduce"] = iwall / k2 * 1e3
        if world > 1:
            # the same iteration with the update exchanged only over the nodes some rank's rays reach
            # (ionotomo_amd/parallel.py:GradientExchange; the plan is built once per geometry)
            from ionotomo_amd.parallel import GradientExchange
            ones = torch.ones(R, dtype=torch.float64, device=eng.device)
            xch = GradientExchange("compact").plan(eng.adjoint(o_t, d_t, ones, TMAX, NS, order=order_t))

            def iteration_compact():
                fwd()
                adj()
                xch.sum_(grad_t)
            cwall, _ = time_steps(iteration_compact, k2, 1, torch, dist, world)
            extra["iteration_ms_compact_exchange"] = cwall / k2 * 1e3
            extra["exchange_active_node_fraction"] = xch.fraction
            xch32 = GradientExchange("compact", reduce_dtype=torch.float32)
            xch32.index, xch32.fraction = xch.index, xch.fraction

            def iteration_compact32():
                fwd()
                adj()
                xch32.sum_(grad_t)
            c32wall, _ = time_steps(iteration_compact32, k2, 1, torch, dist, world)
            extra["iteration_ms_compact_exchange_f32"] = c32wall / k2 * 1e3
        if order_t is not None:
            eng.ctx.walk_partition_set(1, None, R)       # the tuned partition belongs to the ordered walk

            def adj_unordered():
                grad_t.zero_()
                eng.adjoint_residual(o_t, d_t, tec_t, dobs_t, cdct_t, NA, 0, TMAX, NS, out=grad_t)
            wn, kn = time_steps(adj_unordered, k2, 1, torch, dist, world)
            extra["adjoint_unordered_walk_ms"] = kn * 1e3
        # ---- float32 grid storage (float64 arithmetic) and the single-timestep launch
        eng32 = RayEngine(local, storage="f32")
        eng32.set_grid(w["xvec"], w["yvec"], w["zvec"])
        eng32.set_log_model(m_t, w["K_ne"] / 1e13)
        tec32 = torch.empty_like(tec_t)
        w32, k32 = time_steps(lambda: eng32.forward(o_t, d_t, TMAX, NS, out=tec32), k2, 1, torch, dist, world)
        extra["f32_grid_ray_integrals_per_s"] = world * R * k2 / w32
        extra["f32_grid_roofline_frac"] = R * algorithmic_bytes_per_ray(NS, 4) / k32 / 1e9 / HBM_PEAK_GBS
        extra["f32_grid_max_rel_err_vs_f64"] = float((tec32 - tec_t).abs().div(tec_t.abs()).max().item())
        # one timestep's worth of rays, contiguous in [Na][Nt*Nd] order is not one timestep; build it explicitly
        sel = torch.arange(R, device=eng.device).reshape(NA, NT, ND)[:, 0, :].reshape(-1)
        o1, d1 = o_t[sel].contiguous(), d_t[sel].contiguous()
        t1 = torch.empty(o1.shape[0], dtype=torch.float64, device=eng.device)
        _, k1 = time_steps(lambda: eng.forward(o1, d1, TMAX, NS, out=t1), 50, 5, torch, dist, 1)
        extra["single_timestep_rays"] = int(o1.shape[0])
        extra["single_timestep_us"] = k1 * 1e6
    except Exception as exc:                                    # noqa: BLE001
        extra["error"] = "%s: %s" % (type(exc).__name__, exc)

    line = {
        "metric": "ray-integrals/sec through 256^3 ne grid",
        "value": value, "unit": "ray-integrals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "62 ant (LOFAR-HBA) x 42 dir x 100 times per GPU = %d straight rays, Ns=%d, 256^3 f64 ne "
                               "grid, trilinear + Simpson, forward TEC" % (R, NS),
                   "rays_per_gpu": R, "samples_per_ray": NS, "grid": [NGRID] * 3, "interp": "trilinear",
                   "quadrature": "simpson", "sharding": "rays by (time,direction) block, grid replicated"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "k_forward_straight_u<double>", "kernel_ms": kern * 1e3,
                     "algorithmic_bytes_per_ray": bytes_ray,
                     "note": "achieved = ALGORITHMIC bytes (Ns*8 corners*8 B + 56 per ray, no credit for reuse) / kernel "
                             "time; it can exceed the HBM peak because the 128 MiB grid is served from L2 / Infinity "
                             "Cache -- `traffic` is the PMC-measured HBM-side bytes per launch"},
        "extra": extra,
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_forward.json")
    if os.path.exists(pmc):
        try:
            p = json.load(open(pmc))
            if p.get("rays_per_launch") == R and p.get("samples_per_ray") == NS:
                line["roofline"]["traffic"] = p.get("hbm_bytes_per_launch")
                line["roofline"]["traffic_source"] = p.get("source")
        except Exception:
            pass
    if rank == 0 and world == 1 and not args.no_cpu:
        cb, relerr = cpu_baseline(w, tec_gpu)
        line["cpu_baseline"] = cb
        line["parity_max_rel_err_vs_cpu_oracle"] = relerr
        assert relerr < 1e-6, "GPU TEC differs from the CPU oracle by %g" % relerr
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

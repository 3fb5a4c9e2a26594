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
Rank 0 prints ONE JSON line on stdout (the headline part of it also goes to stderr as soon as it
is measured, before any leg that contains a collective).  `roofline` prices the kernel against the
on-chip path its bytes flow through (LDS reads: the 128 MiB grid is cache-resident, so HBM is not
the limiter), with the denominators measured on this box by profiles/tools/cache_peaks, and reports
the HBM-side, L2-side, vector-L1 and compulsory figures next to it (DESIGN.md section 4 gives the
formulas).  Extra keys: the adjoint, a forward + adjoint + all-reduce iteration, CGLS / SIRT
iterations, the tricubic path, float32 storage, the single-timestep launch, and `cfg4`: BASELINE
config 4 (62 x 100 x 100 = 620,000 rays split over the N ranks: STRONG scaling) with the
per-iteration exchange measured by itself.

    python bench.py --only forward|adjoint|cubic_forward|cubic_adjoint|cgls|sirt --steps K
runs ONE leg alone (clean rocprofv3 --stats / --pmc averages) and prints a short JSON line.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NA, ND, NT, NGRID, NS, TMAX = 62, 42, 100, 256, 257, 1000.0
# /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 achievable); L2 ~34.5 TB/s aggregate; LDS 256 B/clk/CU for
# ds_read_b64 / b128; vector L1 64 B/clk/CU (not in the guide: measured by profiles/tools/cache_peaks, 37.9 TB/s = 61.7 B/clk/CU);
# memory-side float atomics ~1.3 TB/s of added bytes.  Every on-chip denominator is ALSO measured on the box in the same run.
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0
VL1D_NOMINAL_GBS = 64 * 256 * 2.4
LDS_B64_NOMINAL_GBS = 256 * 256 * 2.4
LDS_GUIDE_GBS = 150000.0       # the guide's LDS section: "Aggregate with every CU streaming (~2.4 GHz): ~150 TB/s for ds_read_b64/b128"
ATOMIC_PEAK_GBS = 1300.0
MAX_RANKS_FOR_DOMAIN = 8
WATCHDOG_EXIT_CODE = 3      # every rank's exit code after the watchdog printed the headline without the later legs
PMC_JSON = os.path.join(ROOT, "profiles", "pmc_counters.json")
PEAKS_TOOL = os.path.join(ROOT, "profiles", "tools", "cache_peaks")


def algorithmic_bytes_per_ray(ns, grid_elem_bytes, corners=8):
    """Ns * C * sizeof(T_grid) + 56 (origin + direction in, TEC out)."""
    return ns * corners * grid_elem_bytes + 56


def csrc_sha():
    """Hash of everything the kernels are built from: committed PMC counters are used only while it matches."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ionotomo_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/ionotomo_hip.h"]:
        p = os.path.normpath(os.path.join(d, f))
        if os.path.isfile(p) and p.endswith((".h", ".hip")):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


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


def build_cfg4(w):
    """BASELINE config 4: 62 antennas x 100 directions x 100 timesteps = 620,000 rays (the FULL problem, identical on every
    rank: ShardedRays takes this rank's (time, direction) block) on a 256^3 grid of their own bounding box.  The node values
    are the headline workload's (a synthetic ionosphere either way; parity at this size: tests/test_gpu_configs.py)."""
    from ionotomo_amd import synthetic as syn
    ants = syn.lofar_enu_km()
    dirs = syn.rotate_about_pole(syn.facet_directions(100, 4.0, 1), 100)
    origins, directions = syn.ray_bundle(ants, dirs)                                   # [Na,Nt,Nd,3]
    xv, yv, zv = syn.domain_for(origins.reshape(-1, 3), directions.reshape(-1, 3), NGRID, TMAX)
    return dict(xvec=xv, yvec=yv, zvec=zv, m=w["m"], K_ne=w["K_ne"], origins=origins.reshape(NA, -1, 3),
                directions=directions.reshape(NA, -1, 3))


SETTLE_MS = 0.0             # set from --settle-ms in main()


FORCE_GROUP = False          # IONO_BENCH_FORCE_GROUP=1: a process group of ONE rank (torch `nccl` = RCCL) and every collective of the
                             # multi-rank control flow issued on it -- the rehearsal a 1-GPU box allows (profiles/r06_bench_nccl_1rank_group.json)


def grouped(world):
    return world > 1 or FORCE_GROUP


def settle(fn, torch, dist, world, ms):
    """Untimed launches of the SAME leg for about `ms` milliseconds before its W warmups.  Why: after any idle stretch this device
    takes tens of milliseconds of continuous load to reach its sustained state -- the per-launch device time of the headline kernel
    reads 106-109 us for the first ten launches, 115-131 us between the 20th and the 50th, and 96-97 us from the ~400th on
    (profiles/r05_clock_ramp.json: rocprofv3 kernel trace of 1 500 back-to-back launches) -- and an inversion runs its iterations
    back to back for seconds.  Nothing is skipped or shortened by this: the W warmups and the K timed steps follow as the contract
    says; `extra.headline_cold_window` keeps the number of the first K launches after idle next to it."""
    if ms <= 0:
        return 0
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    n = int(min(20000, max(1, ms * 1e-3 / max(time.perf_counter() - t0, 2e-5))))
    if grouped(world):                   # a leg may hold a collective: every rank runs the same number of launches
        t = torch.tensor([n], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n = int(t.item())
    for _ in range(n):
        fn()
    return n


def time_steps(fn, steps, warmup, torch, dist, world, settle_ms=None):
    """[settle, see above;] W warmups, then EXACTLY K steps bracketed by barrier + synchronize; one pair of HIP events on the
    launch stream (torch's current stream == the ctx stream) around the K launches: device time per step with nothing between the
    launches but the launches themselves.  Returns (wall seconds MAX over ranks, mean device seconds per step on this rank)."""
    settle(fn, torch, dist, world, SETTLE_MS if settle_ms is None else settle_ms)
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if grouped(world):
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    if grouped(world):
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if grouped(world):
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)        # MAX over ranks
        wall = float(t.item())
    kern = a.elapsed_time(b) * 1e-3 / steps
    return wall, kern


MIN_TIMED_MS = 20.0          # the headline's timed region is at least this long: K steps are repeated as M windows (see time_windows)


def time_windows(fn, steps, warmup, torch, dist, world, min_ms=MIN_TIMED_MS, max_windows=50):
    """The headline measurement.  [settle;] W warmups; then the contract's window -- EXACTLY K steps between barrier + synchronize, one
    pair of HIP events on the launch stream around them -- M times back to back, M chosen from the first window so that the timed
    region lasts >= ``min_ms`` (K = 20 steps of 0.1 ms are a 2 ms window that one disturbed box moved by 17 %: VERDICT r5).
    Returns (total wall seconds of the M windows, MAX over ranks per window; mean device seconds per step over all windows; M;
    per-window wall ms per step)."""
    settle(fn, torch, dist, world, SETTLE_MS)
    for _ in range(warmup):
        fn()
    walls, kerns, m = [], [], 1
    while len(walls) < m:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if grouped(world):
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.record()
        for _ in range(steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        if grouped(world):
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if grouped(world):
            t = torch.tensor([wall], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)        # MAX over ranks
            wall = float(t.item())
        walls.append(wall)
        kerns.append(a.elapsed_time(b) * 1e-3 / steps)
        if len(walls) == 1:                                  # (the all-reduced wall: every rank computes the same M)
            m = int(min(max_windows, max(1, -(-min_ms * 1e-3 // wall))))
    return sum(walls), sum(kerns) / len(kerns), m, [w_ / steps * 1e3 for w_ in walls]


def cpu_baseline(w, tec_gpu):
    """The oracle's C/OpenMP restatement on this box's host cores (rank 0, N=1): reported next to the GPU number
    and used as the in-run parity gate.  It is a straightforward, UNOPTIMISED port (binary search + three divisions
    per sample, weights recomputed per ray), built on this box (``-march=native`` here, not in the build container)."""
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
    # ... and the vectorised form of the phase observable (iterative_newton.py:86-127: one gather over all samples, Nf = 2)
    o6, d6 = (w[k].reshape(NA, NT, ND, 3)[:, :1].reshape(-1, 3) for k in ("origins", "directions"))     # one timestep
    rays6 = O.straight_rays(o6, d6, TMAX, NS).reshape(NA, 1, ND, 4, NS)
    t2 = time.perf_counter()
    O.forward_phase(w["m"] + np.log(w["K_ne"] / 1e11), np.zeros((NA, 1)), np.zeros(NA), w["xvec"], w["yvec"], w["zvec"], rays6,
                    np.array([120e6, 150e6]), K=1e11, i0=0)
    phase_rate = rays6.shape[0] * rays6.shape[2] / (time.perf_counter() - t2)
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        cpu_model = "unknown"
    return dict(value=R / dt, unit="ray-integrals/s", cores=threads, kind="port", optimised=False,
                note="context only: a straightforward port of the reference's arithmetic (binary search + three divisions per sample, "
                     "weights recomputed per ray), NOT a tuned CPU implementation -- the GPU / CPU ratio says nothing about kernel quality",
                host_cpus=os.cpu_count(), cpu_model=cpu_model,
                sample="full per-GPU batch (%d rays x %d samples, 256^3 f64 grid) x %d repetitions, unoptimised C/OpenMP "
                       "port oracle/oracle_c.c on %d threads; numpy per-ray-loop port (1 thread, %d rays): %.3g "
                       "ray-integrals/s; numpy vectorised phase port (1 process, %d rays x 2 frequencies): %.3g ray-integrals/s"
                       % (R, NS, reps, threads, rays.shape[0], numpy_rate, rays6.shape[0] * rays6.shape[2], phase_rate)), rel


def load_pmc(sha):
    """Committed per-launch PMC counters (profiles/summarize.py) -- only if they were taken on THESE kernel sources."""
    try:
        p = json.load(open(PMC_JSON))
    except Exception:
        return None, "no profiles/pmc_counters.json"
    if p.get("csrc_sha") != sha:
        return None, "profiles/pmc_counters.json was taken on csrc %s, this build is %s: stale, dropped" % (p.get("csrc_sha"), sha)
    return p, p.get("source", "")


def measured_peaks():
    """On-chip denominators measured on THIS box, in this run (a child process, after the timed region): vector-L1 / L2 /
    Infinity-Cache / HBM rates of 16-B wave-loads by access shape, LDS rates by read form (profiles/tools/cache_peaks.hip,
    compiled by __graft_entry__.build)."""
    if not os.path.exists(PEAKS_TOOL):
        return None
    try:
        out = subprocess.run([PEAKS_TOOL], capture_output=True, timeout=120, text=True).stdout
        return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    except Exception as exc:                                    # noqa: BLE001
        return {"error": "%s: %s" % (type(exc).__name__, exc)}


def fabric_bytes(c):
    """L2 fabric-side read bytes from the request-size counters (exact), else FETCH_SIZE (KiB; counts 64 B per
    request whatever its size -- MI355X_MICROARCH.md, HBM section) as a lower bound."""
    if all(k in c for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_128B_sum")):
        n32, n128 = c["TCC_EA0_RDREQ_32B_sum"], c["TCC_EA0_RDREQ_128B_sum"]
        n64 = c.get("TCC_EA0_RDREQ_64B_sum", c["TCC_EA0_RDREQ_sum"] - n32 - n128)
        return 32.0 * n32 + 64.0 * n64 + 128.0 * n128, "sum over request sizes 32/64/128 B of TCC_EA0_RDREQ_*"
    if "FETCH_SIZE" in c:
        return 1024.0 * c["FETCH_SIZE"], "FETCH_SIZE (64 B per request: lower bound)"
    return None, None


def forward_roofline(R, kern, pmc, pmc_note, copy_gbs, grid_bytes, peaks, planned):
    """Where the algorithmic bytes (Ns x 8 corners x 8 B + 56 per ray, no credit for reuse) flow, and how busy that
    path is.  Planned (bundle-stationary, k_forward_bundle): every corner value a lane uses is read from LDS (eight ds_read_b64
    per sample), so the LDS read path carries the algorithmic volume; the vector L1 / L2 only carry the window copies
    (counters).  Unplanned (k_forward_straight_u): every corner value crosses the vector L1."""
    bytes_ray = algorithmic_bytes_per_ray(NS, 8)
    achieved = R * bytes_ray / kern / 1e9
    compulsory = (grid_bytes + R * 56) / kern / 1e9
    pk = peaks if isinstance(peaks, dict) and "lds" in peaks else {}
    lds_meas = pk.get("lds", {}).get("read_b64_gbs")
    vl1d_meas = pk.get("vl1d", {}).get("dense_gbs")
    l2_meas = pk.get("l2", {}).get("dense_gbs")
    if planned:
        bound, kernel, nominal, meas = "lds", "k_forward_bundle", LDS_B64_NOMINAL_GBS, lds_meas
        note = ("Bundle-stationary forward: the voxel neighbourhood of <= 64 nearly coincident rays is copied to LDS once per 8 "
                "samples (LDS-DMA) and every corner value a lane asks for is an LDS read (eight ds_read_b64 per sample), so the "
                "ALGORITHMIC bytes (Ns x 8 corners x 8 B + 56 per ray, no credit for reuse) are the load of the LDS read path and "
                "`frac` = `frac_vs_guide` is its utilisation against the guide's ~150 TB/s for ds_read_b64 (`frac_vs_measured`: against "
                "the rate profiles/tools/cache_peaks reached on this box in this run; `peak_nominal`: 256 B/clk/CU x 2.4 GHz).  The "
                "counters next to it say how busy each unit was (`units`): float64 vector-instruction issue first, the LDS "
                "second.  The 128 MiB grid is L2 / Infinity-Cache resident: the HBM side carries far less "
                "(`hbm.counter_gbs`: L2 fabric-side requests, Infinity-Cache hits included; `hbm.compulsory_gbs`: grid + ray I/O "
                "once per launch), which is why the algorithmic rate exceeds the HBM peak (`hbm.algorithmic_over_peak`) -- HBM is "
                "not the roof of this kernel.")
    else:
        bound, kernel, nominal, meas = "vl1d", "k_forward_straight_u<double>", VL1D_NOMINAL_GBS, vl1d_meas
        note = ("lanes = samples kernel: every corner value a lane asks for crosses the per-CU vector L1 / texture-address path, so "
                "the algorithmic bytes are that path's load; `peak` is its dense 16-B wave-load rate measured on this box.")
    # `frac` is ALWAYS against the guide's figure for the path (it never depends on whether the peaks tool was built or ran);
    # `frac_vs_measured` is against the rate profiles/tools/cache_peaks reached on THIS box in this run (None without it)
    guide = LDS_GUIDE_GBS if planned else nominal
    rl = {"bound": bound, "achieved": achieved, "peak": guide, "unit": "GB/s", "frac": achieved / guide,
          "peak_source": ("MI355X_MICROARCH.md, LDS: ~150 TB/s aggregate for ds_read_b64/b128" if planned else
                          "64 B/clk/CU x 256 CUs x 2.4 GHz (vector L1 return path)"),
          "frac_vs_guide": achieved / guide, "frac_vs_measured": (achieved / meas) if meas else None,
          "valu_busy": None, "issue_frac": None,
          "peak_nominal": nominal, "peak_measured": meas, "frac_of_nominal": achieved / nominal,
          "traffic": None, "kernel": kernel, "kernel_ms": kern * 1e3,
          "algorithmic_bytes_per_ray": bytes_ray, "algorithmic_gbs": achieved,
          "hbm": {"peak": HBM_PEAK_GBS, "algorithmic_over_peak": achieved / HBM_PEAK_GBS, "compulsory_gbs": compulsory,
                  "compulsory_frac": compulsory / HBM_PEAK_GBS, "copy_gbs_measured": copy_gbs,
                  "dense_read_gbs_measured": pk.get("hbm", {}).get("dense_gbs"),
                  "counter_gbs": None, "counter_frac": None},
          "l2": {"peak": L2_PEAK_GBS, "peak_measured": l2_meas, "request_gbs": None, "frac": None},
          "vl1d": {"peak_nominal": VL1D_NOMINAL_GBS, "peak_measured": vl1d_meas,
                   "ns_per_16B_wave_load_per_cu": {k: pk.get("vl1d", {}).get(k)
                                                   for k in ("dense_ns_per_wave_load_per_cu", "same_ns", "l4_ns", "l20_ns")}},
          "lds": {"peak_nominal_b64": LDS_B64_NOMINAL_GBS, "measured": pk.get("lds")},
          "pmc": pmc_note, "note": note}
    c = (pmc or {}).get("forward")
    if c and c.get("rays") == R and c.get("Ns") == NS and c.get("kernel", "").startswith(kernel.split("<")[0]):
        fb, how = fabric_bytes(c)
        if fb is not None:
            fb += 1024.0 * c.get("WRITE_SIZE", 0.0)
            rl["traffic"] = fb
            rl["traffic_source"] = how + " + WRITE_SIZE, per launch (L2 fabric side: includes Infinity-Cache hits)"
            rl["hbm"]["counter_gbs"] = fb / kern / 1e9
            rl["hbm"]["counter_frac"] = fb / kern / 1e9 / HBM_PEAK_GBS
            rl["hbm"]["traffic_over_compulsory"] = fb / (grid_bytes + R * 56)
        if "TCP_TCC_READ_REQ_sum" in c:
            req = 128.0 * c["TCP_TCC_READ_REQ_sum"] / kern / 1e9          # L1 -> L2 read requests are 128-B lines
            rl["l2"]["request_gbs"], rl["l2"]["frac"] = req, req / (l2_meas or L2_PEAK_GBS)
            rl["l2"]["request_bytes_per_launch"] = 128.0 * c["TCP_TCC_READ_REQ_sum"]
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
            rl["vl1d"]["tag_lookups_per_launch"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
            rl["vl1d"]["hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
        if "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0                      # per-XCD active cycles of the launch (profiled run)
            units = {"profiled_clock_ghz": cyc / (kern * 1e9)}
            if "SQ_ACTIVE_INST_VALU" in c:                       # quad-cycles summed over the chip; 1 024 SIMDs
                units["valu_busy_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc)
                # the unit that actually leads this kernel: float64 vector-instruction issue (committed PMC pass of this build)
                rl["valu_busy"] = rl["issue_frac"] = units["valu_busy_frac"]
            if "SQ_LDS_IDX_ACTIVE" in c:
                units["lds_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
                units["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c["SQ_LDS_IDX_ACTIVE"], 1.0)
            if "TA_TA_BUSY_sum" in c:
                units["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / (256.0 * cyc)
            for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_WAVES"):
                if k in c:
                    units[k] = c[k]
            rl["units"] = units
    return rl


def exchange_legs(eng, fwd, adj, grad_t, o_t, d_t, order_t, R, k2, torch, dist, world, backend):
    """The per-iteration exchange of the back-projected update over the N ranks, by itself and inside an iteration, with
    what proves which library carried it: backend, world size, RCCL version, bytes and milliseconds per all-reduce."""
    from ionotomo_amd.parallel import GradientExchange
    out = {"backend": dist.get_backend(), "requested_backend": backend, "world_size": dist.get_world_size()}
    try:
        out["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                               # noqa: BLE001
        out["nccl_version"] = None
    ones = torch.ones(R, dtype=torch.float64, device=eng.device)
    xch = GradientExchange("compact").plan(eng.adjoint(o_t, d_t, ones, TMAX, NS, order=order_t))
    xch32 = GradientExchange("compact", reduce_dtype=torch.float32)
    xch32.index, xch32.fraction = xch.index, xch.fraction
    dense = GradientExchange("dense")
    out["exchange_active_node_fraction"] = xch.fraction
    n_active = int(xch.index.numel()) if xch.index is not None else grad_t.numel()
    for name, x, nbytes in (("dense_f64", dense, grad_t.numel() * 8), ("compact_f64", xch, n_active * 8),
                            ("compact_f32", xch32, n_active * 4)):
        wall, _ = time_steps(lambda x=x: x.sum_(grad_t), k2, 2, torch, dist, world)
        out["allreduce_" + name] = {"bytes": nbytes, "ms": wall / k2 * 1e3,
                                    "bus_gbs": 2.0 * (world - 1) / world * nbytes / (wall / k2) / 1e9}

        def iteration(x=x):
            fwd()
            adj()
            x.sum_(grad_t)
        wall, _ = time_steps(iteration, k2, 1, torch, dist, world)
        out["iteration_ms_" + name] = wall / k2 * 1e3
    return out


def other_grid_leg(n, w, local, o_t, d_t, forder_t, R, kern256, k2, torch, dist):
    """The headline rays through an n^3 grid over the same box (n odd: columns start on 8-byte boundaries), planned forward with
    Ns = nz samples per ray, checked against the unplanned kernel of the same engine."""
    from ionotomo_amd.engine import RayEngine
    e = RayEngine(local, storage="f64")
    e.set_grid(*[np.linspace(w[k][0], w[k][-1], n) for k in ("xvec", "yvec", "zvec")])
    gen = torch.Generator(device=e.device)
    gen.manual_seed(n)
    e.set_values(torch.rand(n ** 3, dtype=torch.float64, device=e.device, generator=gen) + 1.0)
    ns = n if n % 2 else n + 1
    out = torch.empty(R, dtype=torch.float64, device=e.device)
    direct = e.forward(o_t, d_t, TMAX, ns, order=forder_t).clone()
    info = e.plan_forward(o_t, d_t, TMAX, ns)
    _, k = time_steps(lambda: e.forward(o_t, d_t, TMAX, ns, out=out, order=forder_t), k2, 2, torch, dist, 1)
    assert not e.check_oob()
    return {"grid": [n] * 3, "Ns": ns, "bundles": info[0], "lds_chunk_fraction": info[2], "forward_ms": k * 1e3,
            "ray_integrals_per_s": R / k, "samples_per_s": R * ns / k, "per_sample_rate_vs_256_cubed": (R * ns / k) / (R * NS / kern256),
            "max_rel_dev_vs_unplanned_kernel": float(((out - direct).abs() / direct.abs()).max())}


def engine_with_env(env, local, **kw):
    """A RayEngine whose context read ``env`` at creation (the A/B switches are read once per context)."""
    from ionotomo_amd.engine import RayEngine
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return RayEngine(local, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def coherence_legs(w, local, m_t, k2, torch, dist):
    """`extra.coherence_sweep`: how far the planned kernels' rates reach beyond the 100-timestep batch.
    (a) `same_rays`: the SAME 260,400 rays per GPU as 62 stations x (4200 / Nt) directions x Nt timesteps, Nt = 1, 4, 16, 100 -- what
        the rate owes to temporal coherence.  (b) `pipeline_batches`: 62 x 42 directions x Nt timesteps, Nt = 1, 4, 16, 32 -- the batches
        the reference's pipeline forms (one coherence window = 4 timesteps, inversion/inversion_pipeline.py:41-50); the plan decides per
        batch which bundles are worth a workgroup (iono_forward_plan_split) and the launch floor shows.  (c) `mixed`: half the headline
        rays + as many scattered rays (bundles of one or two): the case the per-bundle choice is for, against both all-or-nothing
        dispatches.  (d) `config2_sized_cgls`: a CGLS iteration at 2,604 and 10,416 rays through 128^3, eager and as a hipGraph.
    (e) `parallel_solves`: 1 and 32 single-time-step solves (2,604 rays, 128^3 each) side by side in one set of launches."""
    from ionotomo_amd import parallel, solvers, synthetic as syn
    ants = syn.lofar_enu_km()

    def rays_for(nd, nt):
        o, d = syn.ray_bundle(ants, syn.rotate_about_pole(syn.facet_directions(nd, 4.0, 1), nt))
        return o.reshape(-1, 3), d.reshape(-1, 3)

    def grid_engine(env=None):
        e = engine_with_env(env or {}, local, storage="f64")
        e.set_grid(w["xvec"], w["yvec"], w["zvec"])
        e.set_log_model(m_t, w["K_ne"] / 1e13)
        return e

    def forward_ms(e, o, d, steps):
        ot, dt = e.tensor(o), e.tensor(d)
        out = torch.empty(ot.shape[0], dtype=torch.float64, device=e.device)
        e.plan_forward(ot, dt, TMAX, NS)
        sp = e.forward_plan_split()
        order = None if sp["bundles_served"] else e.coherent_order(ot, dt)
        # (median of three short windows: these legs are 10-100 us launches, and one disturbed window must not write the sweep)
        ks = [time_steps(lambda: e.forward(ot, dt, TMAX, NS, out=out, order=order), max(steps, 20), 2, torch, dist, 1,
                         settle_ms=min(SETTLE_MS, 40.0) if i == 0 else 0.0)[1] for i in range(3)]
        assert not e.check_oob()
        return float(np.median(ks)) * 1e3, sp, (ot, dt, out)

    out = {"same_rays": [], "pipeline_batches": []}
    e = grid_engine()
    ref_rate = None
    for nt in (100, 16, 4, 1):
        o, d = rays_for(ND * NT // nt, nt)
        ms, sp, (ot, dt, _) = forward_ms(e, o, d, k2)
        R = o.shape[0]
        ainfo = e.plan_adjoint(ot, dt, TMAX, NS)
        y = torch.ones(R, dtype=torch.float64, device=e.device)
        g = torch.zeros(e.shape, dtype=torch.float64, device=e.device)

        def adj():
            g.zero_()
            e.adjoint(ot, dt, y, TMAX, NS, out=g)
        _, ka = time_steps(adj, max(3, k2 // 2), 1, torch, dist, 1, settle_ms=min(SETTLE_MS, 40.0))
        rate = R / (ms * 1e-3)
        ref_rate = rate if nt == 100 else ref_rate
        out["same_rays"].append({"Nt": nt, "Nd": ND * NT // nt, "rays": R, "forward_ms": ms, "ray_integrals_per_s": rate,
                                 "rate_vs_Nt_100": rate / ref_rate, "rays_per_bundle_cut": R / max(sp["bundles_cut"], 1), "plan": sp,
                                 "adjoint_ms": ka * 1e3, "adjoint_segments": ainfo[0], "adjoint_work_units": ainfo[1]})
        del y, g
    for nt in (1, 4, 16, 32):
        o, d = rays_for(ND, nt)
        ms, sp, _ = forward_ms(e, o, d, 50)
        out["pipeline_batches"].append({"Nt": nt, "rays": o.shape[0], "forward_us": ms * 1e3, "ray_integrals_per_s": o.shape[0] / (ms * 1e-3),
                                        "rate_vs_Nt_100": o.shape[0] / (ms * 1e-3) / ref_rate, "plan": sp})
    # (c) mixed geometry: 130,200 headline rays + 130,200 scattered ones
    rng = np.random.default_rng(7)
    o, d = rays_for(ND, NT)
    half = o.shape[0] // 2
    keep = rng.choice(o.shape[0], half, replace=False)
    lo, hi = ants.min(0), ants.max(0)
    o = np.concatenate([o[keep], np.stack([rng.uniform(lo[a], hi[a], half) for a in range(3)], 1)])
    d = np.concatenate([d[keep], syn.facet_directions(half, 4.0, 11)])
    mixed = {"rays": o.shape[0]}
    ref = None
    for name, env in (("plan_choice", {}), ("all_bundles", {"IONOTOMO_HYBRID_MIN": 1}), ("lanes_samples_only", {"IONOTOMO_HYBRID_MIN": 65})):
        em = e if not env else grid_engine(env)
        ms, sp, (_, _, tec) = forward_ms(em, o, d, max(3, k2 // 2))
        mixed["forward_ms_" + name] = ms
        if not env:
            mixed["plan"] = sp
            ref = tec.clone()
        else:
            mixed["max_rel_dev_%s_vs_plan_choice" % name] = float(((tec - ref).abs() / ref.abs()).max())
        if env:
            del em
    out["mixed"] = mixed
    del e
    # (d) config-2-sized CGLS iterations (128^3 grid, 62 x 42 x Nt rays): eager loop against the captured graph
    out["config2_sized_cgls"] = []
    for nt in (1, 4):
        w2 = syn.make_workload(antennas="lofar", na=NA, nd=ND, nt=nt, n=128)
        e2 = engine_with_env({}, local, storage="f64")
        e2.set_grid(w2["xvec"], w2["yvec"], w2["zvec"])
        oo, dd = w2["origins"].reshape(NA, -1, 3), w2["directions"].reshape(NA, -1, 3)
        P = oo.shape[1]
        prob = parallel.ShardedRays(e2, oo, dd, w2["tmax"], w2["Ns"], dobs=np.zeros((NA, P)), cdct=np.full((NA, P), 1e-6), i0=0, tune=False)
        x0 = e2.tensor(w2["ne"] / 1e13).reshape(e2.shape)
        e2.set_values((x0 * 1.1).reshape(-1))
        prob.dobs = prob.forward().clone()
        rec = {"Nt": nt, "rays": NA * P, "grid": [128] * 3}
        for tag, graph in (("eager", False), ("graph", True)):
            solvers.cgls(prob, x0, n_iter=4, graph=graph)
            w40, _ = time_steps(lambda: solvers.cgls(prob, x0, n_iter=40, graph=graph), 3, 1, torch, dist, 1, settle_ms=0.0)
            w10, _ = time_steps(lambda: solvers.cgls(prob, x0, n_iter=10, graph=graph), 3, 1, torch, dist, 1, settle_ms=0.0)
            rec["cgls_us_per_iteration_marginal_" + tag] = (w40 - w10) / 3 / 30 * 1e6
        out["config2_sized_cgls"].append(rec)
        del prob, e2
    # (e) the pipeline's own batch -- ONE time step per solve (inversion/inversion_pipeline.py:131-216), `num_parallel_solves` of them at
    #     once -- as B solves stacked along x in one set of launches (inversion/parallel_solves.py): microseconds PER SOLVE
    from ionotomo_amd.inversion.parallel_solves import StackedSolves
    out["parallel_solves"] = []
    Bs = (1, 32)
    dirs = syn.rotate_about_pole(syn.facet_directions(ND, 4.0, 1), max(Bs))
    o_all, d_all = syn.ray_bundle(ants, dirs)                                      # [Na, B, Nd, 3]: the field at B consecutive time steps
    grid = syn.domain_for(o_all, d_all, 128, TMAX, 4)
    ne0 = syn.ne_model(*grid, seed=7, corr=30.0) / 1e11
    for B in Bs:
        st = StackedSolves(tuple(grid), count=B, device=local)
        o, d = st.rays([o_all[:, b] for b in range(B)], [d_all[:, b] for b in range(B)], TMAX)
        es = st.engine
        models = [torch.as_tensor(ne0 * (1.0 + 0.01 * b)) for b in range(B)]
        es.set_values(st.stack_grids([m_ * 1.05 for m_ in models]).reshape(-1))
        ot, dt = es.tensor(o.reshape(-1, 3)), es.tensor(d.reshape(-1, 3))
        t = es.forward(ot, dt, TMAX, 129).reshape(NA, -1)
        dobs = (t - t[0:1]).cpu().numpy()
        prob = parallel.ShardedRays(es, o, d, TMAX, 129, dobs=dobs, cdct=np.full(dobs.shape, 1e-4), i0=0, tune=False)
        x0 = st.stack_grids(models)
        es.set_values(x0.reshape(-1))
        fwd_ = lambda: prob.forward_tec()
        y = torch.randn(ot.shape[0], dtype=torch.float64, device=es.device)
        ks = sorted(time_steps(fwd_, 50, 3, torch, dist, 1, settle_ms=20.0 if i == 0 else 0.0)[1] for i in range(3))
        g = torch.zeros(es.shape, dtype=torch.float64, device=es.device)
        ka = sorted(time_steps(lambda: es.adjoint(prob.origins, prob.dirs, y, TMAX, 129, out=g, order=prob._adjoint_order()), 30, 3, torch, dist, 1,
                               settle_ms=0.0)[1] for i in range(3))
        del g
        rec = {"solves": B, "rays": int(ot.shape[0]), "grid_per_solve": [128] * 3, "forward_us_per_solve": ks[1] * 1e6 / B,
               "adjoint_us_per_solve": ka[1] * 1e6 / B, "forward_kernel": es.describe("forward", prob.origins, prob.dirs, TMAX, 129)[0]}
        solvers.sirt(prob, x0, n_iter=4)
        w40, _ = time_steps(lambda: solvers.sirt(prob, x0, n_iter=40), 3, 1, torch, dist, 1, settle_ms=0.0)
        w10, _ = time_steps(lambda: solvers.sirt(prob, x0, n_iter=10), 3, 1, torch, dist, 1, settle_ms=0.0)
        rec["sirt_us_per_iteration_marginal_per_solve"] = (w40 - w10) / 3 / 30 * 1e6 / B
        out["parallel_solves"].append(rec)
        del prob, es, st, x0, y, ot, dt
    return out


def fermat_problems(w, local, torch, which=("cfg3", "cfg4")):
    """The Fermat (refractive-bending) integrator of north_star at BASELINE config 3 -- 62 x 42 = 2,604 curved rays through 128^3 -- and
    at config 4's ray count -- 620,000 curved rays through 256^3, traced AND integrated in one launch without a ray tensor
    (iono_forward_tec_fermat_dev).  Grids with a 16-cell margin: this synthetic ionosphere bends 120 MHz rays by kilometres, and a ray
    that leaves the grid raises (the reference's bounds_error=True).  Returns {name: (engine, origins, directions, tmax, Ns, frequency,
    substeps)}; the node values are ne in m^-3 (the refractive index is derived from them)."""
    from ionotomo_amd import synthetic as syn
    from ionotomo_amd.engine import RayEngine
    out = {}
    if "cfg3" in which:
        w3 = syn.make_workload("cfg2", margin_cells=16)
        e3 = RayEngine(local)
        e3.set_grid(w3["xvec"], w3["yvec"], w3["zvec"])
        e3.set_values(e3.tensor(w3["ne"]))
        out["cfg3"] = (e3, e3.tensor(w3["origins"].reshape(-1, 3)), e3.tensor(w3["directions"].reshape(-1, 3)), w3["tmax"], w3["Ns"], 120e6, 4)
    if "cfg4" in which:
        c4 = build_cfg4(w)
        o, d = c4["origins"].reshape(-1, 3), c4["directions"].reshape(-1, 3)
        e4 = RayEngine(local)
        e4.set_grid(*syn.domain_for(o, d, NGRID, TMAX, margin_cells=16))
        e4.set_values(torch.exp(e4.tensor(w["m"])).mul_(w["K_ne"]).reshape(-1))
        out["cfg4"] = (e4, e4.tensor(o), e4.tensor(d), TMAX, NS, 150e6, 2)
    return out


def fermat_bytes_per_ray(ns, substeps, index_kind, integrand_corners=8):
    """Algorithmic bytes of one curved ray, no credit for reuse (the convention of SURVEY 8d): every RK4 stage evaluates n and grad n
    in the 8 corners of its cell -- 64-byte Lekien-Marsden records through a tricubic index, 8-byte values through a trilinear one --
    (Ns - 1) x substeps x 4 times; every sample the integrand's 8 corner values; origin + direction in, TEC out."""
    return (ns - 1) * substeps * 4 * 8 * (64 if index_kind == "cubic" else 8) + ns * integrand_corners * 8 + 56


def fermat_leg(w, local, torch, dist):
    """`extra.fermat`: config 3 and the 620,000-ray fused forward, both refractive-index interpolants, default routes; + a
    `fermat_roofline` block for the 620,000-ray kernels (the unit that binds them comes from profiles/r05_pmc_summary.json)."""
    out = {}
    for name, (e, o, d, tmax, ns, freq, sub) in fermat_problems(w, local, torch).items():
        R = int(o.shape[0])
        t = torch.empty(R, dtype=torch.float64, device=e.device)
        for kind in ("cubic", "linear"):
            fn = lambda: e.forward_fermat(o, d, tmax, ns, freq, bend=True, kind=kind, substeps=sub, out=t)      # noqa: E731
            _, k = time_steps(fn, 10 if name == "cfg3" else 3, 2 if name == "cfg3" else 1, torch, dist, 1)
            assert not e.check_oob(), "bending rays left the grid"
            b = fermat_bytes_per_ray(ns, sub, kind)
            # error control: step doubling on a strided sample (<= 1 % of the rays, >= 208) at the reference's odeint
            # tolerance (inversion/fermat.py:163-167).  What the fixed `substeps` of the timed launch means is read off the levels: the
            # largest position difference (km) and the relative TEC change between s and 2 s steps.  On this turbulent test ionosphere a
            # trilinear index converges first order (its gradient jumps at cell faces), a tricubic one second order: 1.49e-8 is out of
            # reach of any affordable fixed step, and the report says so (`met`: false) instead of hiding it.
            _, rep = e.choose_fermat_substeps(o, d, tmax, ns, freq, kind=kind, max_substeps=16)
            out["%s_%s_index" % (name, kind)] = {
                "rays": R, "Ns": ns, "substeps": sub, "frequency_hz": freq, "ms": k * 1e3, "rays_per_s": R / k,
                "step_control": {"rtol": rep["rtol"], "atol": rep["atol"], "sample_rays": rep["sample_rays"], "met_within_16_substeps": rep["met"],
                                 "levels": rep["levels"],
                                 "timed_substeps_position_diff_km": next((max(l["max_abs_diff"].values()) for l in rep["levels"] if l["substeps"] == sub), None),
                                 "timed_substeps_tec_rel_diff": next((l.get("observable_max_rel_diff") for l in rep["levels"] if l["substeps"] == sub), None)},
                "kernel": ("k_fermat_tec_lm<true, %d, false>" % (2 if R >= 32768 else 8) if e.fermat_lm_ok(kind, "linear", R)
                           else "k_fermat_tec<%d, true, false>" % (kind == "cubic")),      # (lanes per ray: the library's default rule)
                "algorithmic_bytes_per_ray": b, "algorithmic_gbs": R * b / k / 1e9}
        del t
        if name == "cfg4":          # the TRANSPOSE of the same launch (re-trace + back-project, no ray tensor): default routes
            y = torch.ones(R, dtype=torch.float64, device=e.device)
            g = torch.zeros(e.shape, dtype=torch.float64, device=e.device)
            for kind in ("cubic", "linear"):
                fn = lambda: e.adjoint_fermat(o, d, y, tmax, ns, freq, bend=True, kind=kind, substeps=sub, out=g)      # noqa: E731
                _, k = time_steps(fn, 3, 1, torch, dist, 1)
                e.check_oob()
                out["%s_%s_index" % (name, kind)]["transpose_ms"] = k * 1e3
                out["%s_%s_index" % (name, kind)]["transpose_kernel"] = (
                    "k_fermat_tec_lm<true, 2, true>" if e.fermat_lm_ok(kind, "linear", R, transpose=True)
                    else "k_trace_fermat_lm + k_adjoint_rays (ray tensor)" if e._two_step_fermat(R, ns, kind, None, adjoint=True)
                    else "k_fermat_tec<%d, true, true>" % (kind == "cubic"))
            del y, g
    return out


def cfg4_leg(w, local, k2, torch, dist, world):
    """BASELINE config 4 as written: 62 x 100 x 100 = 620,000 rays, 256^3 grid, rays sharded over the N ranks by (time,
    direction) block (ShardedRays / pair_block), the adjoint update summed over ranks every iteration.  Total work is fixed:
    STRONG scaling."""
    from ionotomo_amd import parallel, solvers
    from ionotomo_amd.engine import RayEngine
    c4 = build_cfg4(w)
    e4 = RayEngine(local, storage="f64")
    e4.set_grid(c4["xvec"], c4["yvec"], c4["zvec"])
    x0 = torch.exp(e4.tensor(c4["m"])).mul_(c4["K_ne"] / 1e13).reshape(e4.shape)
    e4.set_values(x0.reshape(-1))
    P = c4["origins"].shape[1]
    prob = parallel.ShardedRays(e4, c4["origins"], c4["directions"], TMAX, NS, dobs=np.zeros((NA, P)), cdct=np.full((NA, P), 1e-6),
                                i0=0, tune=False)
    Rtot = NA * P
    out = {"rays_total": Rtot, "rays_this_rank": prob.R_local, "scaling": "strong", "exchange": prob.exchange.mode,
           "forward_plan": prob.forward_plan, "adjoint_plan": prob.plan}
    wall, kern = time_steps(prob.forward_tec, k2, 2, torch, dist, world)
    assert not e4.check_oob(), "config-4 rays left the grid"
    out["forward_ray_integrals_per_s"] = Rtot * k2 / wall
    out["forward_ms"] = wall / k2 * 1e3
    e4.set_values((x0 * 1.1).reshape(-1))
    prob.dobs = prob.forward().clone()
    e4.set_values(x0.reshape(-1))
    tec = prob.forward_tec()
    k3 = max(3, k2 // 2)

    def iteration():
        prob.forward_tec()
        prob.gradient_from_tec(tec)
    wall, _ = time_steps(iteration, k3, 1, torch, dist, world)
    out["iteration_ms_fwd_adj_exchange"] = wall / k3 * 1e3
    for name in ("cgls", "sirt"):
        fn = getattr(solvers, name)
        fn(prob, x0, n_iter=2)
        # (whole solves, set-up included -- normalisations, active set, buffers -- timed like every other leg: settled, wall clock, MAX over ranks)
        w10, _ = time_steps(lambda: fn(prob, x0, n_iter=10), 3, 1, torch, dist, world)
        out["%s_ms_per_iteration" % name] = w10 / 3 / 10 * 1e3
        # BASELINE config 5 "as 4": the whole 50-iteration inversion
        w50, _ = time_steps(lambda: fn(prob, x0, n_iter=50), 2, 0, torch, dist, world)
        out["%s_50_iterations_ms" % name] = w50 / 2 * 1e3
        out["%s_ms_per_iteration_marginal" % name] = (w50 / 2 - w10 / 3) / 40 * 1e3        # an iteration without the solve's set-up
    if grouped(world):
        # the same iterations with the exchange hidden behind the back-projection (exchange="overlap": the plan in z-slabs, every slab's
        # finished node levels all-reduced asynchronously while the next slab is back-projected), float64 and float32 on the links
        del prob
        for tag, rd in (("overlap_f64", None), ("overlap_f32", torch.float32), ("compact_f32", torch.float32)):
            try:
                pr = parallel.ShardedRays(e4, c4["origins"], c4["directions"], TMAX, NS, dobs=np.zeros((NA, P)), cdct=np.full((NA, P), 1e-6),
                                          i0=0, tune=False, exchange="overlap" if tag.startswith("overlap") else "compact", reduce_dtype=rd)
                e4.set_values((x0 * 1.1).reshape(-1))
                pr.dobs = pr.forward().clone()
                sub = {"overlapped": bool(pr.overlapped()), "slabs": len(pr.slab_ranges or [])}
                for name in ("cgls", "sirt"):
                    fn = getattr(solvers, name)
                    fn(pr, x0, n_iter=2)
                    torch.cuda.synchronize()
                    dist.barrier()
                    t0 = time.perf_counter()
                    fn(pr, x0, n_iter=10)
                    torch.cuda.synchronize()
                    dist.barrier()
                    sub["%s_ms_per_iteration" % name] = (time.perf_counter() - t0) / 10 * 1e3
                sub["slabs"] = len(pr.slab_ranges or [])             # (the ranges exist once a solver has asked for the active set)
                out[tag] = sub
                del pr
                ok = 1
            except Exception as exc:                                # noqa: BLE001
                out[tag] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=e4.device)      # all ranks or none enter the next variant
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not int(flag.item()):
                break
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--settle-ms", type=float, default=None,
                    help="untimed launches of each leg for this long before its warmups (see settle()); default 150, 0 with --only")
    ap.add_argument("--extras-timeout", type=int, default=420,
                    help="N > 1: seconds the legs after the headline may take before rank 0 prints the line without them (0: no watchdog)")
    ap.add_argument("--test-hang", type=float, default=0.0, help=argparse.SUPPRESS)      # test-only: the legs after the headline "hang"
    ap.add_argument("--no-order", dest="order", action="store_false", help="walk rays in [Na][Nt][Nd] memory order")
    ap.add_argument("--no-plan", dest="plan", action="store_false",
                    help="back-project with the ray-stationary kernel (LDS tile per ray bundle) instead of the box-binned plan")
    ap.add_argument("--no-fwd-plan", dest="fwd_plan", action="store_false",
                    help="forward without the bundle plan (lanes = samples kernel k_forward_straight_u on the coherent walk order)")
    ap.add_argument("--no-cfg4", dest="cfg4", action="store_false", help="skip the config-4 (620,000 rays, strong scaling) leg")
    ap.add_argument("--main-only", action="store_true", help="same as --only forward")
    ap.add_argument("--only", default=None,
                    choices=["forward", "adjoint", "cubic_forward", "cubic_adjoint", "cgls", "sirt", "fermat_cubic", "fermat_linear",
                             "fermat_cfg3", "coherence", "f32_forward"],
                    help="time ONE leg alone (clean rocprofv3 --stats / --pmc averages); implies --no-cpu")
    args = ap.parse_args()
    if args.main_only:
        args.only = "forward"
    global SETTLE_MS
    SETTLE_MS = float(args.settle_ms) if args.settle_ms is not None else (0.0 if args.only else 150.0)

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
    global FORCE_GROUP
    FORCE_GROUP = world == 1 and os.environ.get("IONO_BENCH_FORCE_GROUP", "0") not in ("", "0")
    if grouped(world):
        if FORCE_GROUP:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ionotomo_amd import parallel, solvers
    from ionotomo_amd.engine import RayEngine
    parallel.FORCE_COLLECTIVES = FORCE_GROUP        # (the sharded problems then exchange on the one-rank group too)
    w = build_workload(rank)
    R = w["origins"].shape[0]
    eng = RayEngine(local, storage="f64")
    eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
    m_t = eng.tensor(w["m"])
    eng.set_log_model(m_t, w["K_ne"] / 1e13)
    o_t, d_t = eng.tensor(w["origins"]), eng.tensor(w["directions"])
    tec_t = torch.empty(R, dtype=torch.float64, device=eng.device)
    grid_bytes = NGRID ** 3 * 8

    # walk order: rays whose paths nearly coincide run back to back (geometry only, computed once,
    # reused by every launch of an inversion; results are independent of it)
    order_t = eng.locality_order(o_t, d_t, TMAX) if args.order else None
    # the unplanned forward kernels walk the rays in the "coherent" order (RayEngine.coherent_order; geometry only, once)
    forder_t = eng.coherent_order(o_t, d_t) if args.order else None
    fwd = eng.forward_launcher(o_t, d_t, TMAX, NS, tec_t, order=forder_t)
    # bundle plan of the forward (geometry only, once per inversion like the walk orders: engine.plan_forward)
    fwd_plan_info = None
    if args.fwd_plan:
        t0 = time.perf_counter()
        info = eng.plan_forward(o_t, d_t, TMAX, NS)
        fwd_plan_info = {"bundles": info[0], "chunks_per_ray": info[1], "lds_chunk_fraction": info[2],
                         "build_s": time.perf_counter() - t0}
    planned = bool(fwd_plan_info and fwd_plan_info["bundles"])

    # ---- legs that can run alone under a profiler ---------------------------------------------------------------
    def adjoint_leg():
        fwd()
        tec0 = tec_t.cpu().numpy().reshape(NA, -1)
        dobs = eng.tensor(tec0 - tec0[0] + np.random.default_rng(2 + rank).normal(size=tec0.shape) * 1e-3)
        cdct = torch.full((R,), 1e-6, dtype=torch.float64, device=eng.device)
        grad = torch.zeros(eng.shape, dtype=torch.float64, device=eng.device)

        def adj():
            grad.zero_()
            eng.adjoint_residual(o_t, d_t, tec_t, dobs, cdct, NA, 0, TMAX, NS, out=grad, order=order_t)
        return adj, grad, dobs, cdct

    def cubic_legs():
        ec = RayEngine(local, storage="f64", interp="cubic")
        ec.set_grid(w["xvec"], w["yvec"], w["zvec"])
        ec.set_log_model(m_t, w["K_ne"] / 1e13)
        tc = torch.empty_like(tec_t)
        gc = torch.zeros(ec.shape, dtype=torch.float64, device=ec.device)
        yc = torch.randn(R, dtype=torch.float64, device=ec.device)

        def cf():
            ec.forward(o_t, d_t, TMAX, NS, out=tc, order=forder_t)

        def ca():
            gc.zero_()
            ec.adjoint(o_t, d_t, yc, TMAX, NS, out=gc, order=order_t)
        if args.fwd_plan:          # bundles of neighbouring rays, one Lekien-Marsden field pair per wave (k_forward_bundle_lm)
            ec.plan_forward(o_t, d_t, TMAX, NS)
        return ec, cf, ca, tc

    def solver_problem():
        fwd()
        oo, dd = w["origins"].reshape(NA, -1, 3), w["directions"].reshape(NA, -1, 3)
        prob = parallel.ShardedRays(eng, oo, dd, TMAX, NS, dobs=np.zeros((NA, R // NA)), cdct=np.full((NA, R // NA), 1e-6), i0=0)
        x0 = torch.exp(m_t).mul_(w["K_ne"] / 1e13).reshape(eng.shape)
        eng.set_values((x0 * 1.1).reshape(-1))
        prob.dobs = prob.forward().clone()
        return prob, x0

    if args.only == "coherence":
        print(json.dumps({"only": "coherence", "csrc_sha": csrc_sha(), "coherence_sweep": coherence_legs(w, local, m_t, max(3, min(25, args.steps // 4)), torch, dist)}))
        return
    if args.only:
        k = args.steps
        if args.only == "forward":
            leg = fwd
        elif args.only == "adjoint":
            leg = adjoint_leg()[0]
            if args.plan:
                eng.plan_adjoint(o_t, d_t, TMAX, NS)
            elif order_t is not None:
                eng.tune_adjoint_partition(leg, R)
        elif args.only.startswith("fermat"):
            name = "cfg3" if args.only == "fermat_cfg3" else "cfg4"
            e_, o_, d_, tmax_, ns_, freq_, sub_ = fermat_problems(w, local, torch, which=(name,))[name]
            kind_ = "linear" if args.only == "fermat_linear" else "cubic"
            t_ = torch.empty(o_.shape[0], dtype=torch.float64, device=e_.device)
            leg = lambda: e_.forward_fermat(o_, d_, tmax_, ns_, freq_, bend=True, kind=kind_, substeps=sub_, out=t_)      # noqa: E731
            k = max(2, min(k, 5))
        elif args.only == "f32_forward":         # the float32 fast mode (storage="f32" + a forward plan: k_forward_bundle_f32)
            e32 = RayEngine(local, storage="f32")
            e32.set_grid(w["xvec"], w["yvec"], w["zvec"])
            e32.set_log_model(m_t, w["K_ne"] / 1e13)
            if args.fwd_plan:
                e32.plan_forward(o_t, d_t, TMAX, NS)
            leg = e32.forward_launcher(o_t, d_t, TMAX, NS, tec_t, order=forder_t)
        elif args.only in ("cubic_forward", "cubic_adjoint"):
            ec, cf, ca, _ = cubic_legs()
            leg = cf if args.only == "cubic_forward" else ca
            if args.plan and args.only == "cubic_adjoint":
                ec.plan_adjoint(o_t, d_t, TMAX, NS)
        else:
            prob, x0 = solver_problem()
            fn = getattr(solvers, args.only)
            fn(prob, x0, n_iter=2)
            leg = lambda: fn(prob, x0, n_iter=10)
            k = max(1, args.steps // 10)
        wall, kern = time_steps(leg, k, min(args.warmup, 2), torch, dist, world)
        if rank == 0:
            per = 10 if args.only in ("cgls", "sirt") else 1
            print(json.dumps({"only": args.only, "n_gpus": world, "steps": k, "ms_per_launch_or_iteration": kern * 1e3 / per,
                              "rays": R, "forward_plan": fwd_plan_info, "csrc_sha": csrc_sha(), "settle_ms": SETTLE_MS}))
        if grouped(world):
            dist.destroy_process_group()
        return

    # the first K launches after idle (no settle phase), then the contract's measurement in the device's sustained state
    wall_cold, kern_cold = time_steps(fwd, args.steps, args.warmup, torch, dist, world, settle_ms=0.0)
    wall, kern, n_windows, window_ms = time_windows(fwd, args.steps, args.warmup, torch, dist, world)
    assert not eng.check_oob(), "rays left the grid"
    value = world * R * args.steps * n_windows / wall
    split = eng.forward_plan_split() if planned else None
    tec_gpu = tec_t.cpu().numpy()
    sha = csrc_sha()
    line = {
        "metric": "ray-integrals/sec through 256^3 ne grid",
        "value": value, "unit": "ray-integrals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / (args.steps * n_windows) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "62 ant (LOFAR-HBA) x 42 dir x 100 times per GPU = %d straight rays, Ns=%d, 256^3 f64 ne "
                               "grid, trilinear + Simpson, forward TEC" % (R, NS),
                   "rays_per_gpu": R, "samples_per_ray": NS, "grid": [NGRID] * 3, "interp": "trilinear",
                   "quadrature": "simpson", "sharding": "rays by (time,direction) block, grid replicated",
                   # (the library's dispatch table, asked for THIS launch: iono_dispatch_describe)
                   "forward_kernel": eng.describe("forward", o_t, d_t, TMAX, NS)[0],
                   "forward_plan_split": split,
                   # the contract's window (K steps between barriers) repeated M times back to back so that the timed region lasts >= 20 ms:
                   # `value` = all rays of the M windows / their total wall time; the spread over the windows next to it
                   "windows": n_windows, "window_ms_per_step_median": float(np.median(window_ms)), "window_ms_per_step_min": min(window_ms),
                   "window_ms_per_step_max": max(window_ms), "timed_region_ms": wall * 1e3,
                   # the first K launches after idle (no settle phase): what `--settle-ms 0` measures
                   "cold_window_ms_per_step": wall_cold / args.steps * 1e3, "cold_window_value": world * R * args.steps / wall_cold,
                   "settle_ms": SETTLE_MS},
        "csrc_sha": sha,
    }
    if FORCE_GROUP:
        line["config"]["forced_one_rank_group"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                   "note": "rehearsal: the multi-rank control flow and every collective on a 1-rank group"}
    if rank == 0:
        # the headline is on record before any leg that contains a collective (the ONE stdout line comes at the end)
        print("bench.py headline (repeated in the final stdout line): " + json.dumps(line), file=sys.stderr, flush=True)

    # everything below is reported next to the headline number.  Nothing in it may be able to lose the headline line or
    # hang a multi-rank run: every rank allocates what the legs need FIRST, the ranks agree that all of them succeeded,
    # and only then enter code with collectives (the same on every rank); after every leg with a collective the ranks
    # agree again before the next one starts.
    extra = {"forward_plan": fwd_plan_info,
             "headline_cold_window": {"ms_per_step": wall_cold / args.steps * 1e3, "kernel_ms": kern_cold * 1e3,
                                      "ray_integrals_per_s": world * R * args.steps / wall_cold,
                                      "note": "the same W + K launches straight after idle, without the settle phase (bench.py:settle)"}}
    # N > 1: a watchdog over everything that follows.  A collective that never returns (a rank lost, a mismatch) must not cost the
    # headline: after --extras-timeout seconds rank 0 prints the line with what it has and every rank leaves.
    import threading
    finish_lock, finished = threading.Lock(), [False]
    ctx = (world, rank, dist, sha, R, kern, grid_bytes, planned, args, w, tec_gpu)

    def finish(extra_, copy_gbs_, clean):
        finish_line(line, extra_, copy_gbs_, clean, ctx)

    def on_timeout():
        print("bench.py watchdog: rank %d, %d s after the headline" % (rank, args.extras_timeout), file=sys.stderr, flush=True)
        with finish_lock:
            if finished[0]:
                return
            finished[0] = True
        # the headline line goes out FIRST, then every rank leaves with a non-zero code: a run whose collective never returned is
        # not a success for the harness / torchrun; ranks != 0 wait long enough for rank 0's CPU baseline + print,
        # so that torchrun's teardown cannot cut the line short
        if rank == 0:
            ex = dict(extra)
            ex["error"] = "the legs after the headline did not finish within %d s (watchdog): line printed without them" % args.extras_timeout
            try:
                finish(ex, None, False)
            finally:
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(WATCHDOG_EXIT_CODE)
        time.sleep(90)
        os._exit(WATCHDOG_EXIT_CODE)

    watchdog = None
    if grouped(world) and args.extras_timeout > 0:
        watchdog = threading.Timer(args.extras_timeout, on_timeout)
        watchdog.daemon = True
        watchdog.start()
        if args.test_hang > 0:                          # (--test-hang, hidden: rehearsal of the watchdog by profiles/tools)
            time.sleep(args.test_hang)

    def agree(ok):
        if not grouped(world):
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=eng.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    ok = True
    try:
        adj, grad_t, dobs_t, cdct_t = adjoint_leg()
        eng32 = RayEngine(local, storage="f32")
        eng32.set_grid(w["xvec"], w["yvec"], w["zvec"])
        eng32.set_log_model(m_t, w["K_ne"] / 1e13)
        tec32 = torch.empty_like(tec_t)
        big_a = torch.empty(1 << 27, dtype=torch.float64, device=eng.device)         # 1 GiB
        big_b = torch.empty_like(big_a)
        ec, cf, ca, tc = cubic_legs()
    except Exception as exc:                                    # noqa: BLE001
        ok = False
        extra["error"] = "%s: %s" % (type(exc).__name__, exc)
    ok = agree(ok)
    copy_gbs = None
    k2 = max(3, min(25, args.steps // 4))
    if ok:
        # ---- legs WITHOUT collectives (a failure here costs this rank's extras only) ------------------------------------
        try:
            # ray-stationary back-projection (LDS tile per bundle of the walk) with its work partition balanced by measured
            # cost, then the node-stationary one (segments binned by grid box, engine.plan_adjoint): both depend on the
            # ray geometry only, are set up once per inversion and never change results
            extra["adjoint_partition"] = eng.tune_adjoint_partition(adj, R) if args.order else None
            _, tkern = time_steps(adj, k2, 1, torch, dist, 1)
            extra["adjoint_ray_stationary_ms"] = tkern * 1e3
            if args.plan:
                t0 = time.perf_counter()
                info = eng.plan_adjoint(o_t, d_t, TMAX, NS)
                extra["adjoint_plan"] = {"segments": info[0], "work_units": info[1], "outside_fraction": info[2],
                                         "build_s": time.perf_counter() - t0}
            _, akern = time_steps(adj, k2, 1, torch, dist, 1)
            extra["adjoint_ms"] = akern * 1e3
            extra["adjoint_ray_integrals_per_s_per_gpu"] = R / akern
            if args.plan:
                # the same with order-independent fixed-point accumulation (RayEngine.set_deterministic: run-to-run identical bits)
                eng.set_deterministic(True)
                _, adk = time_steps(adj, k2, 1, torch, dist, 1)
                eng.set_deterministic(False)
                extra["adjoint_deterministic_ms"] = adk * 1e3
            # ---- the lanes = samples forward on the coherent walk order (what the planned kernel replaced)
            if planned:
                eng.clear_forward_plan()
                _, ku = time_steps(fwd, k2, 1, torch, dist, 1)
                extra["forward_unplanned_ms"] = ku * 1e3
                eng.plan_forward(o_t, d_t, TMAX, NS)
            # ---- the reference's default sampling, Ns = nz (even), with the 'avg' rule its own integrate.py spells out
            # (secondary row; tests/golden/forward_tec_even_avg.npz pins the rule)
            # -- with a bundle plan of ITS OWN sample count (a plan is keyed on Ns; round 4 timed this row on the unplanned kernel)
            tec_even = torch.empty(R, dtype=torch.float64, device=eng.device)
            if planned:
                extra["even_ns_forward_plan_bundles"] = eng.plan_forward(o_t, d_t, TMAX, NS - 1)[0]
            _, kev = time_steps(lambda: eng.forward(o_t, d_t, TMAX, NS - 1, out=tec_even, order=forder_t), k2, 1, torch, dist, 1)
            extra["even_ns_avg_rule_ms"] = kev * 1e3
            extra["even_ns_avg_rule_ray_integrals_per_s_per_gpu"] = R / kev
            extra["even_ns_avg_rule_vs_odd_max_rel_dev"] = float(((tec_even - tec_t).abs() / tec_t.abs()).max())
            if planned:
                eng.plan_forward(o_t, d_t, TMAX, NS)
            # ---- grids of the other parity: N = ceil(extent / spacing) per axis comes out odd as often as even
            # (inversion/initial_model.py:29-34), and the reference samples Ns = nz points per ray (geometry/calc_rays.py:111-112).
            # Same box, same rays, 255^3 and 257^3 nodes, Ns = nz (odd: Simpson proper); rate per SAMPLE against the headline's
            if planned:
                extra["odd_grids"] = [other_grid_leg(n, w, local, o_t, d_t, forder_t, R, kern, k2, torch, dist) for n in (255, 257)]
            # ---- float32 grid storage (float64 arithmetic) and the single-timestep launch
            # unplanned: float64 arithmetic on the rounded values (k_forward_straight_q4); with a forward plan: the float32 FAST MODE
            # (k_forward_bundle_f32: float32 window images, packed-float32 interpolation, float64 sums of the chunk sums) -- a storage /
            # precision mode of its own, never the float64 headline
            _, k32u = time_steps(lambda: eng32.forward(o_t, d_t, TMAX, NS, out=tec32, order=forder_t), k2, 1, torch, dist, 1)
            extra["f32_grid_unplanned_ray_integrals_per_s_per_gpu"] = R / k32u
            extra["f32_grid_unplanned_max_rel_err_vs_f64"] = float((tec32 - tec_t).abs().div(tec_t.abs()).max().item())
            if args.fwd_plan:
                info32 = eng32.plan_forward(o_t, d_t, TMAX, NS)
                _, k32 = time_steps(lambda: eng32.forward(o_t, d_t, TMAX, NS, out=tec32, order=forder_t), max(k2, 20), 2, torch, dist, 1)
                assert not eng32.check_oob()
                dt32, dt64 = (t.view(NA, -1) - t.view(NA, -1)[0:1] for t in (tec32, tec_t))
                extra["f32_fast_mode"] = {"kernel": "k_forward_bundle_f32", "bundles": info32[0], "lds_chunk_fraction": info32[2], "forward_ms": k32 * 1e3,
                                          "vs_float64_headline_kernel": kern / k32,
                                          "dtec_max_abs_err_over_max_tec": float((dt32 - dt64).abs().max() / tec_t.abs().max()),
                                          "algorithmic_bytes_per_ray": algorithmic_bytes_per_ray(NS, 4),
                                          "algorithmic_gbs": R * algorithmic_bytes_per_ray(NS, 4) / k32 / 1e9}
            else:
                k32 = k32u
            extra["f32_grid_ray_integrals_per_s_per_gpu"] = R / k32
            extra["f32_grid_max_rel_err_vs_f64"] = float((tec32 - tec_t).abs().div(tec_t.abs()).max().item())
            sel = torch.arange(R, device=eng.device).reshape(NA, NT, ND)[:, 0, :].reshape(-1)
            o1, d1 = o_t[sel].contiguous(), d_t[sel].contiguous()
            t1 = torch.empty(o1.shape[0], dtype=torch.float64, device=eng.device)
            # (a latency figure, 50 launches of a few microseconds each: no settle phase -- a stream of launch-bound launches leaves the
            #  device between power states and reads 10-13 us)
            _, k1 = time_steps(lambda: eng.forward(o1, d1, TMAX, NS, out=t1), 50, 5, torch, dist, 1, settle_ms=0.0)
            extra["single_timestep_rays"] = int(o1.shape[0])
            extra["single_timestep_us"] = k1 * 1e6
            # ---- measured device-to-device copy (1 GiB read + 1 GiB written): the achievable HBM rate on this box
            _, kc = time_steps(lambda: big_b.copy_(big_a), 10, 2, torch, dist, 1)
            copy_gbs = 2.0 * big_a.numel() * 8 / kc / 1e9
            del big_a, big_b
            # ---- tricubic (Lekien-Marsden derivative fields; config 2's interpolant) at the same shape
            _, kcf = time_steps(cf, max(k2, 10), 3, torch, dist, 1)
            if args.fwd_plan:
                # with new node values every call (what an inversion iteration pays: the derivative fields are rebuilt), and lanes = samples
                xc = torch.exp(m_t).mul_(w["K_ne"] / 1e13)

                def cf_new_values():
                    ec.set_values(xc)
                    cf()
                _, kcn = time_steps(cf_new_values, max(k2, 10), 3, torch, dist, 1)
                extra["tricubic_forward_new_values_ms"] = kcn * 1e3
                ec.clear_forward_plan()
                _, kcu = time_steps(cf, max(2, k2 // 2), 1, torch, dist, 1)
                extra["tricubic_forward_unplanned_ms"] = kcu * 1e3
            if args.plan:
                ec.plan_adjoint(o_t, d_t, TMAX, NS)
            _, kca = time_steps(ca, max(k2, 10), 3, torch, dist, 1)            # (steady state: the ~2 ms launches settle after two)
            extra["tricubic_forward_ms"] = kcf * 1e3
            extra["tricubic_forward_ray_integrals_per_s_per_gpu"] = R / kcf
            extra["tricubic_adjoint_ms"] = kca * 1e3
            if args.plan:
                ec.set_deterministic(True)
                _, kcd = time_steps(ca, max(k2, 10), 3, torch, dist, 1)
                ec.set_deterministic(False)
                extra["tricubic_adjoint_deterministic_ms"] = kcd * 1e3
            extra["tricubic_vs_trilinear_max_rel_dev"] = float((tc - tec_t).abs().div(tec_t.abs()).max().item())
            # the binned kernel is bound by LDS float-atomic throughput, the ray-stationary one by the memory-side atomic rate
            extra["adjoint_roofline"] = {"bound": "lds_atomic" if args.plan else "memory_atomic", "kernel_ms": akern * 1e3,
                                         "kernel": "k_adjoint_binned<double, 0, double" if args.plan
                                         else "k_adjoint_straight_tile<double, 1, 4>"}
            if not grouped(world):                          # single-rank only: reach of the planned kernels beyond the 100-timestep batch
                extra["coherence_sweep"] = coherence_legs(w, local, m_t, k2, torch, dist)
            if not grouped(world):                          # single-rank only: the Fermat integrator (config 3 + config 4's ray count)
                extra["fermat"] = fermat_leg(w, local, torch, dist)
            if not grouped(world):                          # single-rank only: the solvers at the bench shape
                prob, x0 = solver_problem()
                for name in ("cgls", "sirt"):
                    fn = getattr(solvers, name)
                    fn(prob, x0, n_iter=2)
                    # (whole solves, set-up included, settled like every other leg; `_marginal`: an iteration without the set-up)
                    w30, _ = time_steps(lambda: fn(prob, x0, n_iter=30), 3, 1, torch, dist, 1)
                    w10, _ = time_steps(lambda: fn(prob, x0, n_iter=10), 3, 1, torch, dist, 1)
                    extra["%s_ms_per_iteration" % name] = w30 / 3 / 30 * 1e3
                    extra["%s_ms_per_iteration_marginal" % name] = (w30 - w10) / 3 / 20 * 1e3
                del prob
                if planned:                                 # (the solver problem planned its own tensors: back to the bench's)
                    eng.plan_forward(o_t, d_t, TMAX, NS)
                if args.plan:
                    eng.plan_adjoint(o_t, d_t, TMAX, NS)
        except Exception as exc:                                    # noqa: BLE001
            extra["error"] = "%s: %s" % (type(exc).__name__, exc)
        # ---- legs WITH collectives: all ranks or none ---------------------------------------------------------------------
        if agree("error" not in extra):
            try:
                def iteration():
                    fwd()
                    adj()
                    if grouped(world):
                        dist.all_reduce(grad_t)
                iwall, _ = time_steps(iteration, k2, 1, torch, dist, world)
                extra["iteration_ms_fwd_adj_allreduce"] = iwall / k2 * 1e3
            except Exception as exc:                                    # noqa: BLE001
                extra["error"] = "%s: %s" % (type(exc).__name__, exc)
        if grouped(world) and agree("error" not in extra):
            try:
                extra["distributed"] = exchange_legs(eng, fwd, adj, grad_t, o_t, d_t, order_t, R, k2, torch, dist, world, backend)
            except Exception as exc:                                    # noqa: BLE001
                extra["error"] = "%s: %s" % (type(exc).__name__, exc)
        # ---- config 4: 620,000 rays split over the ranks (strong scaling), forward / iteration / CGLS / SIRT -------------------
        if args.cfg4 and agree("error" not in extra):
            try:
                extra["cfg4"] = cfg4_leg(w, local, k2, torch, dist, world)
            except Exception as exc:                                    # noqa: BLE001
                extra["cfg4"] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    if watchdog is not None:
        watchdog.cancel()
    with finish_lock:
        if finished[0]:            # (the watchdog is printing the line: let it)
            time.sleep(3600)
        finished[0] = True
    finish(extra, copy_gbs, True)


def finish_line(line, extra, copy_gbs, clean, ctx):
    """The tail of main(): roofline + CPU baseline + the ONE stdout line.  clean=False: called by the watchdog while the main thread
    sits in a leg that did not return -- no collective, no GPU tool."""
    (world, rank, dist, sha, R, kern, grid_bytes, planned, args, w, tec_gpu) = ctx
    pmc, pmc_note = load_pmc(sha)
    if grouped(world) and clean:
        # every collective is behind us: leave the group BEFORE rank 0's host-side legs (peaks tool, CPU baseline), so that no
        # rank waits in a collective while rank 0 computes on the host
        dist.barrier()
        dist.destroy_process_group()
    peaks = measured_peaks() if rank == 0 and clean else None
    rl = forward_roofline(R, kern, pmc, pmc_note, copy_gbs, grid_bytes, peaks, planned)
    ar = extra.get("adjoint_roofline")
    ca_ = (pmc or {}).get("adjoint")
    if ar and ca_ and ca_.get("rays") == R and "TCC_EA0_ATOMIC_sum" in ca_ and ca_.get("kernel", "").startswith(ar["kernel"]):
        ab = 64.0 * ca_["TCC_EA0_ATOMIC_sum"]                      # 64-B atomic requests leaving L2 per launch
        mem = ab / (ar["kernel_ms"] * 1e-3) / 1e9
        ar["memory_atomics"] = {"requests_per_launch": ca_["TCC_EA0_ATOMIC_sum"], "achieved_gbs": mem, "peak_gbs": ATOMIC_PEAK_GBS,
                                "frac": mem / ATOMIC_PEAK_GBS}
        if "SQ_LDS_IDX_ACTIVE" in ca_ and "GRBM_GUI_ACTIVE" in ca_:
            cyc = ca_["GRBM_GUI_ACTIVE"] / 8.0
            ar["lds"] = {"busy_frac": ca_["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc),
                         "bank_conflict_frac": ca_.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(ca_["SQ_LDS_IDX_ACTIVE"], 1.0),
                         "atomic_wave_instructions": ca_.get("SQ_INSTS_LDS_ATOMIC")}
        ar["frac"] = ar["lds"]["busy_frac"] if ar["bound"] == "lds_atomic" and "lds" in ar else ar["memory_atomics"]["frac"]
    # the Fermat integrator's dominant kernel (620,000 curved rays through a tricubic index): what binds it is vector-instruction issue,
    # not bytes -- the corner records of a cell stay in registers while the ray stays in the cell
    fz = (extra.get("fermat") or {}).get("cfg4_cubic_index")
    if fz:
        fr = {"kernel": fz["kernel"], "bound": "valu_issue", "rays": fz["rays"], "kernel_ms": fz["ms"],
              "algorithmic_bytes_per_ray": fz["algorithmic_bytes_per_ray"], "achieved": fz["algorithmic_gbs"], "unit": "GB/s",
              "note": "algorithmic bytes = (Ns - 1) x substeps x 4 RK4 stages x 8 corner records x 64 B + the integrand's 8 x 8 B per sample + 56; "
                      "`issue_*` from the committed PMC pass of this build (profiles/pmc_counters.json, leg fermat_cubic) when it matches"}
        cf = (pmc or {}).get("fermat_cubic")
        if cf and cf.get("rays") == fz["rays"] and "SQ_INSTS_VALU" in cf and "GRBM_GUI_ACTIVE" in cf:
            cyc = cf["GRBM_GUI_ACTIVE"] / 8.0
            fr["issue_vector_wave_instructions"] = cf["SQ_INSTS_VALU"]
            fr["issue_floor_ms_at_4_cycles"] = cf["SQ_INSTS_VALU"] * 4.0 / 1024.0 / (cyc / (fz["ms"] * 1e-3)) * 1e3
            if "SQ_ACTIVE_INST_VALU" in cf:
                fr["valu_busy"] = fr["frac"] = min(1.0, 4.0 * cf["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc))
            if "TA_TA_BUSY_sum" in cf:
                fr["ta_busy"] = cf["TA_TA_BUSY_sum"] / (256.0 * cyc)
            fr["kernel_pmc"] = cf.get("kernel")
        extra["fermat_roofline"] = fr
    # the float32 fast mode's kernel: busy fractions of its units from the committed PMC pass of this build (leg f32_forward)
    f32m, c32 = extra.get("f32_fast_mode"), (pmc or {}).get("f32_forward")
    if f32m and c32 and c32.get("rays") == R and "GRBM_GUI_ACTIVE" in c32:
        cyc = c32["GRBM_GUI_ACTIVE"] / 8.0
        f32m["units"] = {k: v for k, v in (
            ("valu_busy_frac", 4.0 * c32["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc) if "SQ_ACTIVE_INST_VALU" in c32 else None),
            ("lds_busy_frac", c32["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc) if "SQ_LDS_IDX_ACTIVE" in c32 else None),
            ("lds_bank_conflict_frac", c32.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c32.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)),
            ("ta_busy_frac", c32["TA_TA_BUSY_sum"] / (256.0 * cyc) if "TA_TA_BUSY_sum" in c32 else None),
            ("SQ_INSTS_VALU", c32.get("SQ_INSTS_VALU")), ("SQ_INSTS_LDS", c32.get("SQ_INSTS_LDS"))) if v is not None}
        f32m["lds_read_path_frac_of_guide"] = f32m["algorithmic_gbs"] / LDS_GUIDE_GBS
    line["roofline"] = rl
    line["extra"] = extra
    if rank == 0 and not args.no_cpu:
        # rank 0's shard of the timed output against the CPU oracle + the oracle timed on this host (N > 1: the same bounded
        # sample on rank 0, after the group is gone -- VERDICT r3 item 8c)
        cb, relerr = cpu_baseline(w, tec_gpu)
        line["cpu_baseline"] = cb
        line["parity_max_rel_err_vs_cpu_oracle"] = relerr
        assert relerr < 1e-6, "GPU TEC differs from the CPU oracle by %g" % relerr
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()

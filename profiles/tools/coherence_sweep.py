#!/usr/bin/env python3
"""The SAME 260 400 rays as 62 antennas x (4200 / Nt) directions x Nt timesteps, Nt in {1, 4, 16, 100}: how the planned kernels'
rates depend on the temporal coherence of the batch.  Per point: the forward plan (bundles, rays per bundle,
fraction of chunks served from LDS), forward ms by dispatch (default / bundle kernel forced / lanes = samples forced), and the
back-projection (box plan / ray-stationary tiles).  A/B tool: `bench.py` carries the short form as `extra.coherence_sweep`.

    python profiles/tools/coherence_sweep.py [--nt 1 4 16 100] [--hybrid-min 8 16 24 32]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def timed(fn, torch, n=20, warm=3, settle_ms=60.0):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    per = max(time.perf_counter() - t0, 2e-5)
    for _ in range(int(min(5000, settle_ms * 1e-3 / per)) + warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def engine(env, w, torch, interp="linear"):
    from ionotomo_amd.engine import RayEngine
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        e = RayEngine(0, storage="f64", interp=interp)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, nargs="*", default=[1, 4, 16, 100])
    ap.add_argument("--hybrid-min", type=int, nargs="*", default=[], help="forced thresholds to time next to the plan's own choice")
    ap.add_argument("--mixed", action="store_true", help="half the headline rays + as many scattered rays (bundles of one or two)")
    ap.add_argument("--fixed-nd", action="store_true", help="42 directions x Nt timesteps (R grows with Nt): the batches a pipeline forms")
    ap.add_argument("--cubic", action="store_true")
    ap.add_argument("--no-adjoint", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    from ionotomo_amd import synthetic as syn
    w = bench.build_workload(0)
    ants = syn.lofar_enu_km()
    out = {"csrc_sha": bench.csrc_sha(), "points": []}
    for nt in args.nt:
        nd = bench.ND if args.fixed_nd else bench.ND * bench.NT // nt
        dirs = syn.rotate_about_pole(syn.facet_directions(nd, 4.0, 1), nt)
        o, d = syn.ray_bundle(ants, dirs)
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
        if args.mixed:
            rng = np.random.default_rng(7)
            half = o.shape[0] // 2
            keep = rng.choice(o.shape[0], half, replace=False)
            lo, hi = ants.min(0), ants.max(0)
            o2 = np.stack([rng.uniform(lo[0], hi[0], half), rng.uniform(lo[1], hi[1], half), rng.uniform(lo[2], hi[2], half)], 1)
            d2 = syn.facet_directions(half, 4.0, 11)
            o, d = np.concatenate([o[keep], o2]), np.concatenate([d[keep], d2])
        R = o.shape[0]
        pt = {"Nt": nt, "Nd": nd, "rays": R, "mixed": bool(args.mixed)}
        variants = [("default", {}), ("all_bundles", {"IONOTOMO_HYBRID_MIN": 1}), ("unplanned", {"IONOTOMO_HYBRID_MIN": 65})
                    ]
        variants += [("hybrid_min_%d" % h, {"IONOTOMO_HYBRID_MIN": h}) for h in args.hybrid_min]
        ref = None
        for name, env in variants:
            e = engine(env, w, torch, "cubic" if args.cubic else "linear")
            ot, dt = e.tensor(o), e.tensor(d)
            tec = torch.empty(R, dtype=torch.float64, device=e.device)
            forder = e.coherent_order(ot, dt)
            info = e.plan_forward(ot, dt, bench.TMAX, bench.NS)
            if name == "default":
                pt["forward_plan"] = {"bundles": info[0], "lds_chunk_fraction": info[2]}
                try:
                    pt["forward_plan"].update(e.forward_plan_split(histogram=True))
                except AttributeError:
                    pass
            ms = timed(lambda: e.forward(ot, dt, bench.TMAX, bench.NS, out=tec, order=forder), torch)
            assert not e.check_oob()
            pt["forward_ms_" + name] = ms
            if name.startswith("hybrid_min"):
                try:
                    pt["split_" + name] = e.forward_plan_split()
                except AttributeError:
                    pass
            if ref is None:
                ref = tec.clone()
            else:
                pt["max_rel_dev_%s_vs_default" % name] = float(((tec - ref).abs() / ref.abs()).max())
            if name in ("default", "unplanned") and not args.cubic and not args.no_adjoint:
                y = torch.ones(R, dtype=torch.float64, device=e.device)
                g = torch.zeros(e.shape, dtype=torch.float64, device=e.device)
                order = e.locality_order(ot, dt, bench.TMAX)
                if name == "default":
                    ainfo = e.plan_adjoint(ot, dt, bench.TMAX, bench.NS)
                    pt["adjoint_plan"] = {"segments": ainfo[0], "work_units": ainfo[1], "segment_lanes": e.plan_segment_lanes()}

                def adj():
                    g.zero_()
                    e.adjoint(ot, dt, y, bench.TMAX, bench.NS, out=g, order=order)
                pt["adjoint_ms_" + ("planned" if name == "default" else "ray_stationary")] = timed(adj, torch, n=10)
            del e, ot, dt, tec
        pt["forward_rate_default"] = R / (pt["forward_ms_default"] * 1e-3)
        out["points"].append(pt)
        print(json.dumps(pt), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

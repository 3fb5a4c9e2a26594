#!/usr/bin/env python3
"""One pipeline-sized forward (62 x 42 x Nt rays, 256^3 bench grid) launched back to back: run under
`rocprofv3 --kernel-trace --stats` to read the KERNEL's duration next to the launch interval the events see (what bounds a
coherence window: the kernel or the dispatch).   python3 profiles/tools/small_batch_trace.py <Nt> [launches]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    import torch
    import bench
    from ionotomo_amd import synthetic as syn
    from ionotomo_amd.engine import RayEngine
    w = bench.build_workload(0)
    e = RayEngine(0)
    e.set_grid(w["xvec"], w["yvec"], w["zvec"])
    e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
    o, d = syn.ray_bundle(syn.lofar_enu_km(), syn.rotate_about_pole(syn.facet_directions(bench.ND, 4.0, 1), nt))
    ot, dt = e.tensor(o.reshape(-1, 3)), e.tensor(d.reshape(-1, 3))
    out = torch.empty(ot.shape[0], dtype=torch.float64, device=e.device)
    e.plan_forward(ot, dt, bench.TMAX, bench.NS)
    sp = e.forward_plan_split()
    order = None if sp["bundles_served"] else e.coherent_order(ot, dt)
    fn = e.forward_launcher(ot, dt, bench.TMAX, bench.NS, out, order=order)
    for _ in range(50):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(json.dumps({"Nt": nt, "rays": int(ot.shape[0]), "launches": n, "kernel": e.describe("forward", ot, dt, bench.TMAX, bench.NS)[0],
                      "launch_interval_us": a.elapsed_time(b) * 1e3 / n}))


if __name__ == "__main__":
    main()

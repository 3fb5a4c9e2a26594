"""Lanes per ray of the record tracer / fused curved-ray TEC through a tricubic index (IONOTOMO_FERMAT_LM_LANES = 8, 4, 2, 1): times and
agreement at config 4's ray count (620 000 rays, 256^3) and at config 3 (2 604 rays, 128^3).  One JSON line."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench


def timeit(fn, n, warm=1):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


w = bench.build_workload(0)
out, ref = {}, {}
for lanes in ("8", "4", "2", "1"):
    os.environ["IONOTOMO_FERMAT_LM_LANES"] = lanes
    for name, (e, o, d, tmax, ns, freq, sub) in bench.fermat_problems(w, 0, torch).items():
        R = int(o.shape[0])
        for ne_kind in ("linear", "cubic"):
            t = torch.empty(R, dtype=torch.float64, device=e.device)
            fn = lambda: e.forward_fermat(o, d, tmax, ns, freq, bend=True, kind="cubic", substeps=sub, ne_kind=ne_kind, out=t, fused=True)
            ms = timeit(fn, 10 if name == "cfg3" else 2)
            assert not e.check_oob()
            key = "%s_%s_integrand" % (name, ne_kind)
            if lanes == "8":
                ref[key] = t.clone()
            out["%s_lanes%s_ms" % (key, lanes)] = ms
            out["%s_lanes%s_max_rel_dev_vs_8" % (key, lanes)] = float(((t - ref[key]).abs() / ref[key].abs()).max())
        if name == "cfg3":                      # the tracer proper (rays[R,4,Ns]) with the same lanes
            rays = torch.empty((R, 4, ns), dtype=torch.float64, device=e.device)
            out["cfg3_trace_lanes%s_ms" % lanes] = timeit(lambda: e.trace_fermat(o, d, tmax, ns, freq, bend=True, kind="cubic", substeps=sub, out=rays), 10)
            if lanes == "8":
                ref["rays"] = rays.clone()
            out["cfg3_trace_lanes%s_max_abs_dev_km_vs_8" % lanes] = float((rays - ref["rays"]).abs().max())
print(json.dumps(out))

"""620 000 bending rays x 257 samples through 256^3 (config 4's ray count): the curved-ray TRANSPOSE through a tricubic and a
trilinear refractive index (trilinear integrand) -- default route, and trace + explicit-sample transpose (fused=False: a 5.1 GB ray
tensor); IONOTOMO_VARIANT=3 in the environment gives the lanes = rays kernel as the fused route.  One JSON line."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench

w = bench.build_workload(0)
e, o, d, tmax, ns, freq, sub = bench.fermat_problems(w, 0, torch, which=("cfg4",))["cfg4"]
R = int(o.shape[0])
y = torch.randn(R, dtype=torch.float64, device=e.device)
out = {"R": R, "Ns": ns, "substeps": sub, "variant": os.environ.get("IONOTOMO_VARIANT", "default")}
g = torch.zeros(e.shape, dtype=torch.float64, device=e.device)
ref = {}
for kind in ("cubic", "linear"):
    for name, fused in (("default", None), ("two_step", False)):
        if kind == "linear" and fused is False:
            continue
        fn = lambda: e.adjoint_fermat(o, d, y, tmax, ns, freq, bend=True, kind=kind, substeps=sub, out=g.zero_(), fused=fused)   # noqa: E731
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        out["%s_%s_ms" % (kind, name)] = (time.perf_counter() - t0) / 3 * 1e3
        out["%s_%s_two_step_route" % (kind, name)] = bool(e._two_step_fermat(R, ns, kind, fused, adjoint=True))
        if kind == "cubic":
            ref[name] = g.clone()
out["cubic_default_vs_two_step_rel"] = float((ref["default"] - ref["two_step"]).abs().max() / ref["two_step"].abs().max())
print(json.dumps(out))

"""Tricubic Fermat tracer: 8-lanes-per-ray (default) vs lanes = rays (IONOTOMO_VARIANT=3) over the batch size."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from ionotomo_amd import synthetic as syn
from ionotomo_amd.engine import RayEngine

w = syn.make_workload("cfg2", margin_cells=16)
eng = RayEngine(0, interp="linear")
eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
eng.set_values(eng.tensor(w["ne"]))
o, d = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
out = {"variant": os.environ.get("IONOTOMO_VARIANT", "default"), "lin4_max": os.environ.get("IONOTOMO_FERMAT_LIN4_MAX", "default")}
for rep in (1, 4, 10, 30, 100):
    jit = 0.05 * torch.randn((rep, 1, 3), dtype=torch.float64, device="cuda") * torch.tensor([1.0, 1.0, 0.0], device="cuda", dtype=torch.float64)
    ob = (o[None] + jit).reshape(-1, 3).contiguous()
    db = d[None].expand(rep, -1, -1).reshape(-1, 3).contiguous()
    buf = torch.empty((ob.shape[0], 4, w["Ns"]), dtype=torch.float64, device="cuda")
    for kind in ("cubic", "linear"):
        eng.trace_fermat(ob, db, w["tmax"], w["Ns"], 120e6, bend=True, kind=kind, substeps=4, out=buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.trace_fermat(ob, db, w["tmax"], w["Ns"], 120e6, bend=True, kind=kind, substeps=4, out=buf)
        torch.cuda.synchronize()
        out["%s_R%d_ms" % (kind, ob.shape[0])] = (time.perf_counter() - t0) / 3 * 1e3
    del buf
print(json.dumps(out))

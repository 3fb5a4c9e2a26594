"""620 000 bending rays x 257 samples through 256^3 by the fused curved-ray kernel, three launches: the workload of a
`rocprofv3 --pmc ... --kernel-trace -- python3 profiles/tools/fermat_fused_probe.py` pass on k_fermat_tec."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ionotomo_amd import synthetic as syn
from ionotomo_amd.engine import RayEngine

w4 = syn.make_workload("cfg4", margin_cells=16)
e4 = RayEngine(0)
e4.set_grid(w4["xvec"], w4["yvec"], w4["zvec"])
e4.set_values(e4.tensor(w4["ne"]))
o4, d4 = e4.tensor(w4["origins"].reshape(-1, 3)), e4.tensor(w4["directions"].reshape(-1, 3))
t4 = torch.empty(o4.shape[0], dtype=torch.float64, device="cuda")
for _ in range(2):
    e4.forward_fermat(o4, d4, w4["tmax"], w4["Ns"], 150e6, bend=True, kind="linear", substeps=2, out=t4)
torch.cuda.synchronize()
t = time.perf_counter()
e4.forward_fermat(o4, d4, w4["tmax"], w4["Ns"], 150e6, bend=True, kind="linear", substeps=2, out=t4)
torch.cuda.synchronize()
print(json.dumps({"rays": o4.shape[0], "Ns": w4["Ns"], "fused_forward_ms": (time.perf_counter() - t) * 1e3}))
y4 = torch.randn(o4.shape[0], dtype=torch.float64, device="cuda")
g4 = torch.zeros(e4.shape, dtype=torch.float64, device="cuda")
e4.adjoint_fermat(o4, d4, y4, w4["tmax"], w4["Ns"], 150e6, bend=True, kind="linear", substeps=2, out=g4)
torch.cuda.synchronize()
t = time.perf_counter()
e4.adjoint_fermat(o4, d4, y4, w4["tmax"], w4["Ns"], 150e6, bend=True, kind="linear", substeps=2, out=g4)
torch.cuda.synchronize()
print(json.dumps({"fused_adjoint_ms": (time.perf_counter() - t) * 1e3}))

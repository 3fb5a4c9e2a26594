"""Back-projection of ONE coherence window (config 2: 2 604 rays x 129 samples, 128^3) by each kernel family: node-stationary plan,
ray-stationary LDS tiles, plain global atomics.  python profiles/tools/time_small_adjoint.py [variant]   (IONOTOMO_VARIANT is read at ctx creation)"""
import json, os, subprocess, sys, time
CHILD = r'''
import json, sys, time, torch
sys.path.insert(0, ".")
from ionotomo_amd import synthetic as syn
from ionotomo_amd.engine import RayEngine
w = syn.make_workload("cfg2")
eng = RayEngine(0)
eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
o, d = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
y = torch.randn(o.shape[0], dtype=torch.float64, device="cuda")
out = torch.zeros(eng.shape, dtype=torch.float64, device="cuda")
res = {}
for name in ("unplanned", "planned"):
    if name == "planned":
        res["plan"] = eng.plan_adjoint(o, d, w["tmax"], 129)
    for _ in range(20):
        eng.adjoint(o, d, y, w["tmax"], 129, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(500):
        eng.adjoint(o, d, y, w["tmax"], 129, out=out)
    torch.cuda.synchronize()
    res[name + "_us"] = (time.perf_counter() - t) / 500 * 1e6
print(json.dumps(res))
'''
out = {}
for v in ("0", "2"):
    env = dict(os.environ, IONOTOMO_VARIANT=v)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr); sys.exit(1)
    out["variant_" + v] = json.loads(r.stdout.strip().splitlines()[-1])
print(json.dumps(out, indent=1))

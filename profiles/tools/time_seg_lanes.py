"""Back-projection time against the lanes a segment occupies (IONOTOMO_SEG_LANES) for sparse and dense sampling of the bench grid.
Usage: python profiles/tools/time_seg_lanes.py > profiles/rNN_seg_lanes.json   (one process per width: the variable is read at ctx creation)"""
import json
import os
import subprocess
import sys

CHILD = r'''
import json, sys, time, torch
sys.path.insert(0, ".")
import bench
from ionotomo_amd.engine import RayEngine
Ns = int(sys.argv[1])
w = bench.build_workload(0)
eng = RayEngine(0)
eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
ot, dt = eng.tensor(w["origins"].reshape(-1, 3)), eng.tensor(w["directions"].reshape(-1, 3))
y = torch.randn(ot.shape[0], dtype=torch.float64, device="cuda")
segs, units, _ = eng.plan_adjoint(ot, dt, bench.TMAX, Ns)
out = torch.zeros(eng.shape, dtype=torch.float64, device="cuda")
for _ in range(5):
    eng.adjoint(ot, dt, y, bench.TMAX, Ns, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50):
    eng.adjoint(ot, dt, y, bench.TMAX, Ns, out=out)
torch.cuda.synchronize()
print(json.dumps({"Ns": Ns, "lanes": eng.plan_segment_lanes(), "segments": segs, "units": units, "adjoint_ms": (time.perf_counter() - t) / 50 * 1e3}))
'''
res = []
for Ns in (65, 129, 257):
    for lanes in ("", "4", "8", "16"):
        env = dict(os.environ)
        env.pop("IONOTOMO_SEG_LANES", None)
        if lanes:
            env["IONOTOMO_SEG_LANES"] = lanes
        r = subprocess.run([sys.executable, "-c", CHILD, str(Ns)], env=env, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            sys.exit(1)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        rec["forced"] = lanes or "auto"
        res.append(rec)
        sys.stderr.write(json.dumps(rec) + "\n")
print(json.dumps(res, indent=1))

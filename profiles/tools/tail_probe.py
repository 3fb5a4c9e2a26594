import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
e = RayEngine(0)
e.set_grid(w["xvec"], w["yvec"], w["zvec"])
e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
def run(o, d, tag):
    ot, dt = e.tensor(o), e.tensor(d)
    nb = e.plan_forward(ot, dt, bench.TMAX, bench.NS)[0]
    out = torch.empty(ot.shape[0], dtype=torch.float64, device="cuda")
    for _ in range(3): e.forward(ot, dt, bench.TMAX, bench.NS, out=out)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10): e.forward(ot, dt, bench.TMAX, bench.NS, out=out)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 10)
    print(tag, "bundles", nb, "ms %.4f" % np.median(ts), "us per bundle-round(1024) %.2f" % (np.median(ts) * 1e3 / (nb / 1024)))
o, d = w["origins"], w["directions"]
run(o, d, "1x")
run(np.concatenate([o, o]), np.concatenate([d, d]), "2x")
run(np.concatenate([o, o, o, o]), np.concatenate([d, d, d, d]), "4x")
n = int(len(o) * 4096 / 4597 * 0.98)
run(o[:n], d[:n], "~4 rounds")

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["IONOTOMO_PLAN_STATS"] = "1"
import bench
from ionotomo_amd.engine import RayEngine
w = bench.build_workload(0)
e = RayEngine(0, interp="cubic")
e.set_grid(w["xvec"], w["yvec"], w["zvec"])
e.set_log_model(e.tensor(w["m"]), w["K_ne"] / 1e13)
o, d = e.tensor(w["origins"]), e.tensor(w["directions"])
print(e.plan_forward(o, d, bench.TMAX, bench.NS))

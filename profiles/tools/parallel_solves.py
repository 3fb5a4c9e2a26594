#!/usr/bin/env python3
"""B independent single-time-step solves (62 antennas x 42 directions = 2 604 rays each, 128^3 grid: BASELINE config 2, the batch the
reference's pipeline forms per task) stacked along x (ionotomo_amd/inversion/parallel_solves.py) against the same solves one at a
time: forward, planned back-projection, SIRT iteration, microseconds PER SOLVE.

    python profiles/tools/parallel_solves.py [--B 1 4 16 32 64] > profiles/r06_parallel_solves.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, nargs="+", default=[1, 4, 16, 32, 64])
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--nt", type=int, default=1, help="time steps per solve (4 = one coherence window of the reference's pipeline)")
    ap.add_argument("--no-solvers", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    from ionotomo_amd import parallel, solvers, synthetic as syn
    from ionotomo_amd.inversion.parallel_solves import StackedSolves
    n, tmax = args.n, 1000.0
    Ns = n + 1
    ants = syn.lofar_enu_km()
    Bmax = max(args.B)
    nt = args.nt
    dirs = syn.rotate_about_pole(syn.facet_directions(42, 4.0, 1), Bmax * nt)     # the field at Bmax * nt consecutive time steps
    o_all, d_all = syn.ray_bundle(ants, dirs)                                       # [Na, Bmax nt, Nd, 3]
    grid = syn.domain_for(o_all, d_all, n, tmax, 4)
    Na, Nd = o_all.shape[0], o_all.shape[2] * nt
    o_all = o_all.reshape(Na, Bmax, Nd, 3)                                          # solve b = time steps [b nt, (b + 1) nt)
    d_all = d_all.reshape(Na, Bmax, Nd, 3)
    ne0 = syn.ne_model(*grid, seed=7, corr=30.0) / 1e11
    out = {"what": __doc__.split("\n\n")[0], "grid": [n] * 3, "rays_per_solve": Na * Nd, "time_steps_per_solve": nt, "Ns": Ns, "csrc_sha": bench.csrc_sha(), "points": []}
    bench.SETTLE_MS = 50.0

    def med(fn, steps):
        ks = sorted(bench.time_steps(fn, steps, 3, torch, None, 1)[1] for _ in range(3))
        return ks[1] * 1e6

    for B in args.B:
        st = StackedSolves(tuple(grid), count=B)
        o, d = st.rays([o_all[:, b] for b in range(B)], [d_all[:, b] for b in range(B)], tmax)
        eng = st.engine
        models = [torch.as_tensor(ne0 * (1.0 + 0.01 * b)) for b in range(B)]
        eng.set_values(st.stack_grids(models).reshape(-1))
        ot, dt = eng.tensor(o.reshape(-1, 3)), eng.tensor(d.reshape(-1, 3))
        R = ot.shape[0]
        tec = torch.empty(R, dtype=torch.float64, device=eng.device)
        rec = {"B": B, "rays": R, "nodes": int(np.prod(eng.shape))}
        fwd0 = eng.forward_launcher(ot, dt, tmax, Ns, tec)
        rec["forward_unplanned_us"] = med(fwd0, 50)
        info = eng.plan_forward(ot, dt, tmax, Ns)
        fwd = eng.forward_launcher(ot, dt, tmax, Ns, tec)
        rec["forward_us"] = med(fwd, 50)
        rec["forward_kernel"] = eng.describe("forward", ot, dt, tmax, Ns)[0]
        rec["forward_plan"] = {"bundles": info[0], "split": eng.forward_plan_split()}
        assert not eng.check_oob()
        y = torch.randn(R, dtype=torch.float64, device=eng.device)
        g = torch.zeros(eng.shape, dtype=torch.float64, device=eng.device)
        rec["adjoint_unplanned_us"] = med(lambda: eng.adjoint(ot, dt, y, tmax, Ns, out=g), 30)
        pinfo = eng.plan_adjoint(ot, dt, tmax, Ns)
        rec["adjoint_us"] = med(lambda: eng.adjoint(ot, dt, y, tmax, Ns, out=g), 30)
        rec["adjoint_plan"] = {"segments": pinfo[0], "work_units": pinfo[1]}
        del g, y
        eng.clear_forward_plan()
        eng.clear_adjoint_plan()
        if args.no_solvers:
            for k in list(rec):
                if k.endswith("_us"):
                    rec[k.replace("_us", "_us_per_solve")] = rec[k] / B
            out["points"].append(rec)
            print(json.dumps(rec), file=sys.stderr, flush=True)
            del st, eng, ot, dt, tec
            torch.cuda.empty_cache()
            continue
        # SIRT on the stacked problem (the solves' own data: forward of a perturbed model)
        eng.set_values(st.stack_grids([m * 1.05 for m in models]).reshape(-1))
        t = eng.forward(ot, dt, tmax, Ns).reshape(Na, -1)
        dobs = (t - t[0:1]).cpu().numpy()
        prob = parallel.ShardedRays(eng, o, d, tmax, Ns, dobs=dobs, cdct=np.full(dobs.shape, 1e-4), i0=0, tune=False)
        x0 = st.stack_grids(models)
        for name in ("sirt", "cgls"):
            fn = getattr(solvers, name)
            fn(prob, x0, n_iter=5)
            torch.cuda.synchronize()
            ts = {}
            for k in (10, 30):
                t0 = time.perf_counter()
                for _ in range(3):
                    fn(prob, x0, n_iter=k)
                torch.cuda.synchronize()
                ts[k] = (time.perf_counter() - t0) / 3
            rec[name + "_us_per_iteration_marginal"] = (ts[30] - ts[10]) / 20 * 1e6
        # B SEPARATE CGLS solves (one alpha / beta per solve: StackedSolves.cgls, torch vector passes)
        st.cgls(prob, x0, n_iter=3)
        torch.cuda.synchronize()
        ts = {}
        for k in (5, 15):
            t0 = time.perf_counter()
            for _ in range(3):
                st.cgls(prob, x0, n_iter=k)
            torch.cuda.synchronize()
            ts[k] = (time.perf_counter() - t0) / 3
        rec["cgls_separate_us_per_iteration_marginal"] = (ts[15] - ts[5]) / 10 * 1e6
        for k in list(rec):
            if k.endswith("_us") or k.endswith("_marginal"):
                rec[k.replace("_us", "_us_per_solve") if k.endswith("_us") else k + "_per_solve"] = rec[k] / B
        out["points"].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
        del prob, x0, st, eng, ot, dt, tec
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

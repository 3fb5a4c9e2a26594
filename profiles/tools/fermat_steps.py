#!/usr/bin/env python3
"""Step-doubling control of the RK4 Fermat tracer on the bench problems: per problem and index interpolant the
levels' estimates at the reference's odeint tolerance (rtol = atol = 1.49e-8) and at 1e-6, the chosen step count, its cost.

    python profiles/tools/fermat_steps.py > profiles/r06_fermat_steps.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    w = bench.build_workload(0)
    out = {"csrc_sha": bench.csrc_sha()}
    for name, (e, o, d, tmax, ns, freq, sub_old) in bench.fermat_problems(w, 0, torch).items():
        for kind in ("cubic", "linear"):
            rec = {"rays": int(o.shape[0]), "Ns": ns, "frequency_hz": freq, "substeps_rounds_1_to_5": sub_old}
            for rtol in (1.49012e-8, 1e-6):
                t0 = time.perf_counter()
                sub, rep = e.choose_fermat_substeps(o, d, tmax, ns, freq, kind=kind, rtol=rtol, max_substeps=64)
                torch.cuda.synchronize()
                rep["choose_ms"] = (time.perf_counter() - t0) * 1e3
                t = torch.empty(o.shape[0], dtype=torch.float64, device=e.device)
                fn = lambda: e.forward_fermat(o, d, tmax, ns, freq, bend=True, kind=kind, substeps=sub, out=t)      # noqa: E731
                _, k = bench.time_steps(fn, 3, 1, torch, None, 1, settle_ms=0.0)
                rep["forward_ms_at_chosen"] = k * 1e3
                rec["rtol_%.3g" % rtol] = rep
            out["%s_%s_index" % (name, kind)] = rec
            print(name, kind, json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel resources of the shipped library as the compiler emitted them: VGPRs, AGPRs, SGPRs, scratch bytes, LDS, occupancy
(waves per SIMD by registers) -- from the metadata of `hipcc --cuda-device-only -S` on ionotomo_hip.hip (no GPU needed).

    python profiles/tools/isa_resources.py [--filter SUBSTR] [--json out.json] [extra hipcc flags ...]

Exit code 1 when a kernel matching --must-not-spill uses scratch."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "ionotomo_amd", "csrc", "ionotomo_hip.hip")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--json", default=None)
    ap.add_argument("--asm", default=None, help="keep the assembly here")
    ap.add_argument("--must-not-spill", default=None, help="regex over demangled kernel names")
    args, extra = ap.parse_known_args()
    asm = args.asm or os.path.join(tempfile.mkdtemp(), "iono.s")
    if not (args.asm and os.environ.get("ISA_REUSE") and os.path.exists(asm)):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S", "-o", asm, SRC] + extra,
                              stderr=subprocess.DEVNULL)
    txt = open(asm).read()
    meta = txt[txt.index("amdhsa.kernels:"):]
    kernels = []
    for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda k: re.search(r"\.%s:\s*(\S+)" % k, blk)
        name = get("name").group(1)
        kernels.append({"symbol": name, "vgpr": int(get("vgpr_count").group(1)), "agpr": int(get("agpr_count").group(1)),
                        "sgpr": int(get("sgpr_count").group(1)), "scratch_bytes": int(get("private_segment_fixed_size").group(1)),
                        "lds_static_bytes": int(get("group_segment_fixed_size").group(1)),
                        "vgpr_spills": int(get("vgpr_spill_count").group(1)) if get("vgpr_spill_count") else 0,
                        "max_flat_workgroup_size": int(get("max_flat_workgroup_size").group(1))})
    dm = demangle([k["symbol"] for k in kernels])
    for k in kernels:
        full = re.sub(r"^void ", "", dm[k["symbol"]]).replace("(anonymous namespace)::", "")
        depth, cut = 0, len(full)
        for i, ch in enumerate(full):                      # the argument list: the first "(" outside template brackets
            depth += ch == "<"
            depth -= ch == ">"
            if ch == "(" and depth == 0:
                cut = i
                break
        k["name"] = full[:cut]
        tot = k["vgpr"] + k["agpr"]
        k["waves_per_simd_by_vgprs"] = min(8, 512 // max(8, (tot + 7) // 8 * 8))
    sel = [k for k in kernels if args.filter in k["name"]]
    sel.sort(key=lambda k: k["name"])
    for k in sel:
        print("%-110s vgpr %3d agpr %3d sgpr %3d scratch %4d B  waves/SIMD(regs) %d" % (k["name"][:110], k["vgpr"], k["agpr"], k["sgpr"],
                                                                                       k["scratch_bytes"], k["waves_per_simd_by_vgprs"]))
    print("%d kernels, %d with scratch" % (len(sel), sum(1 for k in sel if k["scratch_bytes"])))
    if args.json:
        json.dump({"kernels": sel, "n_kernels_total": len(kernels), "n_with_scratch": sum(1 for k in kernels if k["scratch_bytes"])},
                  open(args.json, "w"), indent=1)
    if args.must_not_spill:
        bad = [k["name"] for k in kernels if re.search(args.must_not_spill, k["name"]) and k["scratch_bytes"]]
        if bad:
            print("scratch in:", *bad, sep="\n  ")
            sys.exit(1)


if __name__ == "__main__":
    main()

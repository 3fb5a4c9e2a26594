"""Per-launch device durations of the headline kernel over N back-to-back launches after idle, from a rocprofv3 kernel trace:

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/ramp -- python3 $REPO/bench.py --only forward --steps 5000 --warmup 2
    python profiles/tools/clock_ramp.py gpurun_out/ramp > profiles/r05_clock_ramp.json"""
import csv, glob, json, statistics, sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
NAME = sys.argv[2] if len(sys.argv) > 2 else "k_forward_bundle"
rows = [r for r in csv.DictReader(open(f)) if NAME in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
s = [int(r["Start_Timestamp"]) for r in rows]
windows = [(0, 10), (10, 20), (20, 30), (30, 50), (50, 100), (100, 200), (200, 400), (400, 800), (800, 1600), (1600, 3200), (3200, len(d))]
out = {"what": ("k_forward_bundle<0>, bench shape (260 400 rays x 257 samples, 256^3)" if NAME == "k_forward_bundle" else NAME) + ": device duration of each of %d back-to-back launches "
               "after idle (rocprofv3 --kernel-trace of `bench.py --only forward --steps %d`), mean per window of launch indices" % (len(d), len(d) - 3),
       "mean_us_by_launch_index": {"%d-%d" % (a, min(b, len(d))): round(statistics.mean(d[a:b]), 2) for a, b in windows if a < len(d)},
       "elapsed_ms_at_index": {str(i): round((s[i] - s[0]) / 1e6, 2) for i in (10, 50, 100, 200, 400, 800, 1600, 3200) if i < len(d)},
       "first_30_us": [round(x, 1) for x in d[:30]],
       "reading": "Straight after idle the kernel runs at 106-109 us, slows to 115-131 us a few milliseconds in and reaches its sustained "
                  "96-98 us only after ~40 ms of continuous load, where it stays.  A W = 5, K = 20 measurement right after set-up sees launches "
                  "6-25 (~0.112-0.118 ms); bench.py therefore runs each leg untimed for ~150 ms before its warmups (settle()) and reports "
                  "the cold window next to the headline (extra.headline_cold_window)."}
print(json.dumps(out, indent=1))

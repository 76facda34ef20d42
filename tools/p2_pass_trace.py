"""Per-pass kernel time of the keyframe search from a rocprofv3 kernel trace (tools/p2_kstats.sh leaves it in gpurun_out/p2k/):
one line per pass (a pass starts at window_bounds_base_kernel) with the stages' kernel durations [us], their sum and the pass's
wall time.  usage: python tools/p2_pass_trace.py [trace.csv] [last N passes]"""
import csv
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/p2k/p2_kernel_trace.csv"
last = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["slice_hash_ref_both", "slice_lds_kernel<5120", "slice_big", "dbscan_pixel_both", "extract_both", "grid_order_kernel"]
passes, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "window_bounds_base" in n:
        cur = {"t0": int(r["Start_Timestamp"]), "sum": 0.0}
        passes.append(cur)
    if cur is None:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k in names:
        if k in n:
            cur[k] = cur.get(k, 0) + d
    cur["t1"] = max(cur.get("t1", 0), int(r["End_Timestamp"]))
    cur["sum"] += d
print("pass  " + " ".join("%9s" % k[:9] for k in names) + "       sum      wall")
tot = 0.0
for i, p in enumerate(passes[-last:]):
    wall = (p["t1"] - p["t0"]) / 1e3
    tot += wall if wall < 5000 else 0
    print("%4d  " % i + " ".join("%9.0f" % p.get(k, 0) for k in names) + "  %8.0f  %8.0f" % (p["sum"], wall))
print("wall of the passes shorter than 5 ms: %.1f ms" % (tot / 1e3))

"""Per-pass kernel times of the LAST adaptive-window search in a rocprofv3 kernel trace (tools/p2_kstats.sh leaves
gpurun_out/p2k/p2_kernel_trace.csv): passes are cut at adaptive_alloc_kernel; averages per kernel over three ranges of passes.
python tools/p2_passes.py [trace.csv]"""
import collections, csv, os, sys
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "p2k", "p2_kernel_trace.csv")
rows = list(csv.DictReader(open(path)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
            for r in rows if "ecal::" in r["Kernel_Name"] or "adaptive" in r["Kernel_Name"])
short = lambda n: n.replace("void ", "").replace("ecal::", "").replace("(anonymous namespace)::", "").split("(")[0][:46]
passes, cur = [], []
for k in ks:
    cur.append(k)
    if "adaptive_alloc_kernel" in k[2]:
        passes.append(cur)
        cur = []
inits = [i for i, p in enumerate(passes) if any("adaptive_init" in k[2] for k in p)]
sel = passes[inits[-1]:]
print("%d passes; kernels %.1f ms in all (under the profiler kernels do not overlap)" % (len(sel) - 1, sum(k[1] - k[0] for p in sel for k in p) / 1e6))
n = len(sel)
for lo, hi in ((1, min(20, n)), (20, min(50, n)), (50, n)):
    if hi <= lo:
        continue
    agg = collections.OrderedDict()
    for p in sel[lo:hi]:
        for k in p:
            a = agg.setdefault(short(k[2]), [0, 0])
            a[0] += k[1] - k[0]
            a[1] = max(a[1], k[3])
    tot = sum(v[0] for v in agg.values())
    print("--- passes %d .. %d: %.0f us of kernels per pass" % (lo, hi - 1, tot / 1e3 / (hi - lo)))
    for name, v in sorted(agg.items(), key=lambda x: -x[1][0])[:18]:
        print("   %-46s %7.1f us / pass   (workgroups launched: %d)" % (name, v[0] / 1e3 / (hi - lo), v[1]))

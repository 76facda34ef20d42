# Run on the GPU box: per-kernel time of one adaptive-window search (tools/p2_probe.py) under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/p2k -o p2 -- python3 $GRAFT_REPO_ROOT/tools/p2_probe.py 50000000 ${1:-1270} 1 ${2:-dev} > /tmp/p2k.log 2>&1
grep -E "P2:|round|ecal_detect" /tmp/p2k.log | tail -20
python3 - <<'PY'
import csv, os
rows = list(csv.reader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/p2k/p2_kernel_stats.csv")))
tot = sum(float(r[2]) for r in rows[1:] if "ecal" in r[0])
print("ecal kernels %.1f ms" % (tot / 1e6))
for r in rows[1:]:
    if "ecal" in r[0] and float(r[2]) / 1e6 > float(os.environ.get("P2K_MIN_MS", "0.8")):
        print("%-50s calls %4s total_ms %7.2f avg_us %8.1f" % (r[0].split("(")[0].replace("void ", "")[:50], r[1], float(r[2]) / 1e6, float(r[3]) / 1e3))
PY

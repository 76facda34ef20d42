cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/e2e_tr
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e_tr -- python3 $R/tools/prof_e2e.py > /tmp/e2e_tr.log 2>&1
f=$(find /tmp/e2e_tr -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r["Name"]
    if int(r["Calls"]) <= 8 and float(r["TotalDurationNs"]) > 2e5 or any(k in n for k in ("rectify","pnp","associate","normal_eq")):
        print("%-70s calls %5s total_ms %8.2f avg_us %9.1f" % (n.split("(")[0][-70:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY

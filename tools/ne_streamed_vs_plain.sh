#!/bin/bash
# kernel durations of normal_eq_kernel inside ecal_solver_solve: streamed evaluations (NeProgress: groups delivered to the host while the
# kernel runs) against plain ones (ECAL_FORCE=solver_no_stream), from one kernel trace of tools/solver_probe.py (modes stream, plain)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ne_tr
PROBE_MODES=2 rocprofv3 --kernel-trace --output-format csv -d /tmp/ne_tr -- python3 $R/tools/solver_probe.py 50000000 8 2 > /tmp/ne_tr.log 2>&1
f=$(find /tmp/ne_tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "normal_eq_kernel<false, true" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(len(d), "launches; durations in ms, in order:")
print(" ".join("%.3f" % x for x in d))
PY
grep "it/s" /tmp/ne_tr.log

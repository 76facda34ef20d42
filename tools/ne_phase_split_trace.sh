#!/bin/bash
# the same phase split (tools/ne_phase_split.sh) read from a kernel trace: per-variant average duration of normal_eq_kernel
cd /tmp 2>/dev/null; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/eventcalib_amd/libecal.so /tmp/libecal_orig.so
for v in ${NE_VARIANTS:-full SKIP_P1 SKIP_P2 P2_NOLDS P2_NOFMA}; do
  [ "$v" != "full" ] && cp $R/ab_libs/libecal_$v.so $R/eventcalib_amd/libecal.so
  rm -rf /tmp/ne_trace_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ne_trace_$v -- python3 $R/tools/ne_kernel_time.py > /tmp/ne_trace_$v.log 2>&1
  echo "== $v"
  f=$(find /tmp/ne_trace_$v -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'normal_eq' in r['Name']: print(r['Name'][:60], 'calls', r['Calls'], 'average ns', r['AverageNs'])
"
  grep "role" /tmp/ne_trace_$v.log | tail -2
  cp /tmp/libecal_orig.so $R/eventcalib_amd/libecal.so
done

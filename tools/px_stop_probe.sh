#!/bin/bash
# Run on the GPU box: per-phase instruction counts of the pixel DBSCAN kernel from -DECAL_PX_STOP=k builds
# (ab_libs/libecal_stop<k>.so, made by tools/build_px_stop.sh here).  Output: gpurun_out/<tag>/px_stop.txt
set -u
TAG=${1:-r01x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KERNEL=${PX_KERNEL:-dbscan_pixel_kernel}   # PX_KERNEL=extract_kernel PX_PREFIX=det ECAL_PROBE_DETECT=1 for the extraction kernel
PREFIX=${PX_PREFIX:-stop}
for k in ${PX_STOPS:-1 2 3 4 5 6 full}; do
  lib=$ROOT/ab_libs/libecal_$PREFIX$k.so
  [ "$k" = full ] && lib=$ROOT/eventcalib_amd/libecal.so
  export ECAL_LIB=$lib
  rm -rf $OUT/p
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/p -- python3 $ROOT/tools/px_stop_probe.py > $OUT/log_$k.txt 2>&1
  python3 - $OUT/p $k $KERNEL >> $OUT/px_stop.txt <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(float); disp = set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[3] in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
L = max(len(disp), 1)
w = acc["SQ_WAVES"] / L or 1
print("stop", sys.argv[2], "launches", L, " per wave:", {k: round(v / L / w, 1) for k, v in sorted(acc.items())})
PY
done
rm -rf $OUT/p
cat $OUT/px_stop.txt

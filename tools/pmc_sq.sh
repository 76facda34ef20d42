#!/bin/bash
# gpurun -- 'bash tools/pmc_sq.sh <tag>': SQ counter passes of a 10 M-event bench run; summary in gpurun_out/<tag>_sq.txt
set -u
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--events 10000000 --steps 3 --warmup 1 --cpu-sample 0 --solver-iters 0 --p2-pieces 0 --no-h2d"
: > $ROOT/gpurun_out/${TAG}_sq.txt
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py $ARGS > $OUT/log$i.txt 2>&1
  F=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  python3 $ROOT/tools/pmc_summary.py $F ecal >> $ROOT/gpurun_out/${TAG}_sq.txt
done
rm -rf $OUT
cat $ROOT/gpurun_out/${TAG}_sq.txt

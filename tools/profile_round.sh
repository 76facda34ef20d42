#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round.sh r01c'): kernel-trace stats of the default
# bench.py plus two separate PMC passes (FETCH_SIZE, WRITE_SIZE); summaries land in gpurun_out/<tag>/.
set -u
TAG=${1:-r01x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --cpu-sample 0 --solver-cpu-sample 0 --p2-pieces 0 --no-h2d --ingest-events 0 --e2e-events 0 --no-fixed-cost"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/bench.py $ARGS > $OUT/bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/bench_write.log 2>&1
# the same pass on an eighth of the stream (what one rank of eight gets under --gpus 8): which kernels do not shrink with it
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt8 -- python3 $ROOT/bench.py $ARGS --events 6250000 --solver-iters 0 --calib-views 0 > $OUT/bench_kt8.log 2>&1
cd $ROOT
KS8=$(find $OUT/kt8 -name '*kernel_stats.csv' | head -1)
grep -E 'Name|ecal::' $KS8 > $OUT/${TAG}_kernel_stats_ecal_eighth_of_the_stream.csv
rm -rf $OUT/kt8
KS=$(find $OUT/kt -name '*kernel_stats.csv' | head -1)
cp $KS $OUT/${TAG}_kernel_stats.csv
grep -E 'Name|ecal::' $KS > $OUT/${TAG}_kernel_stats_ecal.csv
F=$(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py $F $W $OUT/${TAG}_traffic.json 50000000
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_write
tail -1 $OUT/bench_kt.log | cut -c1-400

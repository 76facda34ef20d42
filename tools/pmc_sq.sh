#!/bin/bash
# Run on the GPU box (gpurun -- 'timeout 900 bash tools/pmc_sq.sh r01f'): two SQ counter passes over the default
# M1 bench (no other legs); per-kernel sums land in gpurun_out/<tag>/<tag>_sq.json.
set -u
TAG=${1:-r01x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=${PMC_ARGS:-"--steps 4 --warmup 1 --cpu-sample 0 --solver-cpu-sample 0 --p2-pieces 0 --no-h2d --ingest-events 0 --calib-views 0 --solver-iters 0"}
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
B="SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES"
rocprofv3 --pmc $A --output-format csv -d $OUT/pa -- python3 $ROOT/bench.py $ARGS > $OUT/bench_pa.log 2>&1
rocprofv3 --pmc $B --output-format csv -d $OUT/pb -- python3 $ROOT/bench.py $ARGS > $OUT/bench_pb.log 2>&1
cd $ROOT
python3 - $OUT $TAG <<'PY'
import sys, csv, glob, json, collections
out, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for d in ("pa", "pb"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "ecal::" not in k: continue
            res[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
js = {k: dict(launches=len(cnt[k]) // 2 or len(cnt[k]), **{c: v for c, v in sorted(res[k].items())}) for k in res}
json.dump(js, open(f"{out}/{tag}_sq.json", "w"), indent=1)
print(json.dumps(js, indent=1)[:3000])
PY
rm -rf $OUT/pa $OUT/pb

#!/bin/bash
# gpurun -- 'bash tools/occupancy_probe.sh': DBSCAN stage time vs workgroups per CU (LDS padding of tier 0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for PAD in 0 14000 30000; do
  ECAL_DBSCAN_LDS_PAD=$PAD python3 $ROOT/bench.py --events 20000000 --steps 5 --warmup 1 --cpu-sample 0 --solver-iters 0 --p2-pieces 0 --no-h2d 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pad $PAD', d['value'], d['roofline']['stage_ms'])"
done

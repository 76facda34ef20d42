#!/bin/bash
# DBSCAN pixel kernel time vs workgroups per CU (ECAL_DBSCAN_LDS_PAD bytes of extra dynamic LDS): is it
# latency bound (time ~ 1 / workgroups per CU) or issue bound (time flat)?
for pad in 0 2200 8000 16000 30000 60000; do
  echo -n "pad $pad: "
  ECAL_DBSCAN_LDS_PAD=$pad python bench.py --steps 5 --warmup 2 --cpu-sample 0 --solver-iters 0 --p2-pieces 0 --no-h2d --calib-views 0 --ingest-events 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['stage_ms'])"
done

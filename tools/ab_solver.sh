#!/bin/bash
# A/B of the solver leg with two builds of libecal.so
for lib in "$@"; do
  cp eventcalib_amd/libecal.so /tmp/libecal_orig.so
  [ "$lib" != "cur" ] && cp "$lib" eventcalib_amd/libecal.so
  echo -n "$lib: "
  timeout 200 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --solver-cpu-sample 0 --p2-pieces 0 --no-h2d --calib-views 0 --ingest-events 0 --e2e-events 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['solver']; print(d['value'], 'it/s', d['iterations'], 'it', 'eval', d['seconds_evaluate'], 'solve', d['seconds_linear_solve_host'], 'total', d['seconds'])"
  cp /tmp/libecal_orig.so eventcalib_amd/libecal.so
done

#!/bin/bash
# A/B: bench stage times with two builds of libecal.so (the variant is swapped in by path)
for lib in "$@"; do
  cp eventcalib_amd/libecal.so /tmp/libecal_orig.so
  [ "$lib" != "cur" ] && cp "$lib" eventcalib_amd/libecal.so
  echo -n "$lib: "
  timeout 120 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --solver-iters 0 --p2-pieces 0 --no-h2d --calib-views 0 --ingest-events 0 --e2e-events 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['stage_ms'])"
  cp /tmp/libecal_orig.so eventcalib_amd/libecal.so
done

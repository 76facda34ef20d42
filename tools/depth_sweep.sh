#!/bin/bash
# Run on the GPU box: the shared-map keyframe search under different window slots per piece (ECAL_ADAPTIVE_SHAPE depth=; the result
# does not depend on it).  usage: bash tools/depth_sweep.sh "<pieces>:<depth> <depth> ..." ...   (depth 0 = the library's choice)
for rep in 1 2; do
for spec in "$@"; do
  p=${spec%%:*}
  for d in ${spec#*:}; do
    if [ "$d" = 0 ]; then unset ECAL_ADAPTIVE_SHAPE; else export ECAL_ADAPTIVE_SHAPE=depth=$d; fi
    echo -n "pieces $p depth=$d: "
    python tools/p2_probe.py 50000000 $p 1 shared 2>&1 | tail -2 | tr '\n' ' ' | awk '{printf "%s %s s %s passes\n", $3, $11, $13}'
  done
done; done

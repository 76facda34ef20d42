#!/bin/bash
# A/B of the shared-map keyframe search between the library in the tree and other builds of it (ab_libs/*.so), same box, alternating
for rep in 1 2 3; do
for lib in cur "$@"; do
  echo -n "$lib: "
  for p in 1270 4096; do
    if [ "$lib" = cur ]; then unset ECAL_AB_LIB; else export ECAL_AB_LIB=$lib; fi
    python tools/p2_probe.py 50000000 $p 1 shared 2>&1 | tail -2 | tr '\n' ' ' | awk '{printf "%s %s pieces %s s | ", $3, $9, $11}'; done; echo
done; done

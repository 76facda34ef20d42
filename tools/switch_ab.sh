#!/bin/bash
# A/B of one "set or not set" switch on the shared-map search, same box, alternating:  bash tools/switch_ab.sh ECAL_ADAPTIVE_VERIFY_KERNELS
sw=$1
for rep in 1 2 3; do
for t in unset set; do
  if [ $t = set ]; then export $sw=1; else unset $sw; fi
  echo -n "$sw $t: "
  for p in 1270 4096; do python tools/p2_probe.py 50000000 $p 1 shared 2>&1 | tail -2 | tr '\n' ' ' | awk '{printf "%s %s pieces %s s | ", $3, $9, $11}'; done; echo
done; done

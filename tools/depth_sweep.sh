#!/bin/bash
# the shared-map search against the window slots per piece and pass (ECAL_ADAPTIVE_SHAPE depth=)
for rep in 1 2; do
for d in ${DEPTHS:-4 5 6 8}; do
  echo -n "slots per piece $d: "
  for p in 1270 4096; do ECAL_ADAPTIVE_SHAPE=depth=$d python tools/p2_probe.py 50000000 $p 1 shared 2>&1 | tail -1 | awk '{printf "%s pieces %s s | ", $6, $8}'; done; echo
done; done

#!/bin/bash
# the shared-map search against ECAL_ADAPTIVE_LIVE_FLOOR (the slots the pieces at work share when they are few)
for fl in ${FLOORS:-1024 2048 4096}; do
  echo -n "live floor $fl: "
  for p in 1270 4096; do ECAL_ADAPTIVE_LIVE_FLOOR=$fl python tools/p2_probe.py 50000000 $p 1 shared 2>&1 | tail -1 | awk '{printf "%s pieces %s s | ", $6, $8}'; done; echo
done

#!/bin/bash
# A/B of the tree of chains (ECAL_ADAPTIVE_SHAPE=tree=0: chains and side chains only), both gates, same box, alternating
for rep in 1 2 3; do
for t in on off; do
  if [ $t = off ]; then export ECAL_ADAPTIVE_SHAPE=tree=0; else unset ECAL_ADAPTIVE_SHAPE; fi
  echo -n "tree $t: "
  for p in 1270 4096; do for g in shared dev; do python tools/p2_probe.py 50000000 $p 1 $g 2>&1 | tail -1 | awk -v g=$g '{printf "%s %s pieces %s s (%s passes) | ", g, $6, $8, $10}'; done; done; echo
done; done

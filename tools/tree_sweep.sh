#!/bin/bash
# the shared-map search against the tree of chains (ECAL_ADAPTIVE_TREE = main | len << 8 | c1 << 16 | c2 << 20 | c3 << 24 | from << 28) and the live floor
enc() { echo $(( $1 | ($2 << 8) | ($3 << 16) | ($4 << 20) | ($5 << 24) | ($6 << 28) )); }
for fl in ${FLOORS:-1024 1536}; do
for cfg in ${CFGS:-"8 8 5 4 3 1" "8 8 4 3 2 1" "8 6 4 3 2 1" "8 8 3 3 3 1" "6 6 4 3 2 1" "8 8 5 4 0 1" "10 8 5 4 3 1" "8 8 6 2 2 1"}; do
  set -- $cfg
  v=$(enc $1 $2 $3 $4 $5 $6)
  echo -n "floor $fl main $1 len $2 c $3 $4 $5 from $6: "
  for p in 1270 4096; do ECAL_ADAPTIVE_LIVE_FLOOR=$fl ECAL_ADAPTIVE_TREE=$v python tools/p2_probe.py 50000000 $p 1 shared 2>&1 | tail -1 | awk '{printf "%s pieces %s s | ", $6, $8}'; done; echo
done; done

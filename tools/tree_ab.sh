#!/bin/bash
# Run on the GPU box: the shared-map search at 1270 pieces under other shapes of the tree of chains (ECAL_ADAPTIVE_SHAPE tree=<word>:
# main | len << 8 | c1 << 16 | c2 << 20 | c3 << 24 | from << 28; the result does not depend on it) and other builds (ab_libs/*.so)
w() { echo $(( $1 | ($2 << 8) | ($3 << 16) | ($4 << 20) | ($5 << 24) | ($6 << 28) )); }
for rep in 1 2; do
for spec in "default" ${TREES:-"tree=$(w 8 7 5 4 3 1)" "tree=$(w 8 7 5 4 4 1)" "tree=$(w 8 6 5 4 3 1)" "tree=$(w 7 7 5 4 3 1)" "tree=$(w 8 7 6 4 3 1)" "tree=$(w 8 7 5 5 3 1)" "tree=$(w 8 6 5 4 4 1)" "tree=$(w 8 7 4 4 4 1)"} "$@"; do
  unset ECAL_ADAPTIVE_SHAPE ECAL_AB_LIB
  case "$spec" in default) ;; tree=*) export ECAL_ADAPTIVE_SHAPE=$spec ;; *) export ECAL_AB_LIB=$spec ;; esac
  echo -n "$spec: "; python tools/p2_probe.py 50000000 1270 1 shared 2>&1 | tail -2 | tr '\n' ' ' | awk '{printf "%s %s s\n", $3, $11}'
done; done

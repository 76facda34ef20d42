#!/bin/bash
# Phase split of the normal-equations kernel: the launch alone on the benchmark's spline problem with profiling builds of
# ecal_solver.hip (-DECAL_NE_SKIP_P1 / SKIP_P2 / P2_NOLDS / P2_NOFMA, built into ab_libs/libecal_<variant>.so; results of those
# builds are meaningless, only their time is read).
cp eventcalib_amd/libecal.so /tmp/libecal_orig.so
for v in ${NE_VARIANTS:-full SKIP_P1 SKIP_P2 P2_NOLDS P2_NOFMA full}; do
  [ "$v" != "full" ] && cp ab_libs/libecal_$v.so eventcalib_amd/libecal.so
  echo "== $v"
  timeout 300 python tools/ne_kernel_time.py 2>&1 | grep "role" | tail -2
  cp /tmp/libecal_orig.so eventcalib_amd/libecal.so
done

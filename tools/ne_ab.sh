#!/bin/bash
# A/B of the normal-equations launch: the shipped library against ab_libs/libecal_<variant>.so builds of ecal_solver.hip
# (FMA = -DECAL_NE_FMA_TILES, the register-tiled v_fma_f64 Gram accumulation of rounds 2-5)
cp eventcalib_amd/libecal.so /tmp/libecal_orig.so
for v in ${NE_VARIANTS:-cur FMA cur FMA}; do
  [ "$v" != "cur" ] && cp ab_libs/libecal_$v.so eventcalib_amd/libecal.so
  echo "== $v"
  timeout 300 python tools/ne_kernel_time.py 2>&1 | grep "one role" | tail -2
  cp /tmp/libecal_orig.so eventcalib_amd/libecal.so
done

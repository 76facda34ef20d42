#!/bin/bash
# Build container: stop-after-phase debug builds of one translation unit, for tools/px_stop_probe.sh.
#   tools/build_stop_libs.sh <tu> <macro> <prefix> "<k ...>" [extra flags]
#   e.g. tools/build_stop_libs.sh ecal_events ECAL_RO_STOP ro "1 2 3 4" -DECAL_RO_WAVES=6
# Output: ab_libs/libecal_<prefix><k>.so (git-ignored; travels to the GPU box with the snapshot)
set -e
TU=$1; MACRO=$2; PREFIX=$3; STOPS=$4; shift 4
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/eventcalib_amd/csrc
mkdir -p $ROOT/ab_libs
make -s -C $SRC
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -ffp-contract=off"
for k in $STOPS; do
  /opt/rocm/bin/hipcc $FLAGS -D$MACRO=$k "$@" -c $SRC/$TU.hip -o /tmp/stop_${TU}_$k.o
  objs=$(ls $SRC/build/*.o | grep -v "/$TU.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/ab_libs/libecal_$PREFIX$k.so $objs /tmp/stop_${TU}_$k.o -L/opt/rocm/lib -lrccl
  echo "built ab_libs/libecal_$PREFIX$k.so"
done

#!/bin/bash
# A/B builds of libnisqa_hip.so with experiment flags on ONE translation unit:  tools/ab_build.sh NAME UNIT "-DFLAG=1 ..."
# -> ab_libs/NAME.so (all other objects are the regular ones; run `make -C nisqa_amd/csrc` first).  FLAGS "-DNQ_EXPERIMENTAL" compiles
# the unit's phase clock in (csrc/experimental.hpp); such a library loads only with NISQA_ALLOW_DEBUG_LIB=1.  Use with
# NISQA_HIP_LIB=$PWD/ab_libs/NAME.so python bench.py ...
set -e
NAME=$1; UNIT=$2; FLAGS=$3
cd "$(dirname "$0")/../nisqa_amd/csrc"
mkdir -p ../../ab_libs /tmp/nq_ab_$NAME
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -I../../include"
for f in *.hip; do s=${f%.hip}
  if [ $s = $UNIT ]; then /opt/rocm/bin/hipcc $F $FLAGS -c $s.hip -o /tmp/nq_ab_$NAME/$s.o; else cp $s.o /tmp/nq_ab_$NAME/$s.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_libs/$NAME.so /tmp/nq_ab_$NAME/*.o
ls -la ../../ab_libs/$NAME.so

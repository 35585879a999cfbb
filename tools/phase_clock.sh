#!/bin/bash
# Builds ab_libs/NAME.so (default clock): the library with cnn_bf16.hip compiled -DNQ_EXPERIMENTAL (csrc/experimental.hpp: clock64
# stamps inside cnn_front_bf16_kernel).  Load it with NISQA_ALLOW_DEBUG_LIB=1 (lib.load() refuses instrumented builds otherwise).
# The knock-out flags of rounds 2-5 (-DNQ_KO=..) left the sources in round 6: git show c223882:nisqa_amd/csrc/conv_bf16.hpp
set -e
NAME=${1:-clock}; FLAGS=$2
cd "$(dirname "$0")/../nisqa_amd/csrc"
mkdir -p ../../ab_libs /tmp/nq_$NAME
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -I../../include"
for f in *.hip; do s=${f%.hip}
  if [ $s = cnn_bf16 ]; then /opt/rocm/bin/hipcc $F -DNQ_EXPERIMENTAL $FLAGS -c $s.hip -o /tmp/nq_$NAME/$s.o; else cp $s.o /tmp/nq_$NAME/$s.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_libs/$NAME.so /tmp/nq_$NAME/*.o
ls -la ../../ab_libs/$NAME.so

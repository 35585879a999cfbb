#!/bin/bash
# Builds ab_libs/clock.so: the library with -DNQ_PHASE_CLOCK (clock64 stamps inside cnn_front_bf16_kernel).
set -e
cd "$(dirname "$0")/../nisqa_amd/csrc"
mkdir -p ../../ab_libs /tmp/nq_clock
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -I../../include"
for s in api mel cnn cnn_bf16 cnn_std cnn_std_bf16 lstm td td_bf16 train; do
  if [ $s = cnn_bf16 ]; then /opt/rocm/bin/hipcc $F -DNQ_PHASE_CLOCK -c $s.hip -o /tmp/nq_clock/$s.o; else cp $s.o /tmp/nq_clock/$s.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_libs/clock.so /tmp/nq_clock/*.o
ls -la ../../ab_libs/clock.so

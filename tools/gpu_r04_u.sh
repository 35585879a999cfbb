#!/bin/bash
O=gpurun_out/r04u; mkdir -p $O
for rep in 1 2; do
for L in base x6nofence; do
  if [ $L = base ]; then E="X=1"; else E="NISQA_HIP_LIB=$PWD/ab_libs/$L.so"; fi
  echo "$L: $(env $E NISQA_HIP_TRAIN_PRECISION=bf16x6 python tools/bench_train.py 32 30 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])") ms"
done; done | tee $O/ab_x6_train_fence.txt

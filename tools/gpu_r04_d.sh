#!/bin/bash
O=gpurun_out/r04d; mkdir -p $O
./ab_libs/klm4 200 > $O/klm4.txt 2>&1; cat $O/klm4.txt
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > $O/pytest_train.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_train.log
for P in f32 mixed bf16x3; do NISQA_HIP_TRAIN_PRECISION=$P python tools/bench_train.py 32 20 2>/dev/null | tail -1; done > $O/train_bench.json; cat $O/train_bench.json

#!/bin/bash
O=gpurun_out/r04s; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -k "fp32_weight_gradient or fp32_convolutions or segment_resident_convolutions" 2>&1 | tail -5 | tee $O/pytest_kernels.txt
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -s -k "bf16x6" 2>&1 | grep -E "bf16x6|passed|failed|Error|assert" | tail -20 | tee $O/pytest_steps.txt
for P in f32 bf16x6 mixed f32 bf16x6; do NISQA_HIP_TRAIN_PRECISION=$P python tools/bench_train.py 32 30 2>/dev/null | tail -1; done | tee $O/train_bench.txt

#!/bin/bash
# round 4, first GPU call: the new tests (configs[4]-size training parity, live reference loop) + a training-step baseline
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -q -s -k "configs4 or live_reference" > $O/pytest_new.log 2>&1; echo "pytest rc $?"
grep -E "configs\[4\]|cfg5|live reference|passed|failed|Error" $O/pytest_new.log | tail -30
for P in f32 mixed bf16x3; do NISQA_HIP_TRAIN_PRECISION=$P python tools/bench_train.py 32 20 2>/dev/null | tail -1; done > $O/train_bench.json; cat $O/train_bench.json
rm -rf /tmp/ks_train
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_train -o ks -- python bench.py --no-cpu-baseline --leg train --steps 8 > /tmp/ks_train.log 2>&1
cp /tmp/ks_train/ks_kernel_stats.csv $O/train_kernel_stats_start.csv; head -40 $O/train_kernel_stats_start.csv

#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
NQ_PRECISION=bf16x6 NISQA_HIP_LIB=$PWD/ab_libs/clock6.so python tools/phase_clock.py 2>&1 | tail -14 | tee $O/x6_phase_clock.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16x6 or rounding_error" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "bf16x6 or fp32_convolutions" 2>&1 | tail -2
stage() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['stage_ms']['cnn_front'])"; }
for rep in 1 2; do
  echo "== x6: $(python bench.py --no-cpu-baseline --no-extras --precision bf16x6 --steps 100 --warmup 20 2>/dev/null | stage)"
done

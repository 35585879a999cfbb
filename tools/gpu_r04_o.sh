#!/bin/bash
O=gpurun_out/r04o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16x6 or rounding_error or inner_operator" -s 2>&1 | grep -E "float64|passed|failed|Error|error|stage max|assert" | tail -20 | tee $O/pytest_x6.txt
stage() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['frac'])"; }
for P in bf16x6 f32 bf16x6 bf16x3; do
  echo "== $P: $(python bench.py --no-cpu-baseline --no-extras --precision $P --steps 100 --warmup 20 2>/dev/null | stage)"
done 2>&1 | tee $O/bench_x6.txt

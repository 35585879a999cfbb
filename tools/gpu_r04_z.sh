#!/bin/bash
O=gpurun_out/r04z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "tts" 2>&1 | grep -E "bf16x6|passed|failed|Error|assert" | tail -12 | tee $O/pytest_tts_x6.txt
for P in bf16x3 bf16x6 f32; do
  echo "$P: $(NISQA_HIP_PRECISION=$P python bench.py --no-cpu-baseline --no-extras --leg tts --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_job'], d['stage_ms'])")"
done | tee $O/tts_precisions.txt

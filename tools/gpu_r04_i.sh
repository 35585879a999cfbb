#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tts or config4" > $O/pytest_tts.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_tts.log
python bench.py --leg tts --no-cpu-baseline --steps 12 2>/dev/null | tail -1 > $O/tts.json
python - <<PY
import json
d=json.loads(open('$O/tts.json').read())
print(d['value'], d['ms_per_job'], d['value_2_streams'], d['stage_ms'], d['roofline']['frac'], d['lstm']['us_per_step'])
PY

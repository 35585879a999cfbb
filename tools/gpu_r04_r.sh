#!/bin/bash
O=gpurun_out/r04r; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; tail -c 600 $O/bench_driver_form.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04r/bench_driver_form.json') if l.startswith('{')][-1])
print('value',d['value'],'f32',d.get('value_f32'),'x6',d.get('value_bf16x6'), d['stage_ms'])
print('x6', json.dumps(d.get('bf16x6'))[:900])
for k,v in d['side'].items(): print(k, v.get('value'), v.get('unit'), v.get('ms_per_step'))
PY

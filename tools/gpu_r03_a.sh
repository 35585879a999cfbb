#!/bin/bash
# round 3, first GPU call: whole GPU suite + the full default bench line (all side legs) + the driver's short form
O=gpurun_out/r03a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>$O/bench_driver_form.err; echo "bench rc $?"; tail -3 $O/bench_driver_form.err
python - <<PY
import json
d = json.loads(open('$O/bench_driver_form.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'])
print('f32', d.get('value_f32'), (d.get('f32') or {}).get('roofline', {}).get('frac'))
print('steady', d.get('steady'))
for k, v in (d.get('side') or {}).items():
    print(k, json.dumps(v)[:1500])
print('cpu', d.get('cpu_baseline', {}).get('value'))
PY

#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>$O/bench_driver_form.err ) 2>&1 | grep real; echo "bench rc $?"; tail -3 $O/bench_driver_form.err
python - <<PY
import json
d = json.loads(open('$O/bench_driver_form.json').read().strip().split('\n')[-1])
print('value', d['value'], 'ms', d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'])
print('f32', d.get('value_f32'))
print('cpu', json.dumps(d.get('cpu_baseline'))[:1200])
for k, v in (d.get('side') or {}).items():
    print(k, json.dumps(v)[:1800])
PY

#!/bin/bash
O=gpurun_out/r04y; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
S=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; E=$(date +%s); echo "default bench seconds: $((E-S)); lines: $(grep -c '^{' $O/bench_default.json)"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04y/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['value_f32'], d['value_bf16x6'], d['steady']['value'], {k:(v.get('value'),v.get('leg_seconds')) for k,v in d['side'].items()})
print(d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
PY

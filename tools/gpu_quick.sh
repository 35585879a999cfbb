#!/bin/bash
# quick GPU check of the predict path: parity tests (subset by default) + bench line
# usage: tools/gpu_quick.sh OUTDIR ["-k expr" for pytest] [bench args...]
O=${1:-gpurun_out/quick}; K=${2:-}; shift; shift
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q $K > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 300 "$@" > $O/bench.json 2>$O/bench.err || tail -5 $O/bench.err
python - <<PY
import json
d = json.loads(open('$O/bench.json').read().strip().split('\n')[-1])
print('clips/s', d['value'], 'ms/step', d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'], 'alt', d.get('alt_precision', {}).get('value'), 'maxdiff', d.get('alt_precision', {}).get('max_abs_diff_vs_primary'), '3 streams', d.get('overlap_3_streams'))
PY

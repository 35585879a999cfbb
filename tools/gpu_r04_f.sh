#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
for cfg in "1 2" "2 3" "2 2" "3 4"; do
  set -- $cfg
  NISQA_LOOP_INFLIGHT=$1 NISQA_LOOP_DEPTH=$2 python bench.py --leg csv --no-cpu-baseline 2>/dev/null | tail -1 > $O/csv_$1_$2.json
  python - <<PY
import json
d=json.loads(open('$O/csv_$1_$2.json').read())
print('inflight $1 depth $2:', d['value'], 'clips/s', d['seconds'], 's', d['roofline'], d['print_s'], d['loop_host_s'])
PY
done

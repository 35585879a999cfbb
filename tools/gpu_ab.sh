#!/bin/bash
# A/B timing of library builds: tools/gpu_ab.sh NAME [NAME ...]  (ab_libs/NAME.so; "base" = the in-tree library)
export NISQA_BENCH_KO=1
for L in "$@"; do
  if [ $L = base ]; then unset NISQA_HIP_LIB; else export NISQA_HIP_LIB=$PWD/ab_libs/$L.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 200 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-10s' % '$L', d['value'], d['stage_ms'])
except Exception as e: print('$L', 'failed', e)"
done

#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O
stage() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['stage_ms']['cnn_front'])"; }
for rep in 1 2 3; do
  echo "== trunc: $(python bench.py --no-cpu-baseline --no-extras --precision bf16x6 --steps 100 --warmup 20 2>/dev/null | stage)"
  echo "== rne  : $(NISQA_HIP_LIB=$PWD/ab_libs/x6rne.so python bench.py --no-cpu-baseline --no-extras --precision bf16x6 --steps 100 --warmup 20 2>/dev/null | stage)"
done 2>&1 | tee $O/ab_x6_split.txt

#!/bin/bash
O=gpurun_out/r04x; mkdir -p $O
stage() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['stage_ms']['cnn_front'])"; }
for rep in 1 2 3; do
  echo "base: $(python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 30 2>/dev/null | stage)"
  echo "sb2 : $(NISQA_HIP_LIB=$PWD/ab_libs/sb2.so python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 30 2>/dev/null | stage)"
done | tee $O/ab_sb2.txt

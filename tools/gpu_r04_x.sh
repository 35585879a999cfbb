#!/bin/bash
O=gpurun_out/r04x; mkdir -p $O
stage() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['stage_ms']['selfatt'], d['stage_ms']['pool'])"; }
for rep in 1 2 3; do
  for P in bf16x3 bf16x6; do
  echo "$P base  : $(python bench.py --no-cpu-baseline --no-extras --precision $P --steps 200 --warmup 30 2>/dev/null | stage)"
  echo "$P vgpr  : $(NISQA_HIP_LIB=$PWD/ab_libs/tdvf.so python bench.py --no-cpu-baseline --no-extras --precision $P --steps 200 --warmup 30 2>/dev/null | stage)"
  done
done | tee $O/ab_td_vgprform.txt
NISQA_HIP_LIB=$PWD/ab_libs/tdvf.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2

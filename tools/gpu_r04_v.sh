#!/bin/bash
O=gpurun_out/r04v; mkdir -p $O
for rep in 1 2; do
for P in bf16x3 bf16x6; do
  echo "$P: $(NISQA_HIP_PRECISION=$P timeout 600 python bench.py --workload predict_csv --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('seconds'), (d.get('roofline') or d.get('pcie') or {}))")"
done; done | tee $O/predict_csv_precision.txt

#!/bin/bash
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -s -k "fused_self_attention" > $O/pytest_fused.log 2>&1; echo "pytest rc $?"
grep -E "fused vs|passed|failed|Error|error" $O/pytest_fused.log | tail -20
rm -rf /tmp/ks_train
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_train -o ks -- python bench.py --no-cpu-baseline --leg train --steps 8 > /tmp/ks_train.log 2>&1
cp /tmp/ks_train/ks_kernel_stats.csv $O/train_kernel_stats_fused_td.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/train_kernel_stats_fused_td.csv')))
n=[int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']][0]
print('steps',n,'launches/step',sum(int(r['Calls']) for r in rows)/n,'kernel ms/step',sum(float(r['TotalDurationNs']) for r in rows)/n/1e6)
for r in rows[:60]:
    print('%-64s calls/step %5.1f avg %8.1f us  per step %7.1f us'%(r['Name'][:64], int(r['Calls'])/n, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/n/1e3))
PY

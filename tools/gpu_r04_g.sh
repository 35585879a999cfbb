#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "segment_resident_fp32_convolutions" > $O/pytest_f32c.log 2>&1; echo "pytest rc $?"
tail -12 $O/pytest_f32c.log
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q > $O/pytest_train.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_train.log
for P in f32 mixed bf16x3; do NISQA_HIP_TRAIN_PRECISION=$P python tools/bench_train.py 32 20 2>/dev/null | tail -1; done > $O/train_bench.json; cat $O/train_bench.json
for P in f32 mixed; do
rm -rf /tmp/ks_train
NISQA_HIP_TRAIN_PRECISION=$P rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_train -o ks -- python tools/bench_train.py 32 20 > /tmp/ks_train.log 2>&1
cp /tmp/ks_train/ks_kernel_stats.csv $O/train_kernel_stats_$P.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/train_kernel_stats_$P.csv')))
n=[int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']][0]
print('$P steps',n,'launches/step',sum(int(r['Calls']) for r in rows)/n,'kernel ms/step',sum(float(r['TotalDurationNs']) for r in rows)/n/1e6)
for r in rows[:22]:
    print('%-84s calls/step %5.1f avg %8.1f us  per step %7.1f us'%(r['Name'][:84], int(r['Calls'])/n, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/n/1e3))
PY
done

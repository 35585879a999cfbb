#!/bin/bash
O=gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "fused_self_attention or fixture or oracle or published or data_parallel" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for P in f32 mixed bf16x3; do NISQA_HIP_TRAIN_PRECISION=$P python tools/bench_train.py 32 30 2>/dev/null | tail -1; done
rm -rf /tmp/ks_train
NISQA_HIP_TRAIN_PRECISION=mixed rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_train -o ks -- python tools/bench_train.py 32 20 > /tmp/ks_train.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/ks_train/ks_kernel_stats.csv')))
n=[int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']][0]
for r in rows:
    if 'tdt_' in r['Name'] or 'gemm_f32_kernel' in r['Name']:
        print('%-60s calls/step %5.1f avg %8.1f us  per step %7.1f us'%(r['Name'][:60], int(r['Calls'])/n, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/n/1e3))
PY

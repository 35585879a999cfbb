#!/bin/bash
O=gpurun_out/r04t; mkdir -p $O
export TMPDIR=/tmp
for P in bf16x6 f32; do
rm -rf /tmp/ks_$P
NISQA_HIP_TRAIN_PRECISION=$P rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$P -o ks -- python tools/bench_train.py 32 20 > /tmp/ks_$P.log 2>&1
cp /tmp/ks_$P/ks_kernel_stats.csv $O/train_${P}_kernel_stats.csv
done
python - <<'PY'
import csv
for P in ('bf16x6','f32'):
    print('==',P)
    tot=0
    for r in csv.DictReader(open('gpurun_out/r04t/train_%s_kernel_stats.csv'%P)):
        if 'segconv' in r['Name'] or 'segwgrad' in r['Name']:
            print(r['Name'][:86].ljust(86), r['Calls'], '%.1f'%(float(r['AverageNs'])/1e3)); tot+=float(r['AverageNs'])/1e3
    print('conv kernels total us per step', tot)
PY

#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/ks_tts
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_tts -o ks -- python bench.py --no-cpu-baseline --no-extras --leg tts --steps 40 > /tmp/ks_tts.log 2>&1
cp /tmp/ks_tts/ks_kernel_stats.csv $O/r04_tts_kernel_stats.csv
grep '^{' /tmp/ks_tts.log | tail -1 > $O/r04_tts_leg_under_rocprof.json
python - <<PY
import csv, json
rows=list(csv.DictReader(open('$O/r04_tts_kernel_stats.csv')))
d=json.loads(open('$O/r04_tts_leg_under_rocprof.json').read())
for r in rows:
    if 'cnn_std_bf16' in r['Name'] or 'lstm_dir' in r['Name'] or 'mel_frame' in r['Name']:
        print('%-40s calls %4s avg %9.1f us' % (r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3))
print('events: cnn avg launch ms', d['roofline']['avg_launch_ms'], 'stage_ms', d['stage_ms'], 'value', d['value'])
PY

#!/bin/bash
# Round profiles (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats and PMC passes for the
# predict path (both precisions) and the training step.  Output: gpurun_out/prof_rNN/ -> copy into profiles/.
R=${1:-r01}
O=gpurun_out/prof_$R
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/${R}_bench_bf16x3.json 2>/dev/null
python bench.py --precision f32 > $O/${R}_bench_f32.json 2>/dev/null
for P in bf16x3 f32; do
  rm -rf /tmp/ks_$P
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$P -o ks -- python bench.py --no-cpu-baseline --no-extras --precision $P > /tmp/ks_$P.log 2>&1
  cp /tmp/ks_$P/ks_kernel_stats.csv $O/${R}_bench_${P}_kernel_stats.csv
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -o pmc -- python bench.py --no-cpu-baseline --no-extras --steps 5 > /tmp/pmc_$i.log 2>&1
done
python - > $O/${R}_pmc_bench_bf16x3.txt <<'PY'
import glob, pandas as pd
fr = [pd.read_csv(f) for f in glob.glob('/tmp/pmc_*/pmc_counter_collection.csv')]
t = pd.concat(fr)
t = t[~t.Kernel_Name.str.contains('at::|rocclr')]
t['k'] = t.Kernel_Name.str.split('(').str[0]
pd.set_option('display.width', 250)
print('mean counter value per launch (default bf16x3 path, python bench.py --no-cpu-baseline --no-extras --steps 5)')
print(t.groupby(['k', 'Counter_Name'])['Counter_Value'].mean().unstack(1).to_string())
PY
python tools/bench_train.py 32 20 2>/dev/null | tail -1 > $O/${R}_train_bench.json
rm -rf /tmp/ks_train
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_train -o ks -- python tools/bench_train.py 32 5 > /tmp/ks_train.log 2>&1
cp /tmp/ks_train/ks_kernel_stats.csv $O/${R}_train_kernel_stats.csv
python tools/bench_extra.py 2>/dev/null | tail -1 > $O/${R}_side_tts_pcie.json
python tools/bench_ingest.py 1024 8,32 2>/dev/null | tail -1 > $O/${R}_side_ingest.json
# optional extras, when built beforehand (tools/phase_clock.sh; hipcc -o ab_libs/issue tools/micro/issue.hip)
[ -f ab_libs/clock.so ] && NISQA_HIP_LIB=$PWD/ab_libs/clock.so python tools/phase_clock.py 2>/dev/null | grep -v amdgpu.ids > $O/${R}_cnn_phase_clock.txt
[ -x ab_libs/issue ] && ./ab_libs/issue > $O/${R}_micro_issue.txt
ls -la $O

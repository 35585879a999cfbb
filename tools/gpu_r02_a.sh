#!/bin/bash
# round-2 GPU call A: parity tests incl. the new config fixtures, co-issue / co-run micro-benchmarks, priority A/B builds,
# LDS-conflict PMC pass of the final CNN kernel
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 120 ./ab_libs/issue2 > $O/r02_micro_issue2.txt 2>&1; echo "issue2 rc $?"
timeout 120 ./ab_libs/corun3 > $O/r02_micro_corun3.txt 2>&1; echo "corun3 rc $?"
timeout 120 ./ab_libs/corun2 > $O/r02_micro_corun2.txt 2>&1; echo "corun2 rc $?"
for L in base prio1 prio2 prio3; do
  if [ $L = base ]; then unset NISQA_HIP_LIB; else export NISQA_HIP_LIB=$PWD/ab_libs/$L.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 300 > $O/bench_$L.json 2>$O/bench_$L.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_$L.json').read().strip().split('\n')[-1])
    print('$L', d['value'], d['stage_ms'])
except Exception as e:
    print('$L failed', e)
PY
done
unset NISQA_HIP_LIB
rm -rf /tmp/pmc_lds
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/pmc_lds -o pmc -- python bench.py --no-cpu-baseline --no-extras --steps 5 > /tmp/pmc_lds.log 2>&1
python tools/pmc_summary.py /tmp/pmc_lds > $O/r02_pmc_sq_all_kernels.txt 2>&1 || python - <<'PY' > gpurun_out/r02a/r02_pmc_sq_all_kernels.txt
import glob, pandas as pd
fr = [pd.read_csv(f) for f in glob.glob('/tmp/pmc_lds/**/*counter_collection.csv', recursive=True)]
t = pd.concat(fr)
t = t[~t.Kernel_Name.str.contains('at::|rocclr')]
t['k'] = t.Kernel_Name.str.split('(').str[0]
pd.set_option('display.width', 250)
print(t.groupby(['k', 'Counter_Name'])['Counter_Value'].mean().unstack(1).to_string())
PY
cat $O/r02_pmc_sq_all_kernels.txt | head -40
cat $O/r02_micro_issue2.txt
cat $O/r02_micro_corun3.txt

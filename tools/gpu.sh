#!/bin/bash
# One parameterised GPU recipe (replaces the per-experiment tools/gpu_rNN_x.sh files of rounds 2-4).  Runs on the GPU box from the
# repo root:   gpurun --timeout 900 -- 'tools/gpu.sh TAG step [step ...]'      (output: gpurun_out/TAG/)
# steps:
#   tests[:K]         pytest -m gpu (optionally -k K), -x
#   parity[:K]        tests/test_gpu_parity.py only
#   train[:K]         tests/test_gpu_train.py only
#   bench[:ARGS]      python bench.py ARGS (commas in ARGS become spaces), JSON line -> bench_<n>.json, one-line summary printed
#   ab:NAME[,NAME]    A/B of ab_libs/NAME.so builds against the in-tree library (tools/gpu_ab.sh)
#   exe:PATH[,ARGS]   run a microbenchmark binary (ab_libs/...), stdout -> exe_<n>.txt
#   py:SCRIPT[,ARGS]  python SCRIPT ARGS, stdout -> py_<n>.txt
#   stats:ARGS        rocprofv3 --kernel-trace --stats of bench.py ARGS -> stats_<n>_kernel_stats.csv
#   smoke             __graft_entry__.smoke()
TAG=${1:?tag}; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
n=0
for step in "$@"; do
  n=$((n+1)); kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  echo "=== [$n] $step"
  case $kind in
    tests|parity|train)
      files="tests"; [ $kind = parity ] && files=tests/test_gpu_parity.py; [ $kind = train ] && files=tests/test_gpu_train.py
      if [ -n "$arg" ]; then timeout 1500 python -m pytest $files -m gpu -x -q -rP -k "$arg" > $O/pytest_$n.log 2>&1; else timeout 1500 python -m pytest $files -m gpu -x -q -rP > $O/pytest_$n.log 2>&1; fi
      echo "pytest rc $?"; grep -E "max \||float64|live reference" $O/pytest_$n.log | tail -12; tail -6 $O/pytest_$n.log;;
    bench)
      timeout 600 python bench.py ${arg//,/ } > $O/bench_$n.json 2> $O/bench_$n.err || tail -5 $O/bench_$n.err
      python - "$O/bench_$n.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    cfg = d.get('config') if isinstance(d.get('config'), dict) else {}
    print('value', d.get('value'), cfg.get('precision'), 'ms/step', d.get('ms_per_step'), d.get('stage_ms'), 'frac', d.get('roofline', {}).get('frac'))
    if d.get('leg'):
        print(' leg', d['leg'], {k: d.get(k) for k in ('seconds', 'predict_s', 'table_s', 'print_s', 'value_f32', 'value_bf16x6', 'value_f16x4', 'value_f16x3', 'value_bf16x3', 'ms_per_job', 'stage_ms') if k in d},
              {k: v for k, v in d.get('roofline', {}).items() if k in ('achieved', 'link_only_GBps', 'frac_of_link_only', 'frac')}, d.get('loop_host_s'), d.get('lstm'))
    for k in d:
        if k.startswith('value_') and isinstance(d.get(k[6:]), dict):
            print(' ', k, d[k], d[k[6:]].get('stage_ms'), 'vs primary', d[k[6:]].get('max_abs_diff_vs_primary'), 'vs f32', d[k[6:]].get('max_abs_diff_vs_f32'))
    for k, v in d.get('side', {}).items():
        print(' side', k, {q: v.get(q) for q in ('value', 'value_f32', 'value_bf16x6', 'value_f16x4', 'value_f16x3', 'value_bf16x3', 'stage_ms', 'error') if q in v})
except Exception as e:
    print('bench line unreadable:', e)
PY
      ;;
    ab) bash tools/gpu_ab.sh base ${arg//,/ } | tee $O/ab_$n.txt;;
    exe) a=(${arg//,/ }); timeout 300 "${a[@]}" > $O/exe_$n.txt 2>&1; echo "rc $?"; cat $O/exe_$n.txt;;
    py) a=(${arg//,/ }); timeout 900 python "${a[@]}" > $O/py_$n.txt 2>&1; echo "rc $?"; grep -v amdgpu.ids $O/py_$n.txt | tail -40;;
    stats)
      rm -rf /tmp/ks_$n; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$n -o ks -- python $OLDPWD/bench.py ${arg//,/ } > /tmp/ks_$n.log 2>&1)
      cp /tmp/ks_$n/ks_kernel_stats.csv $O/stats_${n}_kernel_stats.csv 2>/dev/null; grep '^{' /tmp/ks_$n.log | tail -1 > $O/stats_${n}_bench.json; head -12 $O/stats_${n}_kernel_stats.csv | cut -c1-160;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4;;
    *) echo "unknown step $step";;
  esac
done

#!/bin/bash
# Round profiles (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats and PMC passes for the
# predict path (every precision form), the nisqa_tts.tar leg and the training step.  Output: gpurun_out/prof_rNN/ -> copy into profiles/.
#   tools/collect_profiles.sh r03
R=${1:-r06}
O=gpurun_out/prof_$R
mkdir -p $O
export TMPDIR=/tmp
# PMC passes first: bench.py reads profiles/rNN_pmc_kernels.json for roofline.traffic / mfma_util of the SAME build.
# Four counter sets x seven workloads (main bf16x6 = the default, f16x4, bf16x3, f32, the tts leg in bf16x6 and f16x4, the training
# leg), each its own rocprofv3 run (counters only: no trace domains next to --pmc).
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD")
dirs=""
w=0
for WL in "--leg main" "--leg main --precision f16x4" "--leg main --precision bf16x3" "--leg main --precision f32" "--leg tts" "--leg tts --precision f16x4" "--leg train"; do
  w=$((w+1)); i=0
  for set in "${SETS[@]}"; do
    i=$((i+1)); d=/tmp/pmcp_${w}_$i; rm -rf $d
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d $d -o pmc -- python $OLDPWD/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 3 $WL > $d.log 2>&1)
    case $w in 5|6) dirs="$dirs tts:=$d";; 7) dirs="$dirs train:=$d";; *) dirs="$dirs $d";; esac
  done
done
python tools/pmc_to_json.py $O/${R}_pmc_kernels.json $dirs > $O/${R}_pmc_kernels.txt 2>&1
cp $O/${R}_pmc_kernels.json profiles/${R}_pmc_kernels.json      # so that the bench lines below carry traffic / mfma_util
# the contract line (default precision bf16x6) in its two forms, then every other precision as the primary of its own run
python bench.py > $O/${R}_bench_bf16x6.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_bf16x6_driver_form.json 2>/dev/null
for P in f16x4 f16x3 bf16x3 f32; do
  python bench.py --precision $P --no-cpu-baseline --no-side --no-extras > $O/${R}_bench_$P.json 2>/dev/null
done
python bench.py --streams 2 --no-cpu-baseline --no-extras > $O/${R}_bench_bf16x6_2streams.json 2>/dev/null
for P in bf16x6 f16x4 f16x3 bf16x3 f32; do
  rm -rf /tmp/ks_$P
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$P -o ks -- python $OLDPWD/bench.py --no-cpu-baseline --no-extras --precision $P > /tmp/ks_$P.log 2>&1)
  cp /tmp/ks_$P/ks_kernel_stats.csv $O/${R}_bench_${P}_kernel_stats.csv
done
for P in bf16x6 f16x4; do
  rm -rf /tmp/ks_tts_$P
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_tts_$P -o ks -- python $OLDPWD/bench.py --no-cpu-baseline --no-extras --leg tts --steps 40 --precision $P > /tmp/ks_tts_$P.log 2>&1)
  cp /tmp/ks_tts_$P/ks_kernel_stats.csv $O/${R}_tts_${P}_kernel_stats.csv
  grep '^{' /tmp/ks_tts_$P.log | tail -1 > $O/${R}_tts_${P}_leg_under_rocprof.json      # the same run's stage events, next to the kernel statistics
done
# the training step per precision mode (bench.py --leg train runs all three in one process: its statistics would mix them)
for P in f32 bf16x6 mixed bf16x3; do
  rm -rf /tmp/ks_train_$P
  NISQA_HIP_TRAIN_PRECISION=$P rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_train_$P -o ks -- python tools/bench_train.py 32 20 > /tmp/ks_train_$P.log 2>&1
  cp /tmp/ks_train_$P/ks_kernel_stats.csv $O/${R}_train_${P}_kernel_stats.csv
done
for P in f32 bf16x6 mixed bf16x3; do NISQA_HIP_TRAIN_PRECISION=$P python tools/bench_train.py 32 20 2>/dev/null | tail -1; done > $O/${R}_train_bench.json
python tools/bench_extra.py 2>/dev/null | tail -1 > $O/${R}_side_tts_pcie.json
python tools/bench_ingest.py 2048 12 2>/dev/null | tail -1 > $O/${R}_side_ingest.json
timeout 600 python bench.py --workload predict_csv --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_predict_csv_1gpu.json
timeout 600 python tools/bench_default_flags.py 2048 2>/dev/null | tail -1 > $O/${R}_side_default_flags.json
python tools/probe_concurrency.py 2>/dev/null | grep -v amdgpu.ids > $O/${R}_probe_concurrency.txt
ls -la $O

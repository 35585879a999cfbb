#!/bin/bash
# Round profiles (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats and PMC passes for the
# predict path (both precisions) and the training step.  Output: gpurun_out/prof_rNN/ -> copy into profiles/.
#   tools/collect_profiles.sh r02
R=${1:-r02}
O=gpurun_out/prof_$R
mkdir -p $O
export TMPDIR=/tmp
# PMC passes first: bench.py reads profiles/rNN_pmc_kernels.json for roofline.traffic / mfma_util of the SAME build
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); rm -rf /tmp/pmcp_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmcp_$i -o pmc -- python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 3 > /tmp/pmcp_$i.log 2>&1
done
python tools/pmc_to_json.py $O/${R}_pmc_kernels.json /tmp/pmcp_1 /tmp/pmcp_2 /tmp/pmcp_3 /tmp/pmcp_4 > $O/${R}_pmc_kernels.txt 2>&1
cp $O/${R}_pmc_kernels.json profiles/${R}_pmc_kernels.json      # so that the bench lines below carry traffic / mfma_util
python bench.py > $O/${R}_bench_bf16x3.json 2>/dev/null
python bench.py --precision f32 --no-cpu-baseline > $O/${R}_bench_f32.json 2>/dev/null
python bench.py --streams 2 --no-cpu-baseline --no-extras > $O/${R}_bench_bf16x3_2streams.json 2>/dev/null
for P in bf16x3 f32; do
  rm -rf /tmp/ks_$P
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$P -o ks -- python bench.py --no-cpu-baseline --no-extras --precision $P > /tmp/ks_$P.log 2>&1
  cp /tmp/ks_$P/ks_kernel_stats.csv $O/${R}_bench_${P}_kernel_stats.csv
done
for P in f32 mixed bf16x3; do NISQA_HIP_TRAIN_PRECISION=$P python tools/bench_train.py 32 20 2>/dev/null | tail -1; done > $O/${R}_train_bench.json
rm -rf /tmp/ks_train
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_train -o ks -- python tools/bench_train.py 32 5 > /tmp/ks_train.log 2>&1
cp /tmp/ks_train/ks_kernel_stats.csv $O/${R}_train_kernel_stats.csv
python tools/bench_extra.py 2>/dev/null | tail -1 > $O/${R}_side_tts_pcie.json
python tools/bench_ingest.py 2048 12 2>/dev/null | tail -1 > $O/${R}_side_ingest.json
timeout 600 python bench.py --workload predict_csv --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_predict_csv_1gpu.json
timeout 600 python bench.py --workload predict_csv --no-cpu-baseline --bs 64 2>/dev/null | tail -1 > $O/${R}_bench_predict_csv_1gpu_bs64.json
REPS=4 python tools/probe_loop.py 8192 256 32 2>&1 | grep "^rep" > $O/${R}_probe_loop.txt
python tools/probe_overlap.py 2>&1 | grep -v "amdgpu.ids\|Warn" > $O/${R}_probe_overlap.txt
for k in 0 1 32; do echo "== NQ_KO=$k"; [ -f ab_libs/clk_ko$k.so ] && NISQA_BENCH_KO=1 NISQA_HIP_LIB=$PWD/ab_libs/clk_ko$k.so python tools/phase_clock.py 2>&1 | grep -v "Warn\|amdgpu.ids"; done > $O/${R}_cnn_phase_clock_ko.txt
python tools/probe_concurrency.py 2>/dev/null | grep -v amdgpu.ids > $O/${R}_probe_concurrency.txt
ls -la $O

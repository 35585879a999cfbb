#!/bin/bash
# PMC passes of the default bench (5 steps): HBM bytes, matrix-pipe busy, LDS conflicts, wait breakdown -> JSON + log
# usage: tools/gpu_pmc.sh OUTDIR [extra bench args]
O=${1:-gpurun_out/pmc}; shift
mkdir -p $O
export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1)); rm -rf /tmp/pmcp_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmcp_$i -o pmc -- python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 3 "$@" > $O/pmc_pass_$i.log 2>&1
  echo "pass $i rc $?"; tail -2 $O/pmc_pass_$i.log
done
python tools/pmc_to_json.py $O/pmc_kernels.json /tmp/pmcp_1 /tmp/pmcp_2 /tmp/pmcp_3 /tmp/pmcp_4 > $O/pmc_kernels.txt 2>&1
cat $O/pmc_kernels.txt

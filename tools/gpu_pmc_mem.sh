#!/bin/bash
# memory-pipe PMC passes of the default bench: vector-memory (TA / TCP) and LDS pipe occupancy per kernel
O=${1:-gpurun_out/pmcmem}; shift
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA_[A-Z_a-z0-9]+|TCP_[A-Z_a-z0-9]+|TD_[A-Z_a-z0-9]+|SQ_[A-Z_a-z0-9]+|MemUnit[A-Za-z]*|LDSBank[A-Za-z]*|VALUBusy|MfmaUtil|[A-Za-z]*Busy[A-Za-z]*)\b" | sort -u > $GRAFT_REPO_ROOT/$O/avail.txt
cd $GRAFT_REPO_ROOT
wc -l $O/avail.txt
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  i=$((i+1)); rm -rf /tmp/pm_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pm_$i -o pmc -- python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 3 "$@" > $O/pass_$i.log 2>&1
  echo "pass $i rc $?"; tail -2 $O/pass_$i.log
done
python tools/pmc_to_json.py $O/pmc_mem.json /tmp/pm_1 /tmp/pm_2 /tmp/pm_3 /tmp/pm_4 2>&1 | grep "cnn_front_bf16\|mel_frame" | tee $O/pmc_mem.txt

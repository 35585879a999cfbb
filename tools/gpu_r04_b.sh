#!/bin/bash
O=gpurun_out/r04b; mkdir -p $O
NISQA_HIP_TRAIN_DEBUG=1 python tools/diag_cfg5.py cfg5_mos f32 > $O/diag_debug.txt 2>&1
grep -v "^  " $O/diag_debug.txt | tail -30

#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
NQ_PRECISION=bf16x6 NISQA_HIP_LIB=$PWD/ab_libs/clock6.so python tools/phase_clock.py 2>&1 | tail -14 | tee $O/x6_phase_clock.txt

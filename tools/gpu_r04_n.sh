#!/bin/bash
O=gpurun_out/r04n; mkdir -p $O
./ab_libs/klx6 200 2>&1 | tee $O/micro_klx6.txt

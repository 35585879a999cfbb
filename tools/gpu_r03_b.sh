#!/bin/bash
# kernel iteration check: parity tests of the CNN path, A/B timing against named builds, phase clocks
# usage: tools/gpu_r03_b.sh OUT "lib1 lib2 ..." "clocklib1 ..." [pytest -k expr]
O=gpurun_out/$1; mkdir -p $O
K=${4:-"fixture or stages or properties or segment or drop_in or config"}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
bash tools/gpu_ab.sh $2 2>&1 | tee $O/ab.txt
for c in $3; do echo "== $c"; NISQA_BENCH_KO=1 NISQA_HIP_LIB=$PWD/ab_libs/$c.so python tools/phase_clock.py 2>&1 | grep -v "Warn\|amdgpu.ids"; done | tee $O/clock.txt

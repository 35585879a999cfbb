#!/bin/bash
# Same-box A/B: -mllvm -amdgpu-mfma-vgpr-form=1 on the one-wave-per-SIMD chain kernels (td, td_bf16, train, train_td)
O=gpurun_out/r04m; mkdir -p $O
V=$PWD/ab_libs/vgprform.so
stage() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['stage_ms'])"; }
{
for rep in 1 2; do
  for L in base vgprform; do
    if [ $L = base ]; then E=""; else E="NISQA_HIP_LIB=$V"; fi
    echo "== $L bf16x3: $(env $E python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 30 2>/dev/null | stage)"
    echo "== $L f32   : $(env $E python bench.py --no-cpu-baseline --no-extras --precision f32 --steps 100 --warmup 20 2>/dev/null | stage)"
    for P in f32 mixed bf16x3; do
      echo "== $L train $P: $(env $E NISQA_HIP_TRAIN_PRECISION=$P python tools/bench_train.py 32 30 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])") ms"
    done
  done
done
} 2>&1 | tee $O/ab_vgprform.txt
NISQA_HIP_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | tee $O/pytest_parity_vgprform.txt
NISQA_HIP_LIB=$V timeout 900 python -m pytest tests/test_gpu_train.py -x -q -k "training_step or fused_self_attention" 2>&1 | tail -3 | tee $O/pytest_train_vgprform.txt

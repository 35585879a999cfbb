#!/bin/bash
# Same-box ablation of the round-4 training-step changes: each switch restores the round-3 path for one component.
O=gpurun_out/r04j; mkdir -p $O
run() { # name, precision, env...
  local name=$1 prec=$2; shift; shift
  ms=$(env "$@" NISQA_HIP_TRAIN_PRECISION=$prec python tools/bench_train.py 32 30 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  printf "%-8s %-70s %s ms\n" "$prec" "$name" "$ms"
}
{
for P in f32 mixed bf16x3; do
  run "round-4 defaults" $P X=1
  run "attention block operator by operator (NISQA_HIP_TRAIN_FUSED_TD=0)" $P NISQA_HIP_TRAIN_FUSED_TD=0
  run "dense z -> dz pass as its own kernel (NISQA_HIP_TRAIN_FOLD_BN_WGRAD=0)" $P NISQA_HIP_TRAIN_FOLD_BN_WGRAD=0
  if [ $P != bf16x3 ]; then run "fp32 forward / dgrad as implicit GEMMs (NISQA_HIP_TRAIN_SEGCONV_F32_FWD=0)" $P NISQA_HIP_TRAIN_SEGCONV_F32_FWD=0; fi
  if [ $P = f32 ]; then run "fp32 wgrad as implicit GEMM, split-K (NISQA_HIP_TRAIN_SEGCONV_F32=0)" $P NISQA_HIP_TRAIN_SEGCONV_F32=0; fi
  run "all of the above off (the round-3 step)" $P NISQA_HIP_TRAIN_FUSED_TD=0 NISQA_HIP_TRAIN_FOLD_BN_WGRAD=0 NISQA_HIP_TRAIN_SEGCONV_F32_FWD=0 NISQA_HIP_TRAIN_SEGCONV_F32=0
  run "round-4 defaults (again)" $P X=1
done
} | tee $O/train_ablation.txt

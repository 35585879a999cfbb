#!/bin/bash
# A/B on ONE box: the segment-resident fp32 convolutions against the implicit GEMMs (boxes of the pool differ by up to 20 %)
O=gpurun_out/r04h; mkdir -p $O
for rep in 1 2; do
for P in f32 mixed; do
  for V in 1 0; do
    echo -n "precision $P SEGCONV_F32_FWD=$V : "
    NISQA_HIP_TRAIN_SEGCONV_F32_FWD=$V NISQA_HIP_TRAIN_PRECISION=$P python tools/bench_train.py 32 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
done | tee $O/ab_segconv_f32.txt
echo -n "bf16x3: "; NISQA_HIP_TRAIN_PRECISION=bf16x3 python tools/bench_train.py 32 30 2>/dev/null | tail -1

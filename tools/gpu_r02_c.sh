#!/bin/bash
# round-2 GPU call C: after moving the half swap of mel's packed adds to src0 -- probe every kernel pair on two streams
# (guard off), bit-identity test, and what the overlap is worth (bench with 1 / 2 / 3 streams, guard on / off)
O=gpurun_out/r02c
mkdir -p $O
export NISQA_HIP_NO_SECTION_GUARD=1
timeout 300 python tools/probe_concurrency.py bf16x3 > $O/probe_guard_off.txt 2>&1; tail -12 $O/probe_guard_off.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_streams or mel_matches or pcm16_input" 2>&1 | tail -3
for S in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 300 --streams $S > $O/bench_off_s$S.json 2>$O/err.txt
  python -c "
import json; d=json.loads(open('$O/bench_off_s$S.json').read().strip().split('\n')[-1]); print('guard off, streams $S:', d['value'], d['stage_ms'])"
done
unset NISQA_HIP_NO_SECTION_GUARD
for S in 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 300 --streams $S > $O/bench_on_s$S.json 2>$O/err.txt
  python -c "
import json; d=json.loads(open('$O/bench_on_s$S.json').read().strip().split('\n')[-1]); print('guard on, streams $S:', d['value'], d['stage_ms'])"
done

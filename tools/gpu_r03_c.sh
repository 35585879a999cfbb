#!/bin/bash
# round 3 side measurements: default flags vs tuned, host-feed curve (ingest only, 1..8 ranks), shared-GPU world 4 / 8
O=gpurun_out/r03c; mkdir -p $O
timeout 600 python tools/bench_default_flags.py 2048 2>/dev/null | tail -1 | tee $O/default_flags.json
timeout 600 python tools/bench_ingest_ranks.py --clips 32768 --pin 1 --ranks 1,2,4,8 2>/dev/null | tail -1 | tee $O/ingest_ranks.json
for n in 4 8; do
  NISQA_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2957$n bench.py --gpus $n --workload predict_csv --clips 32768 --bs 256 2>$O/shared_$n.err | tail -1 | tee $O/shared_gpu_world$n.json
done

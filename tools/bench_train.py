"""BASELINE config 5: train_nisqa_cnn_sa_ap.yaml forward + backward + Adam, bs = 32 ten-second 48 kHz clips, one GPU
(mel front end fused into the step).  Side measurement quoted in DESIGN.md; the driver's bench contract is bench.py."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth
from nisqa_amd.train import HipTrainer

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
world = int(os.environ.get('WORLD_SIZE', '1'))
local = int(os.environ.get('LOCAL_RANK', '0'))
if world > 1:                                     # torchrun: data parallel, bs clips per rank, gradients over RCCL
    torch.cuda.set_device(local)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    torch.distributed.init_process_group('nccl')
dev = torch.device('cuda:%d' % local)
args = dict(synth.MOS_ARGS)                       # model NISQA, cnn_dropout 0.2, td_sa_dropout 0.1 (the yaml's values)
tr = HipTrainer(args, synth.random_state_dict(8, 'NISQA'), dev, lr=1e-3)
pcm = np.concatenate([synth.synth_pcm16(i % 8, 10.0) for i in range(bs)])
plan = tr.eng.plan([480000] * bs, 48000)
x = tr.eng.pcm16_to_f32(torch.from_numpy(pcm).to(dev))
y = np.random.default_rng(0).uniform(1, 5, (bs, 1)).astype(np.float32)
for _ in range(3):
    tr.step_pcm(x, plan, 48000, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = tr.step_pcm(x, plan, 48000, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
if world == 1 or torch.distributed.get_rank() == 0:
    print(json.dumps({'config': 'train_nisqa_cnn_sa_ap bs=%d x 10 s per GPU' % bs, 'n_gpus': world,
                      'precision': tr.precision, 'segments': int(plan.n_wins.sum()), 'ms_per_step': round(dt * 1e3, 2),
                      'clips_per_s': round(world * bs / dt, 1), 'loss': float(loss),
                      'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))

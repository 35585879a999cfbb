import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from nisqa_amd import synth
from nisqa_amd.engine import HipNisqa
dev = torch.device('cuda:0')
margs, sd, _ = bench.model_weights()
eng = HipNisqa(margs, sd, dev)
base = [synth.synth_pcm16(i, 10.0) for i in range(8)]
pcm = torch.from_numpy(np.concatenate([base[i % 8] for i in range(64)])).to(dev)
plan = eng.plan([len(base[0])] * 64, 48000); plan.to(dev)
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 2.0)          # idle like the bench's host-side setup
for _ in range(5): eng.forward_pcm(pcm, plan, 48000)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(120)]
for a, b in ev:
    a.record(); eng.forward_pcm(pcm, plan, 48000); b.record()
torch.cuda.synchronize()
t = [a.elapsed_time(b) for a, b in ev]
print('ms per step:', ' '.join('%.3f' % x for x in t[:40]))
print('mean first 20 %.4f, steps 20-40 %.4f, 40-80 %.4f, 80-120 %.4f' % (np.mean(t[:20]), np.mean(t[20:40]), np.mean(t[40:80]), np.mean(t[80:])))

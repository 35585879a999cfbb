"""Does the sustained-MFMA probe (bench.mfma_sustained, ~100 ms of dense MFMA) leave the GPU at its loaded clock state for
the steps that follow?  argv[1] = 1: probe first; 0: idle first.  Prints per-step times of 60 steps after 5 warm-up steps."""
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
for _ in range(3): eng.forward_pcm(pcm, plan, 48000)
torch.cuda.synchronize()
time.sleep(1.0)
if len(sys.argv) > 1 and sys.argv[1] == '1':
    bench.mfma_sustained(dev)
for _ in range(5): eng.forward_pcm(pcm, plan, 48000)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for a, b in ev[:20]:
    a.record(); eng.forward_pcm(pcm, plan, 48000); b.record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for a, b in ev[20:]:
    a.record(); eng.forward_pcm(pcm, plan, 48000); b.record()
torch.cuda.synchronize()
t = [a.elapsed_time(b) for a, b in ev]
print('probe first' if len(sys.argv) > 1 and sys.argv[1] == '1' else 'idle first', ': 20 steps wall %.3f ms/step; event means: first 20 %.4f, 20-40 %.4f, 40-60 %.4f' % (
    dt / 20 * 1e3, np.mean(t[:20]), np.mean(t[20:40]), np.mean(t[40:])))

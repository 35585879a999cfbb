"""Side measurements quoted in DESIGN.md (not the driver's bench contract):
  1. nisqa_tts.tar architecture (BASELINE config 4): mixed 3-30 s clips, bs 32, 1 GPU
  2. PCIe-inclusive rate of the main config: int16 PCM starts in pinned host memory every step"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth
from nisqa_amd.engine import HipNisqa

dev = torch.device('cuda:0')
res = {}

# ---- 1. tts path ---------------------------------------------------------------------------------------
eng = HipNisqa(dict(synth.TTS_ARGS), synth.random_state_dict(9, 'NISQA_TTS'), dev)
durs = np.random.default_rng(7).uniform(3, 30, 32)
base = synth.synth_pcm16(5, 30.0)
pcm = [base[:int(d * 48000)] for d in durs]
plan = eng.plan([len(p) for p in pcm], 48000)
x = torch.from_numpy(np.concatenate(pcm)).to(dev)             # int16 PCM
for _ in range(2):
    eng.forward_pcm(x, plan, 48000)
torch.cuda.synchronize()
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(5)]
for e in ev:
    for q in e:
        q.record()
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(5):
    out = eng.forward_pcm(x, plan, 48000, stage_events=ev[s])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
res['tts_mixed_3_30s'] = {'clips_per_s': round(32 / dt, 1), 'audio_seconds_per_s': round(float(durs.sum()) / dt, 1),
                          'ms_per_batch32': round(dt * 1e3, 2), 'segments': int(plan.n_wins.sum()),
                          'stage_ms': {n: round(float(np.mean([e[i].elapsed_time(e[i + 1]) for e in ev])), 3)
                                       for i, n in enumerate(['mel', 'cnn_front', 'cnn_back', 'lstm+pool'])}}

# the same batches alternated over two streams, as the product's predict loop does: the LSTM of one batch (64
# workgroups, latency-bound) overlaps the CNN of the next
for ns in (2, 3, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
    outs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(12):
        with torch.cuda.stream(streams[s % ns]):
            outs.append(eng.forward_pcm(x, plan, 48000))
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / 12
    res['tts_mixed_3_30s']['clips_per_s_%d_streams' % ns] = round(32 / dt2, 1)

# ---- 2. PCIe-inclusive main path ------------------------------------------------------------------------
eng = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7), dev)
b16 = np.concatenate([synth.synth_pcm16(i % 8, 10.0) for i in range(64)])
host = torch.from_numpy(b16).pin_memory()
plan = eng.plan([480000] * 64, 48000)
# copies on a stream that never carries a kernel (SDMA engine; a stream that also carries kernels gets a shader blit that
# does not overlap them), kernels of alternate batches on two streams behind an event
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
copy_stream = torch.cuda.Stream(device=dev, priority=-1)
def step(s):
    with torch.cuda.stream(copy_stream):
        d16 = host.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(copy_stream)
    streams[s % 2].wait_event(ev)
    with torch.cuda.stream(streams[s % 2]):
        out = eng.forward_pcm(d16, plan, 48000)
        d16.record_stream(streams[s % 2])
        return out
for s in range(4):
    step(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(100):
    o = step(s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 100
res['pcie_inclusive_int16'] = {'clips_per_s': round(64 / dt, 1), 'ms_per_batch64': round(dt * 1e3, 3),
                               'h2d_GBps_equiv': round(b16.nbytes / dt / 1e9, 1)}
print(json.dumps(res))

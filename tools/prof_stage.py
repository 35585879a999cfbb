"""Run one stage of the hot path in isolation a few times (for rocprofv3 PMC passes).
usage: python tools/prof_stage.py {mel|cnn|td|all} [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth
from nisqa_amd.engine import HipNisqa
stage = sys.argv[1] if len(sys.argv) > 1 else 'all'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7), 'cuda:0')
base = [synth.synth_pcm16(i, 10.0) for i in range(4)]
pcm16 = np.concatenate([base[i % 4] for i in range(64)])
plan = eng.plan([len(base[0])] * 64, 48000)
pcm = eng.pcm16_to_f32(torch.from_numpy(pcm16).to(eng.device))
mel, floor = eng.mel(pcm, plan, 48000, clamp=False)
feat, p3 = eng.cnn(mel, floor, plan)
x = eng.td(feat, plan)
torch.cuda.synchronize()
for _ in range(reps):
    if stage in ('mel', 'all'):
        eng.mel(pcm, plan, 48000, clamp=False)
    if stage in ('cnn', 'all'):
        eng.cnn(mel, floor, plan)
    if stage in ('td', 'all'):
        eng.pool(eng.td(feat, plan), plan)
torch.cuda.synchronize()

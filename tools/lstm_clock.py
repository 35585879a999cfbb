"""Where a step of lstm_dir_kernel spends its cycles (build: tools/ab_build.sh lstmclk lstm "-DNQ_EXPERIMENTAL").
Run on the GPU box:  NISQA_ALLOW_DEBUG_LIB=1 NISQA_HIP_LIB=$PWD/ab_libs/lstmclk.so python tools/lstm_clock.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth, lib
from nisqa_amd.engine import HipNisqa

dev = torch.device('cuda:0')
eng = HipNisqa(dict(synth.TTS_ARGS), synth.random_state_dict(9, 'NISQA_TTS'), dev)
L = ctypes.CDLL(lib.LIB_PATH)
L.nisqa_debug_lstm_clock.restype = ctypes.c_int
L.nisqa_debug_lstm_clock.argtypes = [ctypes.c_void_p, ctypes.c_int]
secs = [30.0, 17.3, 10.0, 8.0, 5.0, 3.0] * 5 + [30.0, 3.0]
pcm = [synth.synth_pcm16(i % 8, s) for i, s in enumerate(secs)]
dev_pcm = torch.from_numpy(np.concatenate(pcm)).to(dev)
plan = eng.plan([len(p) for p in pcm], 48000)
for _ in range(3):
    eng.forward_pcm(dev_pcm, plan, 48000)
torch.cuda.synchronize()
L.nisqa_debug_lstm_clock(None, 1)
for _ in range(5):
    eng.forward_pcm(dev_pcm, plan, 48000)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
assert L.nisqa_debug_lstm_clock(out, 0) == 0
n = out[7]
names = ['h reads issued + input projection', 'recurrent product (64 v_pk_fma_f32)', 'quad sums + gate non-linearity + broadcasts',
         'state update + publish', 'workgroup barrier']
tot = sum(out[k] for k in range(5)) / n
print('steps %d, mean clock64 ticks per step (wave 0 of each workgroup) %.0f' % (n, tot))
for k, nm in enumerate(names):
    print('%-46s %7.0f  %5.1f%%' % (nm, out[k] / n, 100.0 * out[k] / n / tot))

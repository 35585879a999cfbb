"""Per-phase shader-clock profile of mel_frame_kernel (build: tools/ab_build.sh melclk mel "-DNQ_EXPERIMENTAL").
Run on the GPU box:  NISQA_ALLOW_DEBUG_LIB=1 NISQA_HIP_LIB=$PWD/ab_libs/melclk.so python tools/mel_clock.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth, lib
from nisqa_amd.engine import HipNisqa

dev = torch.device('cuda:0')
eng = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), dev)
L = ctypes.CDLL(lib.LIB_PATH)
L.nisqa_debug_mel_clock.restype = ctypes.c_int
L.nisqa_debug_mel_clock.argtypes = [ctypes.c_void_p, ctypes.c_int]
base = [synth.synth_pcm16(1000 + i, 10.0) for i in range(8)]
pcm = torch.from_numpy(np.concatenate([base[i % 8] for i in range(64)])).to(dev)
plan = eng.plan([len(base[0])] * 64, 48000)
for _ in range(10):
    eng.mel(pcm, plan, 48000, clamp=False)
torch.cuda.synchronize()
L.nisqa_debug_mel_clock(None, 1)
for _ in range(20):
    eng.mel(pcm, plan, 48000, clamp=False)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
assert L.nisqa_debug_mel_clock(out, 0) == 0
n = out[8]
names = ['window + prefetch', 'FFT r=0 + magnitudes', 'FFT r=2 + magnitudes', 'FFTs r=1,3', 'magnitudes r=1,3', 'filterbank',
         'dB + store + clip max']
tot = sum(out[q] for q in range(7)) / n
print('frames %d, mean clock64 ticks per frame (one wave, two waves per SIMD) %.0f' % (n, tot))
for q, nm in enumerate(names):
    print('%-26s %8.0f  %5.1f%%' % (nm, out[q] / n, 100.0 * out[q] / n / tot))

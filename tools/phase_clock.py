"""Per-phase shader-clock profile of cnn_front_bf16_kernel (build: tools/phase_clock.sh -> ab_libs/clock.so).

Run on the GPU box:  NISQA_ALLOW_DEBUG_LIB=1 NISQA_HIP_LIB=$PWD/ab_libs/clock.so python tools/phase_clock.py
Prints the mean cycles one wave spends between the layer boundaries (under the real 2-waves-per-SIMD contention)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth, lib
from nisqa_amd.engine import HipNisqa

dev = torch.device('cuda:0')
PREC = os.environ.get('NQ_PRECISION', 'bf16x3')          # bf16x6: cnn_front_bf16x6_kernel (its own stamp buffer)
eng = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), dev, precision=PREC)
L = ctypes.CDLL(lib.LIB_PATH)
DBG = getattr(L, 'nisqa_debug_phase_clock6' if PREC == 'bf16x6' else 'nisqa_debug_phase_clock')
DBG.restype = ctypes.c_int
DBG.argtypes = [ctypes.c_void_p, ctypes.c_int]
base = [synth.synth_pcm16(1000 + i, 10.0) for i in range(8)]
pcm = torch.from_numpy(np.concatenate([base[i % 8] for i in range(64)])).to(dev)
plan = eng.plan([len(base[0])] * 64, 48000)
for _ in range(20):
    eng.forward_pcm(pcm, plan, 48000)
torch.cuda.synchronize()
DBG(None, 1)
for _ in range(50):                      # every wave overwrites its own slot: the last launch's numbers are read
    eng.forward_pcm(pcm, plan, 48000)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
assert DBG(out, 0) == 0
n = out[12]
names = ['stage patch', 'conv1+pool', 'conv2 K loop', 'conv2 epilogue', 'conv3 K loop', 'conv3 epilogue', 'conv4 K loop',
         'barrier+conv4 epilogue+barrier', 'conv5 K loop', 'conv5 epilogue+barrier', 'conv6 K loop', 'conv6 epilogue']
tot = sum(out[q] for q in range(12)) / n
print('waves %d, mean clock64 ticks per wave %.0f (+ %.0f before the first stamp); wall clock (100 MHz) per wave %.2f us'
      ' -> shader clock %.0f MHz' % (n, tot, out[14] / n, out[13] / n / 100.0, (tot + out[14] / n) / (out[13] / n / 100.0)))
for q, nm in enumerate(names):
    print('%-32s %9.0f  %5.1f%%' % (nm, out[q] / n, 100.0 * out[q] / n / tot))

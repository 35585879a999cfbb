"""Per-phase shader-clock profile of td16_proj_kernel / td16_layer_kernel (build: tools/ab_build.sh td16clk td16_bf16x6
"-DNQ_EXPERIMENTAL -mllvm -amdgpu-mfma-vgpr-form=1").
Run on the GPU box:  NISQA_ALLOW_DEBUG_LIB=1 NISQA_HIP_LIB=$PWD/ab_libs/td16clk.so python tools/td16_clock.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth, lib
from nisqa_amd.engine import HipNisqa

dev = torch.device('cuda:0')
eng = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), dev)
L = ctypes.CDLL(lib.LIB_PATH)
base = [synth.synth_pcm16(1000 + i, 10.0) for i in range(8)]
pcm = torch.from_numpy(np.concatenate([base[i % 8] for i in range(64)])).to(dev)
plan = eng.plan([len(base[0])] * 64, 48000)
for _ in range(20):
    eng.forward_pcm(pcm, plan, 48000)
torch.cuda.synchronize()
NAMES = {'proj': ['wait chunk 0', 'K-steps 0-3', 'wait chunk 1', 'K-steps 4-7', 'wait chunk 2', 'K-steps 8-11',
                  'LN + x store + wait QKV frags', 'Q / K / V GEMMs + stores'],
         'layer': ['attention loop', 'wait weights', 'out-proj + LN', 'feed-forward + LN + store',
                   'wait next QKV frags | last layer: own tile terms + values + barrier', 'Q / K / V GEMMs + stores | last layer: blocks x 4 tiles',
                   'last layer: barrier + scores stored', 'last layer: arrival + pooling softmax']}
LOOP = getattr(L, 'nisqa_debug_td16_loop_clock')
LOOP.restype, LOOP.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
LOOP(None, 1)
for k in ('proj', 'layer'):
    DBG = getattr(L, 'nisqa_debug_td16_%s_clock' % k)
    DBG.restype, DBG.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
    DBG(None, 1)
    for _ in range(5):                    # every wave overwrites its own slot: the last launch's numbers are read
        eng.forward_pcm(pcm, plan, 48000)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    assert DBG(out, 0) == 0
    n = out[12]
    tot = sum(out[q] for q in range(12)) / n
    print('%s: waves %d, mean clock64 ticks per wave %.0f (+ %.0f before the first stamp); wall (100 MHz) %.2f us' % (k, n, tot, out[14] / n, out[13] / n / 100.0))
    print('    %-36s %9.0f' % ('requests, clip lookup (layer: first K/V block)', out[14] / n))
    for q, nm in enumerate(NAMES[k]):
        print('    %-36s %9.0f  %5.1f%%' % (nm, out[q] / n, 100.0 * out[q] / n / tot))
out = (ctypes.c_ulonglong * 16)()
assert LOOP(out, 0) == 0
nb = max(1, out[8])
print('attention loop, per key block and wave (%d block iterations):' % nb)
for q, nm in enumerate(['', 'S of the four tiles (K fragments x Q fragments from LDS)', '', 'softmax + rescale + split + P V of the four tiles']):
    if not nm:
        continue
    print('    %-40s %8.0f' % (nm, out[q] / nb))

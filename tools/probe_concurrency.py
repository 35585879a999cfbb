"""Concurrency probe (GPU): does running work on two HIP streams at the same time change any result?

Round 1 found that a mel_frame_kernel workgroup sharing a CU with a split-bf16 conv workgroup of ANOTHER stream computed
wrong spectrogram frames.  Round 2 traced it (tools/micro/corun3..6.hip) to one instruction form: a v_pk_{add,mul,fma}_f32
whose low result reads the high half of a VGPR src1 (op_sel:[x,1]) misreads in lanes 48..63 while bf16 / f16 MFMA waves of
another kernel share the SIMD.  The mel kernel's half-swapped operands now sit in src0 (exact), tests/test_host.py lints
every kernel's ISA for the form, and this probe -- all kernel pairs on two streams, compared bit for bit with their
serial results -- prints zeros (profiles/r02_probe_concurrency_no_guard.txt).

    python tools/probe_concurrency.py [bf16x3|f32]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth
from nisqa_amd.engine import HipNisqa

dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
eng = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), dev, precision=prec)
base = [synth.synth_pcm16(1000 + i, 10.0) for i in range(8)]
L, n = len(base[0]), 32
pcmA = torch.from_numpy(np.concatenate([base[i % 8] for i in range(n)])).to(dev)
pcmB = torch.from_numpy(np.concatenate([base[(i + 3) % 8] for i in range(n)])).to(dev)
plA, plB = eng.plan([L] * n, 48000), eng.plan([L] * n, 48000)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def tup(x):
    return x if isinstance(x, tuple) else (x,)


def cross(name, fa, a, fb, b, reps=10):
    """fa(*a) on stream 1 next to fb(*b) on stream 2; worst deviation of each from its serial result"""
    ra0 = fa(*a); torch.cuda.synchronize(); rb0 = fb(*b); torch.cuda.synchronize()
    worst = [0.0, 0.0]
    for _ in range(reps):
        torch.cuda.synchronize()
        with torch.cuda.stream(s2): rb = fb(*b)
        with torch.cuda.stream(s1): ra = fa(*a)
        torch.cuda.synchronize()
        for i, (x, y) in enumerate(((ra, ra0), (rb, rb0))):
            worst[i] = max(worst[i], max(float((p - q).abs().max()) for p, q in zip(tup(x), tup(y))))
    print('%-28s max|d| first %.3g  second %.3g' % (name, worst[0], worst[1]))


f_mel = lambda p, pl: eng.mel(p, pl, 48000, clamp=False)
f_cnn = lambda m, f, pl: eng.cnn(m, f, pl)[0]
f_td = lambda f, pl: eng.td(f, pl)
f_pool = lambda x, pl: eng.pool(x, pl)
f_all = lambda p, pl: eng.forward_pcm(p, pl, 48000)
melA, flA = f_mel(pcmA, plA); melB, flB = f_mel(pcmB, plB)
fA, fB = f_cnn(melA, flA, plA), f_cnn(melB, flB, plB)
xA, xB = f_td(fA, plA), f_td(fB, plB)
torch.cuda.synchronize()
print(prec)
cross('mel | mel', f_mel, (pcmA, plA), f_mel, (pcmB, plB))
cross('mel | cnn', f_mel, (pcmA, plA), f_cnn, (melB, flB, plB))
pcmAf = eng.pcm16_to_f32(pcmA)
cross('mel (float samples) | cnn', f_mel, (pcmAf, plA), f_cnn, (melB, flB, plB))
cross('mel | self-attention', f_mel, (pcmA, plA), f_td, (fB, plB))
cross('mel | pooling', f_mel, (pcmA, plA), f_pool, (xB, plB))
cross('cnn | cnn', f_cnn, (melA, flA, plA), f_cnn, (melB, flB, plB))
cross('cnn | self-attention', f_cnn, (melA, flA, plA), f_td, (fB, plB))
cross('cnn | pooling', f_cnn, (melA, flA, plA), f_pool, (xB, plB))
cross('self-attention | pooling', f_td, (fA, plA), f_pool, (xB, plB))
cross('whole forward | whole forward', f_all, (pcmA, plA), f_all, (pcmB, plB), reps=30)

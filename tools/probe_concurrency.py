"""Concurrency probe (GPU): does running work on two HIP streams at the same time change any result?

Finding on gfx950 (MI355X, ROCm 7.2): a mel_frame_kernel workgroup that shares a CU with a split-bf16 conv workgroup
(cnn_front_bf16_kernel / cnn_std_bf16_kernel) of ANOTHER stream computes a few hundred to a few thousand wrong
spectrogram values per launch (errors up to ~10 dB in ~1 % of the frames; the conv kernel's own output stays exact).
Every other pair of kernels is bit-exact under overlap, and so is the pair when the two cannot share a CU (LDS request
of the conv kernel inflated to 96 KB).  Synthetic LDS / MFMA / VALU stress kernels (tools/micro/lds_fill.hip,
corun.hip) do not reproduce it; the cause is not found yet.  The engine therefore keeps the mel + CNN sections of
batches on different streams apart (nisqa_model_dev.conv_section_wait / conv_section_done); the last lines below check
that whole forwards on two streams are bit-identical to serial ones.

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

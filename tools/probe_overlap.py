"""Does the H2D copy of batch i+1 overlap the kernels of batch i?  Times 20 x (245 MB page-locked H2D) alone, 20 x
(forward of a 256-clip batch) alone, and both issued on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth
from nisqa_amd.engine import HipNisqa

dev = torch.device('cuda:0')
eng = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), dev)
B = 256
base = [synth.synth_pcm16(1000 + i, 10.0) for i in range(8)]
host = torch.from_numpy(np.concatenate([base[i % 8] for i in range(B)])).pin_memory()
pcm = host.to(dev)
plan = eng.plan([len(base[0])] * B, 48000)
dst = torch.empty_like(pcm)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def run(copy, compute, n=20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        if copy:
            with torch.cuda.stream(sb):
                dst.copy_(host, non_blocking=True)
        if compute:
            with torch.cuda.stream(sa):
                eng.forward_pcm(pcm, plan, 48000)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

for _ in range(2):
    run(True, True, 3)
print('H2D alone      %.2f ms per batch' % run(True, False))
print('kernels alone  %.2f ms per batch' % run(False, True))
print('both, 2 streams %.2f ms per batch' % run(True, True))

def loop_like(n=20):
    """copy + kernels of one batch on the SAME stream, batches alternating over two streams (the predict loop of round 2)"""
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream((sa, sb)[i & 1]):
            d = host.to(dev, non_blocking=True)
            eng.forward_pcm(d, plan, 48000)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

def copy_stream(n=20):
    """all copies on a stream of their own (never carries a kernel), kernels on two alternating streams behind an event"""
    sc = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(sc):
            d = host.to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(sc)
        st = (sa, sb)[i & 1]
        st.wait_event(ev)
        with torch.cuda.stream(st):
            eng.forward_pcm(d, plan, 48000)
            d.record_stream(st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

loop_like(4); copy_stream(4)
print('copy + kernels per batch on one stream, two streams alternating  %.2f ms per batch' % loop_like())
print('dedicated copy stream + two kernel streams                       %.2f ms per batch' % copy_stream())

# stream generations: which part of the pattern changes from one set of freshly created streams to the next?
for gen in range(8):
    sc = torch.cuda.Stream(dev, priority=-1)
    k0, k1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def h2d_only(n=10):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            with torch.cuda.stream(sc):
                dst.copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    def kern_only(n=10):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream((k0, k1)[i & 1]):
                eng.forward_pcm(pcm, plan, 48000)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    def both(n=10):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(sc):
                d = host.to(dev, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(sc)
            st = (k0, k1)[i & 1]
            st.wait_event(ev)
            with torch.cuda.stream(st):
                eng.forward_pcm(d, plan, 48000)
                d.record_stream(st)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    both(2)
    print('generation %d: H2D alone %.2f, kernels alone %.2f, pipeline %.2f ms per batch' % (gen, h2d_only(), kern_only(), both()), flush=True)

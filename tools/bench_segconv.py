"""Per-layer timing of the training convolutions at the bench batch (7 904 segments): implicit GEMM (fp32, split-bf16) against
the segment-resident kernels (csrc/train_conv.hip).  Run on the GPU box: python tools/bench_segconv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nisqa_amd import lib

L = lib.load()
dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 7904
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print('%-22s %9s %9s %9s %9s %9s %9s %9s   (us; TFLOP/s of the segment-resident kernels, 3 products counted once)' % (
    'layer', 'fwd f32', 'fwd bf16', 'fwd seg', 'dgr bf16', 'dgr seg', 'wgr bf16', 'wgr seg'))
for (h, w, ci, co, pad) in [(24, 7, 16, 32, 1), (12, 5, 32, 64, 1), (12, 5, 64, 64, 1), (6, 3, 64, 64, 1), (6, 3, 64, 64, 0)]:
    wo = w + 2 * pad - 2
    g = torch.Generator(device='cpu').manual_seed(1)
    x = torch.randn(S, h * w, ci, generator=g).to(dev)
    dz = torch.randn(S, h * wo, co, generator=g).to(dev)
    wk = (torch.randn(co, 9 * ci, generator=g) * 0.1).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    z = torch.empty(S * h * wo, co, device=dev)
    dx = torch.empty(S, h * w, ci, device=dev)
    dw = torch.zeros(co, 9 * ci, device=dev)
    st2 = torch.zeros(2 * co, dtype=torch.float64, device=dev)
    fr = []
    for mode in (0, 1):
        f = torch.empty(L.nisqa_segconv_frag_bytes(mode, ci, co) // 2, dtype=torch.int16, device=dev)
        lib.check(L.nisqa_segconv_pack(mode, p(wk), ci, co, p(f), st), 'pack')
        fr.append(f)
    rows = S * h * wo
    ks = max(1, min(2048, rows // 128))
    t = [timeit(lambda: L.nisqa_conv3x3_fwd_stats(0, p(x), p(wk), p(z), S, h, w, ci, co, pad, p(b), p(st2), st)),
         timeit(lambda: L.nisqa_conv3x3_fwd_stats(1, p(x), p(wk), p(z), S, h, w, ci, co, pad, p(b), p(st2), st)),
         timeit(lambda: L.nisqa_segconv_bf16(0, p(x), p(fr[0]), p(z), S, h, w, ci, co, pad, p(b), p(st2), st)),
         timeit(lambda: L.nisqa_conv3x3_gemm_bf16(1, p(dz), p(wk), p(dx), S, h, w, ci, co, pad, None, 1, st)),
         timeit(lambda: L.nisqa_segconv_bf16(1, p(dz), p(fr[1]), p(dx), S, h, w, ci, co, pad, None, None, st)),
         timeit(lambda: L.nisqa_conv3x3_gemm_bf16(2, p(x), p(dz), p(dw), S, h, w, ci, co, pad, None, ks, st))]
    wg = getattr(L, 'nisqa_segconv_wgrad_bf16', None)
    t.append(timeit(lambda: wg(p(x), p(dz), p(dw), S, h, w, ci, co, pad, st)) if wg is not None else float('nan'))
    fl = 2.0 * rows * co * 9 * ci
    dbg = getattr(L, 'nisqa_debug_segconv_clock', None)
    if dbg is not None:                      # -DNQ_EXPERIMENTAL build of train_conv.hip (NISQA_ALLOW_DEBUG_LIB=1): mean shader-clock cycles per group and wave, by phase
        import ctypes
        dbg.restype, dbg.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
        names = ['issue loads', 'barrier (prev K loops)', 'wait + split + store', 'barrier', 'K loop', 'epilogue']
        for mode, nm in ((0, 'fwd'), (1, 'dgrad'), (2, 'wgrad')):
            dbg(None, 1)
            if mode == 0:
                L.nisqa_segconv_bf16(0, p(x), p(fr[0]), p(z), S, h, w, ci, co, pad, p(b), p(st2), st)
            elif mode == 1:
                L.nisqa_segconv_bf16(1, p(dz), p(fr[1]), p(dx), S, h, w, ci, co, pad, None, None, st)
            else:                                # wgrad: 'epilogue' = the final atomics (per wave, not per group)
                L.nisqa_segconv_wgrad_bf16(p(x), p(dz), p(dw), S, h, w, ci, co, pad, st)
            torch.cuda.synchronize()
            o8 = (ctypes.c_ulonglong * 16)()
            dbg(o8, 0)
            g = max(1, o8[6])
            print('    %s: %d waves, %.2f groups per wave; cycles per group: ' % (nm, o8[7], g / max(1, o8[7])) +
                  ', '.join('%s %.0f' % (names[q], o8[q] / g) for q in range(6)))
    print('%-22s %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f %9.1f   fwd %.0f dgrad %.0f' % (
        '%dx%d %d->%d pad %d' % (h, w, ci, co, pad), *t, fl / t[2] / 1e6, fl / t[4] / 1e6))

#!/usr/bin/env python
"""bench.py -- clips/sec of the NISQA predict hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default workload (the contract's bench line), BASELINE.json configs[1] per GPU: a "step" is ONE pass of the whole hot path
(PCM -> mel -> AdaptCNN -> self-attention -> pooling heads -> [B,5] rows) over one batch of bs = 64 synthetic 10 s /
48 kHz clips (SURVEY.md section 8d generator), int16 PCM as in the WAV data chunks, already resident in HBM.  Weights:
the real nisqa.tar when it is on this machine (oracle/_ref/weights, staged by __graft_entry__.build(), or
$NISQA_WEIGHTS_DIR), else seeded random weights of the same architecture -- `data` says which.  Weak scaling: each rank
owns its own 64-clip batch (clips shard with no data-path collective); the only exchange is the final all_gather of
result rows, inside the timed region.  value = clips all ranks processed / max-over-ranks wall time.

    ... bench.py --workload predict_csv --clips 100000 --bs 256        (BASELINE.json configs[2], strong scaling)

drives the drop-in surface itself: nisqaModel(predict_csv).predict() over a CSV of N clips (WAV files on local disk,
`--distinct` different ones reused cyclically) -> native ingest -> H2D -> kernels -> all_gather; N is fixed, ranks
shard it (`"scaling": "strong"`), a "step" is one bs-clip batch of a rank.  PCIe-inclusive: not the contract line.

Extra objects on the JSON line (DESIGN.md "Measurement"):
  roofline            dominant kernel (cnn_front_bf16_kernel, the whole AdaptCNN: 160.6 GFLOP per launch at bs 64):
                      algorithmic FLOPs / mean launch duration from HIP events recorded inside the timed region on the
                      launch stream, against the dense bf16 MFMA peak; `traffic` = HBM bytes per launch and `mfma_util`
                      from the rocprofv3 PMC passes of the same build (profiles/rNN_pmc_kernels.json; null when absent).
  roofline_secondary  mel_frame_kernel against the fp32 VALU peak (FLOPs of the pruned FFT actually executed).
  kernels             per stage: mean ms, mfma_util / HBM bytes from the same PMC file.
  alt_precision       the same workload on the exact-fp32 path, measured after the timed region.
  cpu_baseline        the CPU oracle (port of the reference path) at bs = 64 on this box's host cores, rank 0, N = 1.
"""
import argparse
import glob
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from nisqa_amd import synth                      # noqa: E402

BATCH = 64
SECONDS = 10.0
SR = 48000
N_DISTINCT = 16          # distinct synthetic clips per rank (tiled to BATCH); generation is host-bound

# Algorithmic FLOPs per 10 s clip (247 segments), SURVEY.md section 8a / BASELINE.md:
FLOP_CONV1_4 = (51.2 + 382.4 + 546.3 + 1092.6) * 1e6      # what one cnn_front_kernel launch does, per clip
FLOP_CONV5_6 = (327.8 + 109.3) * 1e6
FLOP_TOTAL = 2.93e9                                          # whole path incl. mel in dense FFT form
# mel_frame_kernel, per frame, as executed: window 1 024 + pre-twiddles 3 x 512 complex mul (9 216) + four
# complex FFT-512 at 5 N log2 N (92 160) + real-input recombination of 2 048 bins (~10 flop each, 20 480) + 1 707
# magnitudes (4 each, 6 828) + sparse filter bank 3 414 fma (6 828) + 48 x log10
FLOP_MEL_FRAME = 1024 + 9216 + 92160 + 20480 + 6828 + 6828 + 48 * 4
PEAK_F32 = 157.3                                             # TFLOP/s fp32 (vector = fp32 MFMA), MI355X_MICROARCH.md
PEAK_BF16_MFMA = 2500.0                                      # TFLOP/s dense, MI355X_MICROARCH.md
N_SIMD = 1024                                                # 256 CUs x 4


def find_weights(name='nisqa.tar'):
    for d in (os.environ.get('NISQA_WEIGHTS_DIR', ''), os.path.join(ROOT, 'oracle', '_ref', 'weights')):
        p = os.path.join(d, name) if d else ''
        if p and os.path.isfile(p):
            return p
    return None


def model_weights():
    """(args, state_dict, description): the real nisqa.tar when present, else seeded random weights."""
    p = find_weights('nisqa.tar')
    if p:
        ck = torch.load(p, map_location='cpu', weights_only=False)
        return ck['args'], ck['model_state_dict'], 'weights: nisqa.tar (%s)' % os.path.relpath(p, ROOT)
    return dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), 'weights: random-init nisqa.tar architecture (checkpoint not on this machine)'


def pmc_kernels():
    """Per-kernel PMC means of the newest profiles/rNN_pmc_kernels.json (tools/pmc_to_json.py), or {}."""
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_kernels.json')),
                   key=lambda f: int(re.search(r'r(\d+)_', os.path.basename(f)).group(1)))
    if not files:
        return {}, None
    with open(files[-1]) as f:
        return json.load(f).get('kernels', {}), os.path.relpath(files[-1], ROOT)


def pmc_derived(k):
    """HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB units, the gfx950 x2 correction for wide reads) and the
    matrix-pipe utilisation SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1 024 SIMDs)."""
    if not k:
        return None, None
    tr = None
    if 'FETCH_SIZE' in k and 'WRITE_SIZE' in k:
        tr = int(round((2.0 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024.0))
    mu = None
    if k.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in k:
        mu = round(k['SQ_VALU_MFMA_BUSY_CYCLES'] / (k['GRBM_GUI_ACTIVE'] / 8.0 * N_SIMD), 4)
    return tr, mu


def mfma_sustained(dev):
    """Dense bf16 MFMA rate and shader clock this GPU SUSTAINS on random operands (nisqa_probe_mfma_sustained: two waves
    per SIMD of back-to-back v_mfma_f32_32x32x16_bf16 on registers, ~50 ms).  On MI355X the clock drops from ~2.38 GHz
    (zero operands: the data-sheet 2.5 PFLOP/s) to ~1.8 GHz under random data -- the ceiling a bf16 kernel on real data has."""
    import ctypes
    from nisqa_amd import lib as _lib
    L = _lib.load()
    blocks, iters = 4096, 12000
    g = torch.Generator().manual_seed(5)
    bits = torch.randint(0, 1 << 16, (65536 * 8,), generator=g, dtype=torch.int32)
    bits = ((bits & 0x807f) | 0x3f00 | ((bits >> 3) & 0x0080)).to(torch.int16)        # sign, 7 mantissa bits, exponent 126/127
    res = {}
    for name, ops in (('zeros', torch.zeros_like(bits)), ('random', bits)):
        d_ops = ops.to(dev)
        out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
        clk = torch.zeros(blocks * 4 * 2, dtype=torch.int64, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(2):                                                             # the first launch ramps the clock
            e0.record()
            _lib.check(L.nisqa_probe_mfma_sustained(d_ops.data_ptr(), out.data_ptr(), clk.data_ptr(), blocks, iters, st),
                       'nisqa_probe_mfma_sustained')
            e1.record()
            torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1)
        c = clk.view(-1, 2).double().sum(0)
        res[name] = {'tflops': round(blocks * 4 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12, 1),
                     'shader_clock_mhz': round(float(c[0] / c[1]) * 100.0, 0)}
    return res


def cpu_baseline(args, sd, bs=BATCH):
    """Oracle (CPU port of the reference path) on ONE bs = 64 batch of 10 s clips: mel per clip as the reference's
    dataset does (librosa is serial at num_workers = 0; here numpy's single-threaded pocketfft), then the network on the
    packed batch like predict_dim (NL:1441-1467; Framewise packs all segments into one CNN call, NL:487-502)."""
    from oracle import mel as omel, net as onet
    nthr = int(torch.get_num_threads())
    clips = [synth.synth_pcm16(2000 + i, SECONDS).astype(np.float32) / np.float32(32768.0) for i in range(8)]
    sdt = {k: torch.as_tensor(np.asarray(v)).float() if not torch.is_tensor(v) else v.float()
           for k, v in sd.items() if k.split('.')[-1] != 'num_batches_tracked'}
    onet.predict_from_melspec(sd, args, omel.melspec_db_from_audio(clips[0][:SR], SR))   # warm-up
    t0 = time.perf_counter()
    specs = [omel.melspec_db_from_audio(clips[i % 8], SR) for i in range(bs)]
    t_mel = time.perf_counter() - t0
    t1 = time.perf_counter()
    with torch.no_grad():
        segs = [onet.segment_specs(s, args['ms_seg_length'], args['ms_seg_hop_length'], None) for s in specs]
        feat = onet.adapt_cnn(sdt, torch.cat([x for x, _ in segs], 0), args['cnn_pool_1'], args['cnn_pool_2'], args['cnn_pool_3'])
        o, rows = 0, []
        for _, n in segs:
            td = onet.self_attention(sdt, feat[o:o + n], args['td_sa_num_layers'])
            rows.append(torch.cat([onet.pool_att_ff(sdt, td, 'pool_layers.%d.model.' % h) for h in range(5)]))
            o += n
    t_net = time.perf_counter() - t1
    dt = t_mel + t_net
    return {'value': round(bs / dt, 3), 'unit': 'clips/s', 'cores': nthr, 'kind': 'port',
            'network_only': round(bs / t_net, 3), 'mel_only': round(bs / t_mel, 3),
            'sample': 'one bs = %d batch of 10 s clips (8 distinct): oracle.mel (numpy restatement of librosa 0.8.1, FFT '
                      'on 1 thread like the reference dataset at num_workers = 0) %.1f s + oracle.net (torch CPU fp32, %d '
                      'threads, CNN on the packed %d segments) %.1f s; host has %d cores'
                      % (bs, t_mel, nthr, int(sum(n for _, n in segs)), t_net, os.cpu_count())}


def init_dist(a):
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus and world == 1 and a.gpus > 1:
        raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                         % (a.gpus, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU path)')
    # NISQA_BENCH_SHARED_GPU=1 is a test knob: all ranks share cuda:0 over gloo, to exercise the N > 1 code path on a
    # one-GPU box; the real multi-GPU run is one rank per GPU over RCCL ("nccl")
    shared = os.environ.get('NISQA_BENCH_SHARED_GPU') == '1'
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = 'gloo' if shared else 'nccl'
        if shared:
            torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    return rank, world, dev, backend


def bench_predict_csv(a):
    """BASELINE configs[2]: nisqaModel(predict_csv).predict() over N clips, bs per GPU, ranks shard the CSV."""
    import contextlib
    import io
    import shutil
    import tempfile
    import pandas as pd
    from nisqa_amd.NISQA_model import nisqaModel
    rank, world, dev, backend = init_dist(a)
    margs, sd, wdesc = model_weights()
    tmp = a.tmp_dir or tempfile.gettempdir()
    d = os.path.join(tmp, 'nisqa_bench_csv')
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        for i in range(a.distinct):
            synth.write_wav(os.path.join(d, 'c%05d.wav' % i), synth.synth_pcm16(3000 + i, SECONDS), SR)
        pd.DataFrame({'deg': ['c%05d.wav' % (i % a.distinct) for i in range(a.clips)]}).to_csv(os.path.join(d, 'list.csv'), index=False)
        ck = dict(margs)
        ck.update({'pretrained_model': False, 'tr_bs_val': a.bs, 'tr_num_workers': a.workers})
        torch.save({'args': ck, 'model_state_dict': sd}, os.path.join(d, 'model.tar'))
    if world > 1:
        torch.distributed.barrier()

    def args_for(csv):
        return {'mode': 'predict_csv', 'pretrained_model': os.path.join(d, 'model.tar'), 'deg': None, 'data_dir': d,
                'output_dir': None, 'csv_file': csv, 'csv_deg': 'deg', 'num_workers': a.workers, 'bs': a.bs,
                'ms_channel': None, 'tr_bs_val': a.bs, 'tr_num_workers': a.workers}

    # warm-up: W batches per rank through the same path (page cache, engine, pinned ring)
    if rank == 0:
        pd.DataFrame({'deg': ['c%05d.wav' % (i % a.distinct) for i in range(max(1, a.warmup) * a.bs * world)]}).to_csv(
            os.path.join(d, 'warm.csv'), index=False)
    if world > 1:
        torch.distributed.barrier()
    quiet = io.StringIO()
    with contextlib.redirect_stdout(quiet):
        nisqaModel(args_for('warm.csv')).predict()
        m = nisqaModel(args_for('list.csv'))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(quiet):
        df = m.predict()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        torch.distributed.barrier()
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        assert len(df) == a.clips and np.isfinite(df['mos_pred'].to_numpy()).all()
        steps = -(-(-(-a.clips // world)) // a.bs)
        print(json.dumps({
            'metric': 'clips/sec (10 s, 48 kHz)', 'value': round(a.clips / dt, 2), 'unit': 'clips/s', 'n_gpus': world,
            'steps': steps, 'warmup': max(1, a.warmup), 'ms_per_step': round(1e3 * dt / steps, 4), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'bf16x3 (bf16 hi+lo operands, 3 MFMA products per term, f32 accumulate; mel/attention/pooling f32)',
            'data': 'synthetic 48 kHz / 10 s PCM16 WAV files on local disk (%d distinct, reused cyclically), %s' % (a.distinct, wdesc),
            'config': {'workload': 'predict_csv nisqa.tar bs=%d per GPU, %d synthetic 10 s 48 kHz clips, clip-sharded over '
                                   '%d rank(s); WAV files -> native ingest -> H2D -> kernels -> all_gather (PCIe-inclusive)'
                                   % (a.bs, a.clips, world),
                       'clips': a.clips, 'bs': a.bs, 'distinct_files': a.distinct, 'ingest_workers': a.workers,
                       'parallelism': 'clip-sharded x%d, final all_gather of MOS rows' % world,
                       'collective_backend': backend, 'world_size_seen': world}}))
        shutil.rmtree(d, ignore_errors=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    global BATCH
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # a step is ~1 ms: 50 + 400 steps are under half a second of GPU time and let the clocks settle (20 steps after 3
    # warm-up steps measure ~7 % low)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default=None, choices=['bf16x3', 'f32'])
    ap.add_argument('--no-extras', action='store_true', help='skip the alt-precision and 3-stream passes (profiling runs)')
    ap.add_argument('--streams', type=int, default=1, help='HIP streams the steps alternate over')
    ap.add_argument('--workload', default='predict_dir', choices=['predict_dir', 'predict_csv'])
    ap.add_argument('--clips', type=int, default=100000, help='predict_csv: clips in the CSV (whole job)')
    ap.add_argument('--bs', type=int, default=256, help='predict_csv: batch size per GPU')
    ap.add_argument('--distinct', type=int, default=256, help='predict_csv: distinct WAV files written')
    ap.add_argument('--workers', type=int, default=16, help='predict_csv: native ingest threads per rank')
    ap.add_argument('--tmp-dir', default=None)
    ap.add_argument('--batch', type=int, default=BATCH, help='clips per step (experiments; the contract line is 64)')
    a = ap.parse_args()
    BATCH = a.batch
    if a.workload == 'predict_csv':
        return bench_predict_csv(a)

    rank, world, dev, backend = init_dist(a)
    from nisqa_amd.engine import HipNisqa
    margs, sd, wdesc = model_weights()
    eng = HipNisqa(margs, sd, dev, precision=a.precision)

    # synthetic batch, resident in HBM before the timed region
    base = [synth.synth_pcm16(1000 * rank + i, SECONDS) for i in range(N_DISTINCT)]
    pcm16 = np.concatenate([base[i % N_DISTINCT] for i in range(BATCH)])
    plan = eng.plan([len(base[0])] * BATCH, SR)
    pcm = torch.from_numpy(pcm16).to(dev)        # int16, as read from the WAV data chunks; scaled inside the mel kernel
    plan.to(dev)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, a.streams))]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
    for i in range(max(a.warmup, len(streams))):
        with torch.cuda.stream(streams[i % len(streams)]):
            out = eng.forward_pcm(pcm, plan, SR)
    barrier()

    evs = []
    for _ in range(a.steps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        for x in e:
            x.record(streams[len(evs) % len(streams)])   # forces handle creation; re-recorded inside the library
        evs.append(e)
    barrier()
    t0 = time.perf_counter()
    outs = []
    for s in range(a.steps):
        with torch.cuda.stream(streams[s % len(streams)]):
            outs.append(eng.forward_pcm(pcm, plan, SR, stage_events=evs[s]))
    for st in streams:
        torch.cuda.current_stream(dev).wait_stream(st)
    if world > 1:                           # the path's one exchange step: gather the MOS rows
        rows = torch.cat(outs, 0)
        parts = [torch.empty_like(rows) for _ in range(world)]
        torch.distributed.all_gather(parts, rows)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        torch.distributed.barrier()
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        if os.environ.get('NISQA_BENCH_KO') != '1':           # knock-out builds (tools/ab_build.sh -DNQ_KO=..) compute garbage
            assert torch.isfinite(outs[-1]).all()
        names = ['mel', 'cnn_front', 'cnn_back', 'selfatt', 'pool']
        stage_ms = {n: float(np.mean([evs[s][i].elapsed_time(evs[s][i + 1]) for s in range(a.steps)]))
                    for i, n in enumerate(names)}
        pmc, pmc_file = pmc_kernels()

        def roofline_of(prec, ms_front):
            if prec == 'bf16x3':
                flop, peak, kname, kern = (FLOP_CONV1_4 + FLOP_CONV5_6) * BATCH, PEAK_BF16_MFMA, 'cnn_front_bf16_kernel', \
                    'cnn_front_bf16_kernel (AdaptCNN conv1-6 + pools, split-bf16 MFMA: 3 products per term)'
            else:
                flop, peak, kname, kern = FLOP_CONV1_4 * BATCH, PEAK_F32, 'cnn_front_kernel', 'cnn_front_kernel (conv1-4 + pools, fp32 MFMA)'
            ach = flop / (ms_front * 1e-3) / 1e12
            tr, mu = pmc_derived(pmc.get(kname))
            return {'kernel': kern, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': round(ach / peak, 4), 'traffic': tr, 'traffic_unit': 'HBM bytes/launch (rocprofv3 PMC of this build: %s)' % pmc_file,
                    'mfma_util': mu, 'flop_per_launch': flop, 'avg_launch_ms': round(ms_front, 4)}

        clips = BATCH * a.steps * world
        roof = roofline_of(eng.precision, stage_ms['cnn_front'])
        roof['whole_path_tflops'] = round(FLOP_TOTAL * clips / dt / 1e12 / world, 2)
        k = pmc.get('cnn_front_bf16_kernel' if eng.precision == 'bf16x3' else 'cnn_front_kernel') or {}
        if k.get('GRBM_GUI_ACTIVE') and k.get('_ms'):
            roof['shader_clock_mhz_profiled'] = round(k['GRBM_GUI_ACTIVE'] / 8.0 / (k['_ms'] * 1e-3) / 1e6, 0)
        if world == 1 and not a.no_extras and eng.precision == 'bf16x3':
            sus = mfma_sustained(dev)
            roof['peak_sustained'] = {'what': 'dense v_mfma_f32_32x32x16_bf16 on register operands, measured on this GPU just now: '
                                              'zero operands reach the data-sheet peak, random operands are power-limited',
                                      **sus, 'frac_of_sustained_random': round(roof['achieved'] / sus['random']['tflops'], 4)}
        mel_flop = FLOP_MEL_FRAME * plan.total_frames
        mel_ach = mel_flop / (stage_ms['mel'] * 1e-3) / 1e12
        mtr, _ = pmc_derived(pmc.get('mel_frame_kernel'))
        kern_tab = {}
        for stage, ks in (('mel', ['mel_frame_kernel']), ('cnn', ['cnn_front_bf16_kernel'] if eng.precision == 'bf16x3' else ['cnn_front_kernel', 'cnn_back_kernel']),
                          ('selfatt', ['td_fused_bf16_kernel', 'td_proj_bf16_kernel', 'td_layer_bf16_kernel'] if eng.precision == 'bf16x3' else ['td_proj_kernel', 'td_layer_kernel']),
                          ('pool', ['pool_score_bf16_kernel', 'pool_final_kernel'] if eng.precision == 'bf16x3' else ['pool_score_kernel', 'pool_final_kernel'])):
            for kn in ks:
                if kn in pmc:
                    tr, mu = pmc_derived(pmc[kn])
                    kern_tab[kn] = {'stage': stage, 'hbm_bytes_per_launch': tr, 'mfma_util': mu}
        res = {
            'metric': 'clips/sec (10 s, 48 kHz)', 'value': round(clips / dt, 2), 'unit': 'clips/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16x3 (bf16 hi+lo operands, 3 MFMA products per term, f32 accumulate; mel/attention/pooling f32)'
                     if eng.precision == 'bf16x3' else 'f32',
            'data': 'synthetic 48 kHz / 10 s PCM16 clips (SURVEY 8d generator); ' + wdesc,
            'config': {'workload': 'predict_dir nisqa.tar (NISQA_DIM CNN-SA-AP) bs=64 per GPU, 10 s synthetic 48 kHz '
                                   'clips, int16 PCM resident in HBM' + ('' if BATCH == 64 else ' [EXPERIMENT: bs=%d]' % BATCH), 'batch_clips_per_gpu': BATCH, 'streams': len(streams),
                       'precision': eng.precision,
                       'segments_per_batch': int(plan.n_wins.sum()), 'frames_per_batch': plan.total_frames,
                       'parallelism': 'clip-sharded x%d, final all_gather of MOS rows' % world,
                       'collective_backend': backend, 'world_size_seen': world},
            'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()},
            'roofline': roof,
            'roofline_secondary': {'kernel': 'mel_frame_kernel (pruned 4 x FFT-512 + sparse mel bank + dB)', 'bound': 'valu',
                                   'achieved': round(mel_ach, 2), 'peak': PEAK_F32, 'unit': 'TFLOP/s',
                                   'frac': round(mel_ach / PEAK_F32, 4), 'flop_per_launch': mel_flop, 'traffic': mtr,
                                   'avg_launch_ms': round(stage_ms['mel'], 4)},
            'kernels': kern_tab,
        }
        if world == 1 and not a.no_extras:
            # the other precision path on the same workload (secondary measurement, outside the timed region)
            other = 'f32' if eng.precision == 'bf16x3' else 'bf16x3'
            eng2 = HipNisqa(margs, sd, dev, precision=other)
            for _ in range(2):
                o2 = eng2.forward_pcm(pcm, plan, SR)
            torch.cuda.synchronize()
            n2 = max(5, a.steps // 2)
            ev2 = []
            for _ in range(n2):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
                for x in e:
                    x.record()
                ev2.append(e)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s in range(n2):
                o2 = eng2.forward_pcm(pcm, plan, SR, stage_events=ev2[s])
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            ms2 = float(np.mean([ev2[s][1].elapsed_time(ev2[s][2]) for s in range(n2)]))
            r2 = roofline_of(other, ms2)
            res['alt_precision'] = {'precision': other, 'value': round(BATCH * n2 / dt2, 2), 'unit': 'clips/s',
                                    'steps': n2, 'roofline_frac': r2['frac'], 'roofline_achieved': r2['achieved'],
                                    'roofline_peak': r2['peak'],
                                    'max_abs_diff_vs_primary': float((o2 - outs[-1]).abs().max())}
        if world == 1 and len(streams) == 1 and not a.no_extras:
            # same steps alternated over 3 streams (what the predict loop does with 2): launches and copies overlap
            st3 = [torch.cuda.Stream(device=dev) for _ in range(3)]
            for i in range(3):
                with torch.cuda.stream(st3[i]):
                    eng.forward_pcm(pcm, plan, SR)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s in range(a.steps):
                with torch.cuda.stream(st3[s % 3]):
                    eng.forward_pcm(pcm, plan, SR)
            torch.cuda.synchronize()
            res['overlap_3_streams'] = {'value': round(BATCH * a.steps / (time.perf_counter() - t1), 2), 'unit': 'clips/s'}
        if world == 1 and not a.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(margs, sd)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

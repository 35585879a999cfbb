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
  f32                 the same workload, same --steps / --warmup, on the exact-fp32 path (the reference's own arithmetic):
                      value_f32 at the top level, its own roofline (cnn_front_kernel against the fp32-MFMA peak, traffic and
                      mfma_util from the same PMC file).
  steady              the primary path again over >= 400 steps (the driver's 20-step run ends before the clocks settle).
  overlap_2_streams, overlap_3_streams   the same steps alternated over 2 (the predict loop's form) / 3 streams, >= 200 steps, no events.
  stage_ms            cnn_front from the two events of the timed region; the other stages from an untimed second pass (stage_ms_note).
  side                the other BASELINE.json configurations, each bounded to a few seconds, each with its own roofline
                      and cpu_baseline: predict_csv_1gpu (configs[2] through nisqaModel.predict(), WAV files on disk,
                      PCIe-inclusive), predict_dir_bs64 (configs[1] as run_predict.py runs it: a directory of 98 304 names, bs 64, file-fed),
                      tts_mixed (configs[3]: nisqa_tts.tar, mixed 3-30 s clips), train_step (configs[4]:
                      forward + backward + Adam at bs 32).  `--leg main|tts|train|csv|dir` runs one of them alone (profiling).
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
# nisqa_tts.tar (StandardCNN + fc, BiLSTM 128), per segment: conv1 48x15x16x9, conv2 24x8x32x144, conv3 12x4x64x288, conv4
# 12x4x64x576, conv5 / conv6 6x2x64x576 each, fc 768x20 (x2 flop per multiply-add) = 9.085 MFLOP; one BiLSTM step = 2
# directions x 4 gates x 128 units x (20 + 128) inputs x 2 = 0.303 MFLOP
FLOP_STD_SEG = 2.0 * (48 * 15 * 16 * 9 + 24 * 8 * 32 * 144 + 12 * 4 * 64 * 288 + 12 * 4 * 64 * 576 + 2 * 6 * 2 * 64 * 576 + 768 * 20)
FLOP_LSTM_SEG = 2.0 * 2 * 4 * 128 * (20 + 128)
FLOP_NET_CLIP = (51.2 + 382.4 + 546.3 + 1092.6 + 327.8 + 109.3 + 12.1 + 55.5 + 20.7) * 1e6   # CNN + proj + self-att + pooling, 10 s
PEAK_PCIE = 63.0                                             # GB/s host -> device, PCIe Gen5 x16


# what `dtype` says for each precision mode of the engine (arithmetic of EVERY GEMM of the path: AdaptCNN / StandardCNN, the 384 -> 64
# projection, self-attention, pooling; the mel front end is fp32 VALU in all of them)
DTYPE_TEXT = {
    'f32': 'f32 (every GEMM on v_mfma_f32_32x32x2_f32, fp32 operands and accumulation: the reference\'s own arithmetic; mel f32)',
    'bf16x6': 'bf16x6 (every GEMM -- CNN, projection, self-attention, pooling: fp32 operands as three exact bf16 terms, 6 MFMA products per '
              'term pair, f32 accumulate: all 24 operand mantissa bits; mel f32)',
    'bf16x3': 'bf16x3 (every GEMM -- CNN, projection, self-attention, pooling: fp32 operands as bf16 hi + lo, 3 MFMA products per term pair, '
              'f32 accumulate: 16 of the 24 operand mantissa bits, NARROWER than the reference; mel f32)',
    'f16x4': 'f16x4 (AdaptCNN: fp32 operands as f16 hi + lo of the power-of-two-scaled tensors -- 11 + 11 bits and the low term\'s sign: the fp32 '
             'value for ~75 % of the operands, one fp32 ulp off otherwise -- all 4 MFMA products per term pair, f32 accumulate; projection, '
             'self-attention, pooling as bf16x6; mel f32)',
    'f16x3': 'f16x3 (AdaptCNN: fp32 operands as f16 hi + lo of the power-of-two-scaled tensors, 3 MFMA products per term pair (lo*lo dropped: <= 2^-22 '
             'of a product), f32 accumulate; projection, self-attention, pooling as bf16x6; mel f32)',
}
ALL_PRECISIONS = ('f32', 'bf16x6', 'f16x4', 'f16x3', 'bf16x3')
CNN_KERNEL = {'f32': 'cnn_front_kernel', 'bf16x3': 'cnn_front_bf16_kernel', 'bf16x6': 'cnn_front_bf16x6_kernel', 'f16x4': 'cnn_front_f16_kernel',
              'f16x3': 'cnn_front_f16_kernel'}


def default_precision():
    from nisqa_amd.engine import DEFAULT_PRECISION
    return os.environ.get('NISQA_HIP_PRECISION') or DEFAULT_PRECISION


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
    port = {'value': round(bs / dt, 3), 'unit': 'clips/s', 'cores': nthr, 'kind': 'port',
            'network_only': round(bs / t_net, 3), 'mel_only': round(bs / t_mel, 3),
            'sample': 'one bs = %d batch of 10 s clips (8 distinct): oracle.mel (numpy restatement of librosa 0.8.1, FFT '
                      'on 1 thread like the reference dataset at num_workers = 0) %.1f s + oracle.net (torch CPU fp32, %d '
                      'threads, CNN on the packed %d segments) %.1f s; host has %d cores'
                      % (bs, t_mel, nthr, int(sum(n for _, n in segs)), t_net, os.cpu_count())}
    # The reference's OWN loop beside it: NISQA_lib.py as shipped (staged by build() under oracle/_ref,
    # git-ignored) -- SpeechQualityDataset -> get_librosa_melspec -> segment_specs padded to [B, 1300, 1, 48, 15] -> DataLoader
    # (num_workers 0) -> Framewise pack -> model on the CPU -- with librosa's three entry points served by oracle/mel.py
    # (the only part of the path that is not the reference's code: librosa is not installable here; parity unpinned there)
    try:
        from oracle import ref_shim
        if not ref_shim.reference_available():
            raise RuntimeError('NISQA_lib.py not staged under oracle/_ref (run __graft_entry__.build() where /root/reference exists)')
        import shutil
        import tempfile
        d = tempfile.mkdtemp(prefix='nisqa_cpu_ref_')
        try:
            names = []
            for i in range(bs):
                names.append('c%03d.wav' % i)
                synth.write_wav(os.path.join(d, names[-1]), synth.synth_pcm16(2000 + i % 8, SECONDS), SR)
            synth.write_wav(os.path.join(d, 'warm.wav'), synth.synth_pcm16(1, 1.0), SR)
            ck = os.path.join(d, 'model.tar')
            torch.save({'args': dict(args), 'model_state_dict': {k: torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
                                                                  for k, v in sd.items()}}, ck)
            ref_shim.reference_predict(ck, d, ['warm.wav'], bs=1)
            tm = {}
            y_ref = ref_shim.reference_predict(ck, d, names, bs=bs, timings=tm)
        finally:
            shutil.rmtree(d, ignore_errors=True)
        assert y_ref.shape[0] == bs and np.isfinite(y_ref).all()
        return {'value': round(bs / tm['predict_s'], 3), 'unit': 'clips/s', 'cores': nthr, 'kind': 'reference-torch + oracle-mel',
                'sample': 'the reference\'s predict_dim (NISQA_lib.py:1441-1467, its own DataLoader / SpeechQualityDataset / '
                          'segment_specs padding to [%d, 1300, 1, 48, 15] / Framewise pack / modules) on ONE bs = %d batch of 10 s WAV '
                          'files, device cpu, num_workers 0, %d torch threads, %.1f s; librosa.load / melspectrogram / amplitude_to_db '
                          'served by oracle/mel.py (librosa 0.8.1 is not installable: mel stage parity unpinned); host has %d cores'
                          % (bs, bs, nthr, tm['predict_s'], os.cpu_count()),
                'port': port}
    except Exception as e:                                            # the port alone, and why
        port['reference_loop_error'] = repr(e)[:300]
        return port


def self_launch(gpus):
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run (WORLD_SIZE unset): start the N ranks ourselves --
    the command the contract names, one rank per GPU, rendezvous on 127.0.0.1 at a free port -- pass our argv through, and
    hand rank 0's single JSON line to our own stdout.  Returns the exit code of the launch (0 = all ranks finished).
    Replaces what nn.DataParallel did inside one process (reference nisqa/NISQA_model.py:56-57)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // gpus)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' stdout is read here: JSON lines go to our stdout (rank 0 prints exactly one), anything else a library wrote there
    # (gloo's connection banner) to stderr -- a caller that parses stdout sees the bench line and nothing else
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for ln in p.stdout:
        out = sys.stdout if ln.lstrip().startswith('{') else sys.stderr
        out.write(ln)
        out.flush()
    return p.wait()


def init_dist(a):
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus and a.gpus > 1:
        raise SystemExit('bench.py --gpus %d inside a WORLD_SIZE=%d launch: start it as `python bench.py --gpus %d` (it launches its '
                         'own ranks) or under torch.distributed.run --nproc-per-node %d' % (a.gpus, world, a.gpus, a.gpus))
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


LAST_JOB_TIMING = {}          # nisqaModel.timing of the last predict_csv_job: seconds scoring / seconds writing the table


def predict_csv_job(rank, world, dev, backend, clips, bs, distinct, workers, warmup, tmp_dir=None, tag='csv', mode='predict_csv'):
    """BASELINE configs[2]: nisqaModel(predict_csv).predict() over `clips` rows of a CSV (`distinct` synthetic 10 s WAV files
    on local disk, reused cyclically), ranks shard the CSV.  -> (seconds max over ranks, DataFrame on rank 0, description).
    mode='predict_dir' (BASELINE configs[1] as run_predict.py runs it): the same job over a DIRECTORY of `clips` *.wav names --
    hard links to the `distinct` files (one inode each: the page cache holds `distinct` files, the directory listing, the RIFF
    parse and the read of every name are real)."""
    import contextlib
    import io
    import shutil
    import tempfile
    import pandas as pd
    from nisqa_amd.NISQA_model import nisqaModel
    margs, sd, wdesc = model_weights()
    d = os.path.join(tmp_dir or tempfile.gettempdir(), 'nisqa_bench_' + tag)
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        for i in range(distinct):
            synth.write_wav(os.path.join(d, 'c%05d.wav' % i), synth.synth_pcm16(3000 + i, SECONDS), SR)
        pd.DataFrame({'deg': ['c%05d.wav' % (i % distinct) for i in range(clips)]}).to_csv(os.path.join(d, 'list.csv'), index=False)
        if mode == 'predict_dir':
            for sub, n in (('dir', clips), ('warm', max(1, warmup) * bs * world)):
                os.makedirs(os.path.join(d, sub))
                for i in range(n):
                    os.link(os.path.join(d, 'c%05d.wav' % (i % distinct)), os.path.join(d, sub, 'f%06d.wav' % i))
        ck = dict(margs)
        ck.update({'pretrained_model': False, 'tr_bs_val': bs, 'tr_num_workers': workers})
        torch.save({'args': ck, 'model_state_dict': sd}, os.path.join(d, 'model.tar'))
        # warm-up: W batches per rank through the same path (page cache, engine, pinned ring)
        pd.DataFrame({'deg': ['c%05d.wav' % (i % distinct) for i in range(max(1, warmup) * bs * world)]}).to_csv(
            os.path.join(d, 'warm.csv'), index=False)
    if world > 1:
        torch.distributed.barrier()

    def args_for(csv):
        if mode == 'predict_dir':
            return {'mode': 'predict_dir', 'pretrained_model': os.path.join(d, 'model.tar'), 'deg': None,
                    'data_dir': os.path.join(d, 'warm' if csv == 'warm.csv' else 'dir'), 'output_dir': None, 'csv_file': None,
                    'csv_deg': None, 'num_workers': workers, 'bs': bs, 'ms_channel': None, 'tr_bs_val': bs, 'tr_num_workers': workers}
        return {'mode': 'predict_csv', 'pretrained_model': os.path.join(d, 'model.tar'), 'deg': None, 'data_dir': d,
                'output_dir': None, 'csv_file': csv, 'csv_deg': 'deg', 'num_workers': workers, 'bs': bs,
                'ms_channel': None, 'tr_bs_val': bs, 'tr_num_workers': workers}

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
    LAST_JOB_TIMING.clear()
    LAST_JOB_TIMING.update(getattr(m, 'timing', {}))
    if world > 1:
        torch.distributed.barrier()
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        assert len(df) == clips and np.isfinite(df['mos_pred'].to_numpy()).all()
        shutil.rmtree(d, ignore_errors=True)
    return dt, df, wdesc


def bench_predict_csv(a):
    """`--workload predict_csv`: the configs[2] harness as its own bench line (strong scaling over a fixed CSV)."""
    rank, world, dev, backend = init_dist(a)
    dt, df, wdesc = predict_csv_job(rank, world, dev, backend, a.clips, a.bs, a.distinct, a.workers, a.warmup, a.tmp_dir)
    if rank == 0:
        steps = -(-(-(-a.clips // world)) // a.bs)
        print(json.dumps({
            'metric': 'clips/sec (10 s, 48 kHz)', 'value': round(a.clips / dt, 2), 'unit': 'clips/s', 'n_gpus': world,
            'steps': steps, 'warmup': max(1, a.warmup), 'ms_per_step': round(1e3 * dt / steps, 4), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None,
            'dtype': DTYPE_TEXT[default_precision()],
            'data': 'synthetic 48 kHz / 10 s PCM16 WAV files on local disk (%d distinct, reused cyclically), %s' % (a.distinct, wdesc),
            'config': {'workload': 'predict_csv nisqa.tar bs=%d per GPU, %d synthetic 10 s 48 kHz clips, clip-sharded over '
                                   '%d rank(s); WAV files -> native ingest -> H2D -> kernels -> all_gather (PCIe-inclusive)'
                                   % (a.bs, a.clips, world),
                       'clips': a.clips, 'bs': a.bs, 'distinct_files': a.distinct, 'ingest_workers': a.workers,
                       'parallelism': 'clip-sharded x%d, final all_gather of MOS rows' % world,
                       'collective_backend': backend, 'world_size_seen': world},
            'link': {'bound': 'pcie', 'achieved': round(a.clips * SECONDS * SR * 2 / dt / 1e9, 2), 'peak': PEAK_PCIE, 'unit': 'GB/s',
                     'frac': round(a.clips * SECONDS * SR * 2 / dt / 1e9 / PEAK_PCIE / world, 4)}}))
    if world > 1:
        torch.distributed.destroy_process_group()


# ---- side legs: the other BASELINE.json configurations on the driver-run line ---------------------------------------
def _stage_events(n, only_cnn=False):
    """six events per step (all stage boundaries), or -- for a TIMED region -- only the two around the CNN kernel: an event record
    costs ~5 us of stream time (profiles/r06_stage_event_cost.txt)"""
    ev = []
    for _ in range(n):
        e = [torch.cuda.Event(enable_timing=True) if (not only_cnn or i in (1, 2)) else None for i in range(6)]
        for x in e:
            if x is not None:
                x.record()
        ev.append(e)
    return ev


def _stage_means(ev, names):
    return {n: float(np.mean([e[i].elapsed_time(e[i + 1]) for e in ev])) for i, n in enumerate(names) if ev[0][i] is not None and ev[0][i + 1] is not None}


def link_only_probe(dev, nbytes_per_copy=245_760_000, copies=24):
    """H2D rate of the predict loop's own transport with nothing else going on: page-locked buffers of one batch's size
    (256 clips x 10 s x 48 kHz x 2 bytes) copied on the loop's copy stream (high priority, carries no kernels -> SDMA engine),
    three buffers in rotation like the staging ring.  What the link itself allows on this box; the loop's loss is measured
    against it, the roofline fraction against the 63 GB/s of the data sheet."""
    from nisqa_amd import NISQA_lib as NL
    copy_stream, _ = NL._loop_streams(dev)
    host = [torch.empty(nbytes_per_copy, dtype=torch.uint8, pin_memory=True) for _ in range(3)]
    dst = [torch.empty(nbytes_per_copy, dtype=torch.uint8, device=dev) for _ in range(2)]
    for h in host:
        h.numpy()[::4096] = 1                                         # touch every page
    with torch.cuda.stream(copy_stream):
        for i in range(3):
            dst[i % 2].copy_(host[i % 3], non_blocking=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(copy_stream)
        for i in range(copies):
            dst[i % 2].copy_(host[i % 3], non_blocking=True)
        e1.record(copy_stream)
    e1.synchronize()
    return nbytes_per_copy * copies / (e0.elapsed_time(e1) * 1e-3) / 1e9


def side_predict_dir(dev, cpu, link):
    """configs[1] AS A USER RUNS IT (run_predict.py --mode predict_dir --bs 64, reference run_predict.py:20-39, NISQA_model.py:745-777):
    nisqaModel(predict_dir, bs 64).predict() over a directory of 98 304 ten-second WAV files (64 inodes in the page cache, 98 304 names:
    the size of the csv leg, so that the loop's fill and drain weigh the same in both),
    file-fed and PCIe-inclusive like the csv leg; the contract `value` above is the same batch shape with the PCM already in HBM."""
    clips, bs = 98304, 64
    from nisqa_amd import ingest as _ing
    from nisqa_amd import NISQA_lib as NL
    dt, df, _ = predict_csv_job(0, 1, dev, None, clips, bs, 64, 0, 4, tag='side_dir', mode='predict_dir')
    loop = dict(NL.LOOP_STATS)
    timing = dict(LAST_JOB_TIMING)
    gbs = clips * SECONDS * SR * 2 / dt / 1e9
    return {'config': 'configs[1] predict_dir nisqa.tar bs=64, 1 GPU, a directory of %d *.wav names (hard links to 64 distinct 10 s files '
                      'in the page cache), nisqaModel.predict() end to end incl. directory listing, DataFrame and the printed table; '
                      'reader threads: the CPU budget (%d) - 3' % (clips, _ing.cpu_budget()),
            'value': round(clips / dt, 1), 'unit': 'clips/s', 'seconds': round(dt, 3), 'pcie_inclusive': True,
            'roofline': {'bound': 'pcie', 'kernel': 'H2D copy of the int16 PCM (SDMA, copy-only stream)', 'achieved': round(gbs, 2),
                         'peak': PEAK_PCIE, 'unit': 'GB/s', 'frac': round(gbs / PEAK_PCIE, 4),
                         'link_only_GBps': round(link, 2), 'frac_of_link_only': round(gbs / link, 4)},
            'predict_s': round(timing.get('predict_s', float('nan')), 3), 'table_s': round(timing.get('table_s', float('nan')), 3),
            'loop_host_s': {k: round(v, 3) for k, v in loop.items()},
            'cpu_baseline': cpu and {k: cpu[k] for k in ('value', 'unit', 'cores', 'kind')}}


def side_predict_csv(dev, cpu):
    """configs[2] on ONE GPU: 98 304 rows / 64 distinct 10 s WAV files, bs 256, through nisqaModel.predict() (file list ->
    native ingest -> pinned ring -> H2D -> kernels -> DataFrame -> the printed table): a ~2 s job, so that start-up (engine,
    first batch, page-locking) is not what is measured.  PCIe-inclusive: the bound is the host link (2 bytes per sample
    cross it), not the kernels.  link_only_GBps: the same transport with no loop around it (link_only_probe), so that loop
    loss and link loss separate; print_s: what formatting the 98 304-row table costs inside the timed call."""
    clips, bs = 98304, 256
    from nisqa_amd import ingest as _ing
    from nisqa_amd import NISQA_lib as NL
    link = link_only_probe(dev)
    dt, df, _ = predict_csv_job(0, 1, dev, None, clips, bs, 64, 0, 2, tag='side_csv')
    loop = dict(NL.LOOP_STATS)
    timing = dict(LAST_JOB_TIMING)
    from nisqa_amd.NISQA_model import frame_to_string
    t0 = time.perf_counter()
    frame_to_string(df)
    t_print = time.perf_counter() - t0
    gbs = clips * SECONDS * SR * 2 / dt / 1e9
    return {'config': 'configs[2] predict_csv nisqa.tar bs=256, 1 GPU, %d rows (64 distinct 10 s WAV files in the page cache), '
                      'nisqaModel.predict() end to end incl. file listing, DataFrame and the printed table; reader threads: '
                      'the CPU budget (%d) - 3' % (clips, _ing.cpu_budget()),
            'value': round(clips / dt, 1), 'unit': 'clips/s', 'seconds': round(dt, 3), 'pcie_inclusive': True,
            'roofline': {'bound': 'pcie', 'kernel': 'H2D copy of the int16 PCM (SDMA, copy-only stream)', 'achieved': round(gbs, 2),
                         'peak': PEAK_PCIE, 'unit': 'GB/s', 'frac': round(gbs / PEAK_PCIE, 4),
                         'link_only_GBps': round(link, 2), 'frac_of_link_only': round(gbs / link, 4)},
            'print_s': round(t_print, 3),
            'print_note': 'print_s: formatting the whole table AFTER the run, as rounds 3-4 did inside the timed call; since round 5 the cells '
                          'are formatted batch by batch inside the loop (under the next batches\' transfers) and table_s is what is left behind '
                          'the last batch in the timed call',
            'predict_s': round(timing.get('predict_s', float('nan')), 3), 'table_s': round(timing.get('table_s', float('nan')), 3),
            'loop_host_s': {k: round(v, 3) for k, v in loop.items()},
            'cpu_baseline': cpu and {k: cpu[k] for k in ('value', 'unit', 'cores', 'kind')}}


def tts_weights():
    p = find_weights('nisqa_tts.tar')
    if p:
        ck = torch.load(p, map_location='cpu', weights_only=False)
        return ck['args'], ck['model_state_dict'], 'nisqa_tts.tar (%s)' % os.path.relpath(p, ROOT)
    return dict(synth.TTS_ARGS), synth.random_state_dict(9, 'NISQA_TTS'), 'random-init nisqa_tts.tar architecture'


def side_tts(dev, reps, cpu_baseline_on, pmc, two_streams=True):
    """configs[3]: nisqa_tts.tar (StandardCNN + fc -> BiLSTM(128) -> last-step pooling, segment hop 1), 256 clips with
    durations rng(7).uniform(3, 30) s, int16 PCM resident in HBM, batched by the predict loop's own policy
    (NISQA_lib.batch_policy: sorted by length, >= 128 clips per batch, <= 256 MiB of PCM)."""
    from nisqa_amd.engine import HipNisqa
    from nisqa_amd import NISQA_lib as NL, ingest as _ing
    targs, tsd, wdesc = tts_weights()
    eng = HipNisqa(targs, tsd, dev, precision=default_precision())
    n_clips = 256
    durs = np.random.default_rng(7).uniform(3, 30, n_clips)
    base = synth.synth_pcm16(5, 30.0)
    frames = (durs * SR).astype(np.int64)

    class _DS(object):                                       # the header fields batch_policy looks at
        ms_hop_length, seg_length, seg_hop_length = targs['ms_hop_length'], targs['ms_seg_length'], targs['ms_seg_hop_length']
    pol = NL.batch_policy(eng, _DS, range(n_clips), 1)
    cuts = pol.cut(frames, np.full(n_clips, SR, np.int64), np.full(n_clips, 2, np.int64))
    batches = []
    for c in cuts:
        plan = eng.plan([int(frames[k]) for k in c], SR)
        x = torch.from_numpy(np.concatenate([base[:int(frames[k])] for k in c])).to(dev)
        plan.to(dev)
        batches.append((plan, x))
    for plan, x in batches:
        eng.forward_pcm(x, plan, SR)
    torch.cuda.synchronize()
    ev = _stage_events(reps * len(batches))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for bi, (plan, x) in enumerate(batches):
            out = eng.forward_pcm(x, plan, SR, stage_events=ev[r * len(batches) + bi])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert torch.isfinite(out).all()
    # the same job with the batches alternating over two streams, as the predict loop runs them (the BiLSTM of one batch --
    # one workgroup per (clip, direction), latency-bound -- under the mel + CNN of the next)
    # (skipped under `--leg tts --no-extras`, the rocprofv3 kernel-statistics pass: kernels of two streams run concurrently and
    # stretch each other's durations -- profiles/r03_tts_kernel_stats.csv was 18 % above the stage events because of it)
    dt2 = None
    if two_streams:
        st2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for r in range(reps):
            for bi, (plan, x) in enumerate(batches):
                with torch.cuda.stream(st2[bi % 2]):
                    eng.forward_pcm(x, plan, SR)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t1) / reps
    nb = len(batches)
    segs = np.array([int(p.n_wins.sum()) for p, _ in batches], dtype=np.float64)
    steps = np.array([int(p.n_wins.max()) for p, _ in batches], dtype=np.float64)
    ms = np.array([[np.mean([ev[r * nb + bi][i].elapsed_time(ev[r * nb + bi][i + 1]) for r in range(reps)]) for i in range(4)]
                   for bi in range(nb)])                     # [batch][mel, cnn, (unused), lstm + pool]
    cnn_ms, lstm_ms = float(ms[:, 1].sum()), float(ms[:, 3].sum())
    ach = float(segs.sum()) * FLOP_STD_SEG / (cnn_ms * 1e-3) / 1e12
    kname = {'bf16x3': 'cnn_std_bf16_kernel', 'bf16x6': 'cnn_std_bf16x6_kernel', 'f16x4': 'cnn_std_f16_kernel', 'f16x3': 'cnn_std_f16_kernel'}.get(eng.precision, 'cnn_std_kernel')
    kdesc = {'bf16x3': 'split-bf16 MFMA: 3 products per term pair', 'bf16x6': 'three exact bf16 terms per fp32 operand: 6 MFMA products per term pair',
             'f16x4': 'two f16 terms per fp32 operand of the scaled tensors: 4 MFMA products per term pair',
             'f16x3': 'two f16 terms per fp32 operand of the scaled tensors: 3 MFMA products per term pair', 'f32': 'fp32 MFMA'}[eng.precision]
    kpeak = PEAK_F32 if eng.precision == 'f32' else PEAK_BF16_MFMA
    tr, mu = pmc_derived(pmc.get('tts:' + kname))
    res = {'config': 'configs[3] predict_dir nisqa_tts.tar (Naturalness head), %d clips, durations rng(7).uniform(3, 30) s '
                     '(%.0f s of audio, %d segments at hop 1), int16 PCM resident in HBM, %d length-sorted batches of %s clips '
                     '(NISQA_lib.batch_policy), one stream; weights: %s'
                     % (n_clips, float(durs.sum()), int(segs.sum()), nb, '/'.join(str(p.n_clips) for p, _ in batches), wdesc),
           'value': round(n_clips / dt, 1), 'unit': 'clips/s', 'audio_seconds_per_s': round(float(durs.sum()) / dt, 1),
           'ms_per_job': round(dt * 1e3, 3), 'value_2_streams': round(n_clips / dt2, 1) if dt2 else None,
           'stage_ms': {'mel': round(float(ms[:, 0].sum()), 4), 'cnn_std': round(cnn_ms, 4), 'lstm_pool': round(lstm_ms, 4)},
           'precision': eng.precision, 'dtype': DTYPE_TEXT[eng.precision] + '; BiLSTM and last-step pooling fp32 VALU',
           'roofline': {'kernel': '%s (StandardCNN conv1-6 + fc, %s)' % (kname, kdesc),
                        'bound': 'mfma', 'achieved': round(ach, 2), 'peak': kpeak, 'unit': 'TFLOP/s',
                        'frac': round(ach / kpeak, 4), 'flop_per_job': float(segs.sum()) * FLOP_STD_SEG,
                        'avg_launch_ms': round(cnn_ms / nb, 4), 'launches_per_job': nb, 'traffic': tr, 'mfma_util': mu},
           'lstm': {'kernel': 'lstm_dir_kernel + pool_last_kernel: one 512-thread workgroup per (clip, direction), sequential in time',
                    'bound': 'latency', 'us_per_step': round(1e3 * lstm_ms / float(steps.sum()), 4),
                    'steps_per_job': int(steps.sum()), 'workgroups_per_launch': [2 * p.n_clips for p, _ in batches],
                    'achieved_tflops': round(float(segs.sum()) * FLOP_LSTM_SEG / (lstm_ms * 1e-3) / 1e12, 3)}}
    if two_streams:
        # the same job in the other precision modes: the exact-fp32 kernels (the reference's arithmetic), 'bf16x6' (fp32 operands as
        # three exact bf16 terms, six products: held to the bounds of 'f32' by the parity tests), 'bf16x3' (two terms: the fast mode);
        # the BiLSTM is fp32 VALU in every mode
        for prec in [q for q in ALL_PRECISIONS if q != eng.precision]:
            e2 = HipNisqa(targs, tsd, dev, precision=prec)
            for plan, x in batches:
                o2 = e2.forward_pcm(x, plan, SR)
            torch.cuda.synchronize()
            r2 = max(1, min(reps, 8))
            t1 = time.perf_counter()
            for r in range(r2):
                for plan, x in batches:
                    o2 = e2.forward_pcm(x, plan, SR)
            torch.cuda.synchronize()
            d2 = (time.perf_counter() - t1) / r2
            res['value_' + prec] = round(n_clips / d2, 1)
            res[prec] = {'value': round(n_clips / d2, 1), 'unit': 'clips/s', 'ms_per_job': round(d2 * 1e3, 3), 'jobs_timed': r2,
                         'max_abs_diff_vs_primary_last_batch': float((o2 - out).abs().max())}
            del e2
    if cpu_baseline_on:
        from oracle import mel as omel, net as onet
        n_cpu = 4
        clip = base[:10 * SR].astype(np.float32) / np.float32(32768.0)
        omel.melspec_db_from_audio(clip[:SR], SR, fmax=targs['ms_fmax'])           # builds the filter bank once
        t0 = time.perf_counter()
        specs = [omel.melspec_db_from_audio(np.roll(clip, 1000 * i), SR, fmax=targs['ms_fmax']) for i in range(n_cpu)]
        t_mel = time.perf_counter() - t0
        t1 = time.perf_counter()
        for spec in specs:
            onet.predict_from_melspec(tsd, targs, spec)
        t_net = time.perf_counter() - t1
        aud = 10.0 * n_cpu
        res['cpu_baseline'] = {'value': round(aud / (t_mel + t_net), 3), 'unit': 'audio-seconds/s', 'cores': int(torch.get_num_threads()),
                               'kind': 'port', 'clips_per_s_at_mean_duration': round(aud / (t_mel + t_net) / float(durs.mean()), 3),
                               'sample': '%d clips of 10 s (987 segments each), one after the other like predict_mos at bs 1: '
                                         'oracle.mel %.2f s + oracle.net (StandardCNN on the packed segments, BiLSTM, torch CPU fp32) '
                                         '%.2f s; the GPU job is %.0f audio-seconds' % (n_cpu, t_mel, t_net, float(durs.sum()))}
    return res


def side_train(dev, steps, cpu_baseline_on, pmc):
    """configs[4]: train_nisqa_cnn_sa_ap.yaml's step (NISQA_model.py:131-152) at bs 32 x 10 s: mel front end, forward in
    train mode, bias-aware loss, backward, BatchNorm buffers, Adam -- HipTrainer.  The PRIMARY number of the leg is the
    'f32' mode (every convolution on exact fp32 MFMA: the reference's arithmetic); 'mixed' (HipTrainer's default until round 4: fp32
    forward, split-bf16 gradient convolutions), 'bf16x3' (split-bf16 forward too) and 'bf16x6' (every convolution at fp32
    OPERAND precision on the bf16 matrix pipe: three exact bf16 terms per operand, six products -- held to the bounds of 'f32'
    by the parity tests) are reported beside it."""
    from nisqa_amd.train import HipTrainer
    bs = 32
    args = dict(synth.MOS_ARGS)                               # model NISQA, cnn_dropout 0.2, td_sa_dropout 0.1 (the yaml's values)
    sd = synth.random_state_dict(8, 'NISQA')
    pcm = np.concatenate([synth.synth_pcm16(i % 8, SECONDS) for i in range(bs)])
    y = np.random.default_rng(9).uniform(1, 5, (bs, 1)).astype(np.float32)
    flop = 3.0 * FLOP_NET_CLIP * bs                            # forward + input gradients + weight gradients
    cnn = (FLOP_CONV1_4 + FLOP_CONV5_6) * bs
    rest = FLOP_NET_CLIP * bs - cnn
    modes, segments, peak_mem = {}, 0, 0.0
    for mode in ('f32', 'bf16x6', 'f16x4', 'mixed', 'bf16x3'):
        tr = HipTrainer(args, sd, dev, lr=1e-3, precision=mode)
        plan = tr.eng.plan([int(SECONDS * SR)] * bs, SR)
        x = tr.eng.pcm16_to_f32(torch.from_numpy(pcm).to(dev))
        for _ in range(3):
            tr.step_pcm(x, plan, SR, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.step_pcm(x, plan, SR, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        assert np.isfinite(float(loss))
        # the ideal time of this precision mode: forward convolutions on fp32 MFMA ('mixed', 'f32') or split-bf16 ('bf16x3'),
        # the two gradient passes on split-bf16 ('mixed', 'bf16x3') or fp32 ('f32'); attention / pooling / projection fp32
        # ('bf16x6': all three on the bf16 pipe with six products per term pair instead of three)
        pk_f, pk_b = (PEAK_F32 if mode in ('f32', 'mixed') else PEAK_BF16_MFMA), (PEAK_F32 if mode == 'f32' else PEAK_BF16_MFMA)
        # ('f16x4': forward and input gradient with four products per term pair instead of three, the weight gradient with six)
        ideal = (((2 if mode == 'bf16x6' else 1) * (cnn / pk_f + 2 * cnn / pk_b) if mode != 'f16x4' else (4.0 / 3 * 2 + 2) * cnn / PEAK_BF16_MFMA) + 3 * rest / PEAK_F32) / 1e12
        ach = flop / dt / 1e12
        modes[mode] = {'ms_per_step': round(dt * 1e3, 3), 'value': round(bs / dt, 1), 'achieved_TFLOPs': round(ach, 2),
                       'frac_of_fp32_peak': round(ach / PEAK_F32, 4), 'frac_of_mode_ideal': round(ideal / dt, 4),
                       'mode_ideal_ms': round(ideal * 1e3, 4), 'loss': round(float(loss), 5)}
        segments = int(plan.n_wins.sum())
        peak_mem = max(peak_mem, torch.cuda.max_memory_allocated() / 2 ** 30)
        del tr
    p = modes['f32']
    res = {'config': 'configs[4] train_nisqa_cnn_sa_ap.yaml step (forward + backward + Adam, mel front end inside the step), '
                     'bs=32 x 10 s, %d segments, model NISQA random-init, dropout on; precision mode \'f32\' (exact fp32 MFMA '
                     'everywhere: the reference\'s arithmetic); HipTrainer\'s default mode \'bf16x6\' (fp32-grade on the bf16 matrix pipe), \'mixed\' and \'bf16x3\' beside it' % segments,
           'value': p['value'], 'unit': 'clips/s', 'ms_per_step': p['ms_per_step'], 'steps': steps, 'loss': p['loss'],
           'roofline': {'kernel': 'whole step (~59 launches, profiles/rNN_train_*_kernel_stats.csv; no single kernel dominates)',
                        'bound': 'mfma', 'achieved': p['achieved_TFLOPs'], 'unit': 'TFLOP/s',
                        'flop_per_step': flop, 'flop_rule': '3 x forward network FLOPs (2.598 GFLOP per 10 s clip)',
                        'peak': PEAK_F32, 'frac': p['frac_of_fp32_peak'],
                        'peak_note': 'fp32 MFMA peak (the reference trains in fp32); frac_of_mode_ideal prices the forward / '
                                     'gradient convolutions at the MFMA peak of the operand type each runs on in that mode',
                        'frac_of_mode_ideal': p['frac_of_mode_ideal'], 'mode_ideal_ms': p['mode_ideal_ms']},
           'other_precision_modes': {k: v for k, v in modes.items() if k != 'f32'},
           'peak_mem_GB': round(peak_mem, 2)}
    if cpu_baseline_on:
        from oracle import mel as omel, net as onet, train as otrain, ref_shim
        ref_base = None
        if ref_shim.reference_available():
            # the reference's OWN step (NISQA_model.py:96-152: its dataset + DataLoader + modules in train mode + biasLoss + Adam) at
            # configs[4]'s own batch size, one step; librosa's three entry points served by oracle/mel.py
            import shutil
            import tempfile
            d = tempfile.mkdtemp(prefix='nisqa_bench_train_')
            try:
                names = []
                for i in range(bs):
                    synth.write_wav(os.path.join(d, 't%02d.wav' % i), synth.synth_pcm16(i % 8, SECONDS), SR)
                    names.append('t%02d.wav' % i)
                targs = dict(args)
                targs.update({'ms_sr': None, 'ms_fmax': synth.MOS_ARGS.get('ms_fmax', 20000), 'model': 'NISQA'})
                r = ref_shim.reference_train_step(targs, sd, d, names, y[:, 0], bs, lr=1e-3, steps=2)
                ref_base = {'value': round(bs / r['seconds'][-1], 3), 'unit': 'clips/s', 'cores': int(torch.get_num_threads()),
                            'kind': 'reference-torch + oracle-mel',
                            'sample': 'the SECOND of two steps of the reference\'s own loop (the first one carries DataLoader start-up, table construction and the first Adam state: %.1f s) at bs = %d x 10 s (%d segments; SpeechQualityDataset -> DataLoader -> '
                                      'NISQA.forward in train mode -> biasLoss.get_loss -> backward -> Adam.step, NISQA_model.py:96-152), device '
                                      'cpu, %d torch threads, %.1f s, loss %.4f; librosa served by oracle/mel.py (mel stage parity unpinned)'
                                      % (r['seconds'][0], bs, r['segments'], int(torch.get_num_threads()), r['seconds'][-1], r['loss'])}
            except Exception as e:                            # noqa: BLE001  (the port below still gives a baseline)
                ref_base = {'error': '%s: %s' % (type(e).__name__, e)}
            finally:
                shutil.rmtree(d, ignore_errors=True)
        nb = 8
        clips = [synth.synth_pcm16(i, SECONDS).astype(np.float32) / np.float32(32768.0) for i in range(nb)]
        t0 = time.perf_counter()
        specs = [omel.melspec_db_from_audio(c, SR) for c in clips]
        segs = [onet.segment_specs(sp, args['ms_seg_length'], args['ms_seg_hop_length'], None) for sp in specs]
        xs = torch.cat([onet._t(sg)[:n] for sg, n in segs], 0)            # the valid segments, clip after clip
        torch.manual_seed(0)
        out = otrain.train_step(sd, args, xs, np.array([n for _, n in segs]), y[:nb], lr=1e-3)
        t = time.perf_counter() - t0
        port = {'value': round(nb / t, 3), 'unit': 'clips/s', 'cores': int(torch.get_num_threads()), 'kind': 'port',
                'sample': 'ONE step at bs = %d x 10 s (%d segments): oracle.mel + oracle.train.train_step '
                          '(torch CPU autograd fp32 + Adam) %.2f s, loss %.4f' % (nb, sum(n for _, n in segs), t, out['loss'])}
        if ref_base is not None and 'value' in ref_base:
            res['cpu_baseline'] = dict(ref_base, port=port)
        else:
            res['cpu_baseline'] = dict(port, reference_error=(ref_base or {}).get('error', 'reference NISQA_lib.py not staged (oracle/_ref)'))
    return res


def main():
    global BATCH
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # a step is ~1 ms: 50 + 400 steps are under half a second of GPU time and let the clocks settle (20 steps after 3
    # warm-up steps measure ~7 % low)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default=None, choices=list(ALL_PRECISIONS))
    ap.add_argument('--no-extras', action='store_true', help='skip the alt-precision and 3-stream passes (profiling runs)')
    ap.add_argument('--streams', type=int, default=1, help='HIP streams the steps alternate over')
    ap.add_argument('--workload', default='predict_dir', choices=['predict_dir', 'predict_csv'])
    ap.add_argument('--clips', type=int, default=100000, help='predict_csv: clips in the CSV (whole job)')
    ap.add_argument('--bs', type=int, default=256, help='predict_csv: batch size per GPU')
    ap.add_argument('--distinct', type=int, default=256, help='predict_csv: distinct WAV files written')
    ap.add_argument('--workers', type=int, default=16, help='predict_csv: native ingest threads per rank')
    ap.add_argument('--tmp-dir', default=None)
    ap.add_argument('--batch', type=int, default=BATCH, help='clips per step (experiments; the contract line is 64)')
    ap.add_argument('--leg', default='all', choices=['all', 'main', 'tts', 'train', 'csv', 'dir'],
                    help='profiling runs: only this part (main = the contract workload without side legs)')
    ap.add_argument('--no-side', action='store_true', help='skip the side legs (configs[2], [3], [4])')
    a = ap.parse_args()
    if a.gpus > 1 and int(os.environ.get('WORLD_SIZE', '1')) == 1:   # plain `python bench.py --gpus N`: launch the ranks ourselves
        sys.exit(self_launch(a.gpus))
    BATCH = a.batch
    if a.precision:                                           # every leg of this run (tts, predict_csv) follows it
        os.environ['NISQA_HIP_PRECISION'] = a.precision
    if a.workload == 'predict_csv':
        return bench_predict_csv(a)

    rank, world, dev, backend = init_dist(a)
    if a.leg in ('tts', 'train', 'csv', 'dir'):               # one side leg alone (rocprofv3 runs)
        pmc, _ = pmc_kernels()
        r = (side_tts(dev, max(1, a.steps // 4), not a.no_cpu_baseline, pmc, two_streams=not a.no_extras) if a.leg == 'tts' else
             side_train(dev, a.steps, not a.no_cpu_baseline, pmc) if a.leg == 'train' else
             side_predict_dir(dev, None, link_only_probe(dev)) if a.leg == 'dir' else side_predict_csv(dev, None))
        print(json.dumps({'leg': a.leg, **r}))
        return
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

    # The sustained-MFMA probe of the roofline (peak_sustained, ~100 ms of dense MFMA) runs BEFORE the warm-up steps, on every
    # rank: the power management raises the clocks over the first ~20 ms of load after any quiet period, and a 20-step timed
    # region (16 ms) behind 5 warm-up steps would be measured on that ramp -- 4 % below the rate of every later step
    # (tools/probe_clock_ramp2.py: 0.746 against 0.718 ms per step; `steady` below is the cross-check).
    # NISQA_BENCH_PROBE_LAST=1 restores the old order (probe after the timed region).
    sus, probe_first = None, (not a.no_extras and os.environ.get('NISQA_BENCH_PROBE_LAST') != '1')

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, a.streams))]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
    # Events of the TIMED region: only the two around the dominant kernel (the roofline's live duration).  An event record costs
    # stream time -- six per step read 60.5 k clips/s where the same steps without events read 62.2 k (profiles/
    # r06_stage_event_cost.txt) --, so the other stage times come from a second pass of the same K steps right behind the timed
    # region (evs_all, untimed).
    evs, evs_all = [], []                                # (created before the warm-up: no host work between it and the timed region)
    for _ in range(a.steps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        for x in e:
            x.record(streams[len(evs) % len(streams)])   # forces handle creation; re-recorded inside the library
        evs_all.append(e)
        e = [None] + [torch.cuda.Event(enable_timing=True) for _ in range(2)] + [None] * 3
        for x in e[1:3]:
            x.record(streams[len(evs) % len(streams)])
        evs.append(e)
    if probe_first:                                      # (behind the event set-up: nothing but the warm-up steps between the probe and the timed region;
        sus = mfma_sustained(dev)                        #  1.5 % more than with the 1 ms of event creation in between, same box)
    for i in range(max(a.warmup, len(streams))):
        with torch.cuda.stream(streams[i % len(streams)]):
            out = eng.forward_pcm(pcm, plan, SR)
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
        if backend == 'gloo':               # (the shared-GPU test knob: gloo gathers host tensors)
            rows = rows.cpu()
        parts = [torch.empty_like(rows) for _ in range(world)]
        torch.distributed.all_gather(parts, rows)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        torch.distributed.barrier()
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        if os.environ.get('NISQA_BENCH_KO') != '1':           # knock-out builds (tools/ab_build.sh -DNQ_KO=..) compute garbage
            assert torch.isfinite(outs[-1]).all()
        names = ['mel', 'cnn_front', 'cnn_back', 'selfatt', 'pool']
        for s in range(a.steps):                              # second pass, untimed: all six stage events
            with torch.cuda.stream(streams[s % len(streams)]):
                eng.forward_pcm(pcm, plan, SR, stage_events=evs_all[s])
        torch.cuda.synchronize()
        stage_ms = {n: float(np.mean([evs_all[s][i].elapsed_time(evs_all[s][i + 1]) for s in range(a.steps)]))
                    for i, n in enumerate(names)}
        # the dominant kernel's launch duration: live, over the timed region itself
        stage_ms['cnn_front'] = float(np.mean([evs[s][1].elapsed_time(evs[s][2]) for s in range(a.steps)]))
        pmc, pmc_file = pmc_kernels()

        def roofline_of(prec, ms_front):
            if prec == 'bf16x3':
                flop, peak, kname, kern = (FLOP_CONV1_4 + FLOP_CONV5_6) * BATCH, PEAK_BF16_MFMA, 'cnn_front_bf16_kernel', \
                    'cnn_front_bf16_kernel (AdaptCNN conv1-6 + pools, split-bf16 MFMA: 3 products per term)'
            elif prec == 'bf16x6':
                flop, peak, kname, kern = (FLOP_CONV1_4 + FLOP_CONV5_6) * BATCH, PEAK_BF16_MFMA, 'cnn_front_bf16x6_kernel', \
                    'cnn_front_bf16x6_kernel (AdaptCNN conv1-6 + pools, three exact bf16 terms per fp32 operand: 6 MFMA products per term pair)'
            elif prec in ('f16x4', 'f16x3'):
                flop, peak, kname, kern = (FLOP_CONV1_4 + FLOP_CONV5_6) * BATCH, PEAK_BF16_MFMA, 'cnn_front_f16_kernel' if prec == 'f16x4' else 'n/a (the PMC passes profile the four-product instantiation)', \
                    'cnn_front_f16_kernel (AdaptCNN conv1-6 + pools, two f16 terms per fp32 operand of the scaled tensors: %s MFMA products per term pair; ' \
                    'the f16 matrix peak is the bf16 one)' % prec[-1]
            else:
                flop, peak, kname, kern = FLOP_CONV1_4 * BATCH, PEAK_F32, 'cnn_front_kernel', 'cnn_front_kernel (conv1-4 + pools, fp32 MFMA)'
            ach = flop / (ms_front * 1e-3) / 1e12
            tr, mu = pmc_derived(pmc.get(kname))
            r = {'kernel': kern, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                 'frac': round(ach / peak, 4), 'traffic': tr, 'traffic_unit': 'HBM bytes/launch (rocprofv3 PMC of this build: %s)' % pmc_file,
                 'mfma_util': mu, 'flop_per_launch': flop, 'avg_launch_ms': round(ms_front, 4)}
            if prec in ('bf16x6', 'f16x4', 'f16x3'):     # fp32-grade results: also against the fp32 matrix peak the exact-fp32 kernels are priced on
                r['achieved_vs_fp32_mfma_peak'] = round(ach / PEAK_F32, 4)
            return r

        clips = BATCH * a.steps * world
        roof = roofline_of(eng.precision, stage_ms['cnn_front'])
        roof['whole_path_tflops'] = round(FLOP_TOTAL * clips / dt / 1e12 / world, 2)
        k = pmc.get(CNN_KERNEL[eng.precision]) or {}
        if k.get('GRBM_GUI_ACTIVE') and k.get('_ms', 0) >= 0.05:           # (not a clock for launches under ~50 us: tools/pmc_to_json.py)
            roof['shader_clock_mhz_profiled'] = round(k['GRBM_GUI_ACTIVE'] / 8.0 / (k['_ms'] * 1e-3) / 1e6, 0)
        if world == 1 and not a.no_extras and eng.precision != 'f32':
            sus = sus or mfma_sustained(dev)
            roof['peak_sustained'] = {'what': 'dense v_mfma_f32_32x32x16_bf16 on register operands, measured on this GPU just now: '
                                              'zero operands reach the data-sheet peak, random operands are power-limited',
                                      **sus, 'frac_of_sustained_random': round(roof['achieved'] / sus['random']['tflops'], 4)}
        mel_flop = FLOP_MEL_FRAME * plan.total_frames
        mel_ach = mel_flop / (stage_ms['mel'] * 1e-3) / 1e12
        mtr, _ = pmc_derived(pmc.get('mel_frame_kernel'))
        kern_tab = {}
        tdp = eng.td_precision
        for stage, ks in (('mel', ['mel_frame_kernel']), ('cnn', ['cnn_front_kernel', 'cnn_back_kernel'] if eng.precision == 'f32' else [CNN_KERNEL[eng.precision]]),
                          ('selfatt', ['td_fused_bf16_kernel', 'td_proj_bf16_kernel', 'td_layer_bf16_kernel'] if tdp == 'bf16x3' else ['td16_proj_kernel', 'td16_layer_kernel<false>', 'td16_layer_kernel<true>'] if tdp == 'bf16x6' else ['td_proj_kernel', 'td_layer_kernel']),
                          ('pool', ['pool_score_bf16_kernel', 'pool_final_kernel'] if tdp == 'bf16x3' else [] if tdp == 'bf16x6' else ['pool_score_kernel', 'pool_final_kernel'])):
            for kn in ks:
                if kn in pmc:
                    tr, mu = pmc_derived(pmc[kn])
                    kern_tab[kn] = {'stage': stage, 'hbm_bytes_per_launch': tr, 'mfma_util': mu}
        res = {
            'metric': 'clips/sec (10 s, 48 kHz)', 'value': round(clips / dt, 2), 'unit': 'clips/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': DTYPE_TEXT[eng.precision],
            'data': 'synthetic 48 kHz / 10 s PCM16 clips (SURVEY 8d generator); ' + wdesc,
            'config': {'workload': 'predict_dir nisqa.tar (NISQA_DIM CNN-SA-AP) bs=64 per GPU, 10 s synthetic 48 kHz '
                                   'clips, int16 PCM resident in HBM' + ('' if BATCH == 64 else ' [EXPERIMENT: bs=%d]' % BATCH), 'batch_clips_per_gpu': BATCH, 'streams': len(streams),
                       'precision': eng.precision,
                       'segments_per_batch': int(plan.n_wins.sum()), 'frames_per_batch': plan.total_frames,
                       'parallelism': 'clip-sharded x%d, final all_gather of MOS rows' % world,
                       'collective_backend': backend, 'world_size_seen': world},
            'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()},
            'stage_ms_note': 'cnn_front: HIP events around the dominant kernel inside the timed region (its live launch duration); the other '
                             'stages: a second, untimed pass of the same steps with all six stage events (an event record costs ~5 us of '
                             'stream time; selfatt carries the attention pooling since round 6: nisqa_td_pool_bf16x6)',
            'clock_state': ('the sustained-MFMA probe of roofline.peak_sustained (~100 ms of dense MFMA) ran before the warm-up steps: '
                            'the timed steps start at the loaded clock state (a short region behind an idle GPU measures 4 % low; '
                            'NISQA_BENCH_PROBE_LAST=1 restores the old order)') if probe_first else 'no load before the warm-up steps',
            'roofline': roof,
            'roofline_secondary': {'kernel': 'mel_frame_kernel (pruned 4 x FFT-512 + sparse mel bank + dB)', 'bound': 'valu',
                                   'achieved': round(mel_ach, 2), 'peak': PEAK_F32, 'unit': 'TFLOP/s',
                                   'frac': round(mel_ach / PEAK_F32, 4), 'flop_per_launch': mel_flop, 'traffic': mtr,
                                   'avg_launch_ms': round(stage_ms['mel'], 4)},
            'kernels': kern_tab,
        }
        if world == 1 and not a.no_extras:
            # the other precision path on the same workload: same --steps / --warmup, same barriers, right after the
            # primary region.  For the default run this is the exact-fp32 path -- the reference's own arithmetic.
            # (and 'bf16x6': every GEMM at fp32 operand precision on the bf16 matrix pipe -- three exact bf16 terms per operand, six
            # products: the accuracy of 'f32', tests/test_gpu_parity.py)
            alt_out = {}
            for other in [q for q in ALL_PRECISIONS if q != eng.precision]:
                eng2 = HipNisqa(margs, sd, dev, precision=other)
                for _ in range(max(2, a.warmup)):
                    o2 = eng2.forward_pcm(pcm, plan, SR)
                torch.cuda.synchronize()
                ev2, ev2_all = _stage_events(a.steps, only_cnn=other != 'f32'), _stage_events(a.steps)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for s_ in range(a.steps):
                    o2 = eng2.forward_pcm(pcm, plan, SR, stage_events=ev2[s_])
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
                for s_ in range(a.steps):                      # (untimed second pass: every stage)
                    eng2.forward_pcm(pcm, plan, SR, stage_events=ev2_all[s_])
                torch.cuda.synchronize()
                st2 = _stage_means(ev2_all, names)
                st2.update(_stage_means(ev2, names))           # the CNN kernels' durations from the timed pass
                r2 = roofline_of(other, st2['cnn_front'])
                res['value_' + other] = round(BATCH * a.steps / dt2, 2)
                res[other] = {'precision': other, 'value': round(BATCH * a.steps / dt2, 2), 'unit': 'clips/s', 'steps': a.steps,
                              'warmup': max(2, a.warmup), 'ms_per_step': round(1e3 * dt2 / a.steps, 4),
                              'stage_ms': {k: round(v, 4) for k, v in st2.items()}, 'roofline': r2,
                              'max_abs_diff_vs_primary': float((o2 - outs[-1]).abs().max())}
                alt_out[other] = o2
                if other == 'f32':           # how far every path is from the exact-fp32 kernels on this batch
                    res['max_abs_diff_vs_f32'] = res[other]['max_abs_diff_vs_primary']
                elif 'f32' in alt_out:
                    res[other]['max_abs_diff_vs_f32'] = float((o2 - alt_out['f32']).abs().max())
                if other == 'f32':
                    trb, mub = pmc_derived(pmc.get('cnn_back_kernel'))
                    achb = FLOP_CONV5_6 * BATCH / (st2['cnn_back'] * 1e-3) / 1e12
                    res[other]['roofline_cnn_back'] = {'kernel': 'cnn_back_kernel (conv5-6, fp32 MFMA)', 'bound': 'mfma', 'achieved': round(achb, 2),
                                                       'peak': PEAK_F32, 'unit': 'TFLOP/s', 'frac': round(achb / PEAK_F32, 4),
                                                       'traffic': trb, 'mfma_util': mub, 'avg_launch_ms': round(st2['cnn_back'], 4)}
            # the primary path once more over a run long enough for the clocks to settle (a 20-step region is 16 ms)
            n3 = max(400, a.steps)
            for _ in range(50):
                eng.forward_pcm(pcm, plan, SR)
            torch.cuda.synchronize()
            ev3 = _stage_events(n3, only_cnn=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s_ in range(n3):
                eng.forward_pcm(pcm, plan, SR, stage_events=ev3[s_])
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t1
            st3 = _stage_means(ev3, names)
            res['steady'] = {'value': round(BATCH * n3 / dt3, 2), 'unit': 'clips/s', 'steps': n3, 'warmup': 50,
                             'ms_per_step': round(1e3 * dt3 / n3, 4), 'stage_ms': {k: round(v, 4) for k, v in st3.items()},
                             'roofline_frac': roofline_of(eng.precision, st3['cnn_front'])['frac']}
        if world == 1 and len(streams) == 1 and not a.no_extras:
            # the same steps alternated over 2 streams (what the predict loop does) and over 3: launches overlap, one batch's
            # VALU-bound mel kernel runs next to another's MFMA-bound CNN kernel.  No events inside; >= 200 steps behind 4 per stream
            # (a 20-step region on freshly created streams measures their start-up)
            n_ov = max(200, a.steps)
            for ns in (2, 3):
                sts = [torch.cuda.Stream(device=dev) for _ in range(ns)]
                for i in range(4 * ns):
                    with torch.cuda.stream(sts[i % ns]):
                        eng.forward_pcm(pcm, plan, SR)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for s in range(n_ov):
                    with torch.cuda.stream(sts[s % ns]):
                        eng.forward_pcm(pcm, plan, SR)
                torch.cuda.synchronize()
                res['overlap_%d_streams' % ns] = {'value': round(BATCH * n_ov / (time.perf_counter() - t1), 2), 'unit': 'clips/s', 'steps': n_ov}
        if world == 1 and not a.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(margs, sd)
        if world == 1 and a.leg == 'all' and not a.no_side and not a.no_extras:
            # BASELINE.json configs[2], [3], [4]: bounded side legs (a failure in one must not cost the contract line)
            res['side'] = {}
            for name, fn in (('tts_mixed', lambda: side_tts(dev, 3, not a.no_cpu_baseline, pmc)),
                             ('train_step', lambda: side_train(dev, 30, not a.no_cpu_baseline, pmc)),
                             ('predict_csv_1gpu', lambda: side_predict_csv(dev, res.get('cpu_baseline'))),
                             ('predict_dir_bs64', lambda: side_predict_dir(dev, res.get('cpu_baseline'),
                                                                           res['side']['predict_csv_1gpu']['roofline']['link_only_GBps']))):
                t_leg = time.perf_counter()
                try:
                    res['side'][name] = fn()
                except Exception as e:                       # noqa: BLE001
                    res['side'][name] = {'error': '%s: %s' % (type(e).__name__, e)}
                res['side'][name]['leg_seconds'] = round(time.perf_counter() - t_leg, 1)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

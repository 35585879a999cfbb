#!/usr/bin/env python
"""bench.py -- clips/sec of the NISQA predict hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the whole hot path (PCM -> mel -> AdaptCNN -> self-attention -> pooling
heads -> [B,5] rows) over one batch of synthetic input (int16 PCM, as in the WAV files) already resident in HBM.  Workload at every N:
BASELINE.json configs[1] per GPU -- nisqa.tar architecture (NISQA_DIM, random-init weights: there
are no checkpoints on the GPU box), bs = 64 clips of 10 s / 48 kHz synthetic audio (SURVEY.md
section 8d generator).  Weak scaling: each rank owns its own 64-clip batch (clips shard with no
data-path collective); the only exchange is the final all_gather of result rows, inside the timed
region.  value = clips all ranks processed / max-over-ranks wall time.

Extra objects on the JSON line (see DESIGN.md "Measurement"):
  roofline     dominant kernel: algorithmic FLOPs per launch / its mean launch duration measured with HIP
               events recorded inside the timed region on the launch stream.  precision bf16x3 (default):
               cnn_front_bf16_kernel (whole AdaptCNN, 160.6 GFLOP/launch) against the dense bf16 MFMA peak
               (2500 TFLOP/s; the kernel issues 3 bf16 products per algorithmic product, so 1/3 is its
               ceiling); precision f32: cnn_front_kernel (conv1-4, 132.6 GFLOP) against 157.3 TFLOP/s.
  alt_precision the same workload on the other precision path, measured after the timed region.
  cpu_baseline the CPU oracle (a port of the reference path: numpy mel restatement + torch-CPU
               network) timed on this box's host cores over a bounded sample, rank 0, N = 1 only.
"""
import argparse
import json
import os
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
FLOP_TOTAL = 2.93e9                                          # whole path incl. mel in FFT form
PEAK_F32_MFMA = 157.3                                        # TFLOP/s, MI355X_MICROARCH.md
PEAK_BF16_MFMA = 2500.0                                      # TFLOP/s dense, MI355X_MICROARCH.md
# HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, KB units, the
# gfx950 x2 correction for wide reads): profiles/r01_pmc_bench_bf16x3.txt, profiles/r01_pmc_f32_cnn.txt
PMC_TRAFFIC_BYTES = {'bf16x3': 2 * 12665.2e3 * 1.024 + 23712.0e3 * 1.024, 'f32': 2 * 23916.2e3 * 1.024 + 71136.0e3 * 1.024}


def cpu_baseline(n_distinct=6, min_seconds=12.0):
    """Oracle (CPU port of the reference path) on a bounded sample (~12 s of CPU work); returns the JSON object."""
    from oracle import mel as omel, net as onet
    args, sd = dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM')
    clips = [synth.synth_pcm16(2000 + i, SECONDS).astype(np.float32) / np.float32(32768.0) for i in range(n_distinct)]
    onet.predict_from_melspec(sd, args, omel.melspec_db_from_audio(clips[0][:SR], SR))   # warm-up
    t0 = time.perf_counter()
    t_mel, n = 0.0, 0
    while n < n_distinct or time.perf_counter() - t0 < min_seconds:
        y = clips[n % n_distinct]
        t1 = time.perf_counter()
        spec = omel.melspec_db_from_audio(y, SR)
        t_mel += time.perf_counter() - t1
        onet.predict_from_melspec(sd, args, spec)
        n += 1
    dt = time.perf_counter() - t0
    return {'value': round(n / dt, 3), 'unit': 'clips/s', 'cores': int(torch.get_num_threads()),
            'kind': 'port',
            'sample': '%d x 10 s clips (%d distinct), oracle.mel (numpy restatement of librosa 0.8.1) + oracle.net '
                      '(torch CPU fp32), one clip at a time; %.1f s total, mel share %.0f%%; host has %d cores'
                      % (n, n_distinct, dt, 100.0 * t_mel / dt, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # a step is ~1 ms: 50 + 400 steps are under half a second of GPU time and let the clocks settle (20 steps after 3
    # warm-up steps measure ~7 % low)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default=None, choices=['bf16x3', 'f32'])
    ap.add_argument('--no-extras', action='store_true', help='skip the alt-precision and 3-stream passes (profiling runs)')
    ap.add_argument('--streams', type=int, default=1, help='HIP streams the steps alternate over (kernel tails of one '
                    'batch overlap the next batch)')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
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
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if shared:
            torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
        else:
            torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from nisqa_amd.engine import HipNisqa
    eng = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), dev, precision=a.precision)

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
        assert torch.isfinite(outs[-1]).all()
        names = ['mel', 'cnn_front', 'cnn_back', 'selfatt', 'pool']
        stage_ms = {n: float(np.mean([evs[s][i].elapsed_time(evs[s][i + 1]) for s in range(a.steps)]))
                    for i, n in enumerate(names)}
        def roofline_of(prec, ms_front):
            if prec == 'bf16x3':
                flop, peak, kern = (FLOP_CONV1_4 + FLOP_CONV5_6) * BATCH, PEAK_BF16_MFMA, \
                    'cnn_front_bf16_kernel (AdaptCNN conv1-6 + pools, split-bf16 MFMA: 3 products per term)'
            else:
                flop, peak, kern = FLOP_CONV1_4 * BATCH, PEAK_F32_MFMA, 'cnn_front_kernel (conv1-4 + pools, fp32 MFMA)'
            ach = flop / (ms_front * 1e-3) / 1e12
            return {'kernel': kern, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': round(ach / peak, 4), 'traffic': round(PMC_TRAFFIC_BYTES[prec]), 'traffic_unit': 'bytes/launch (rocprofv3 PMC, profiles/)',
                    'flop_per_launch': flop,
                    'avg_launch_ms': round(ms_front, 4)}

        clips = BATCH * a.steps * world
        roof = roofline_of(eng.precision, stage_ms['cnn_front'])
        roof['whole_path_tflops'] = round(FLOP_TOTAL * clips / dt / 1e12 / world, 2)
        res = {
            'metric': 'clips/sec (10 s, 48 kHz)', 'value': round(clips / dt, 2), 'unit': 'clips/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16x3 (bf16 hi+lo operands, 3 MFMA products per term, f32 accumulate; mel/attention/pooling f32)'
                     if eng.precision == 'bf16x3' else 'f32',
            'data': 'synthetic 48 kHz / 10 s PCM16 clips (SURVEY 8d generator), random-init nisqa.tar architecture',
            'config': {'workload': 'predict_dir nisqa.tar (NISQA_DIM CNN-SA-AP) bs=64 per GPU, 10 s synthetic 48 kHz '
                                   'clips, int16 PCM resident in HBM', 'batch_clips_per_gpu': BATCH, 'streams': len(streams),
                       'precision': eng.precision,
                       'segments_per_batch': int(plan.n_wins.sum()), 'frames_per_batch': plan.total_frames,
                       'parallelism': 'clip-sharded x%d, final all_gather of MOS rows' % world},
            'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()},
            'roofline': roof,
        }
        if world == 1 and not a.no_extras:
            # the other precision path on the same workload (secondary measurement, outside the timed region)
            other = 'f32' if eng.precision == 'bf16x3' else 'bf16x3'
            eng2 = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), dev, precision=other)
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
            # same steps alternated over 3 streams: kernel tails / the latency-bound attention kernels of one batch
            # overlap the next batch (what the predict loop does); per-kernel times are not comparable in this mode
            st3 = [torch.cuda.Stream(device=dev) for _ in range(3)]
            for i in range(3):
                with torch.cuda.stream(st3[i]):
                    eng.forward_pcm(pcm, plan, SR)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s in range(a.steps):
                with torch.cuda.stream(st3[s % 3]):
                    o3 = eng.forward_pcm(pcm, plan, SR)
            torch.cuda.synchronize()
            res['overlap_3_streams'] = {'value': round(BATCH * a.steps / (time.perf_counter() - t1), 2), 'unit': 'clips/s'}
        if world == 1 and not a.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

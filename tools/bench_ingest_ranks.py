"""Host-feed ceiling of BASELINE configs[2] (predict_csv, 8 ranks on one node) WITHOUT GPUs: N processes, each running the
predict loop's own host side -- nisqa_amd.ingest.Ingest with the loop's batching policy, native reader pool, three-slot
staging ring -- over its contiguous shard of one CSV of 10 s / 48 kHz PCM16 WAV files (page cache), the consumer only
recycling the slots (what the H2D copy's event would do).  Reports staged clips/s and GB/s per rank and in aggregate for
N = 1, 2, 4, 8: where the curve bends is where the host (memory bandwidth, CPU quota, page cache) stops feeding the GPUs.

    python tools/bench_ingest_ranks.py [--clips 4096] [--distinct 64] [--bs 256] [--ranks 1,2,4,8] [--pin 0|1]

DESIGN.md 7.2 holds the DRAM-traffic model these numbers are compared with."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main(rank, world, d, clips, bs, pin, workers, start, out):
    os.environ['LOCAL_WORLD_SIZE'] = str(world)              # what torchrun sets: the ranks share the node's CPU budget
    import numpy as np                                        # noqa: F401
    import pandas as pd
    import torch                                              # noqa: F401  (page-locked buffers when --pin 1)
    from nisqa_amd import ingest, synth
    from nisqa_amd import NISQA_lib as NL
    df = pd.read_csv(os.path.join(d, 'list.csv'))
    a = synth.DIM_ARGS
    ds = NL.SpeechQualityDataset(df, data_dir=d, filename_column='deg', mos_column='predict_only',
                                 seg_length=a['ms_seg_length'], max_length=a['ms_max_segments'], seg_hop_length=a['ms_seg_hop_length'],
                                 ms_n_fft=a['ms_n_fft'], ms_hop_length=a['ms_hop_length'], ms_win_length=a['ms_win_length'],
                                 ms_n_mels=a['ms_n_mels'], ms_sr=a['ms_sr'], ms_fmax=a['ms_fmax'], dim=True)
    base, rem = divmod(clips, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)

    class Eng(object):
        arch = 0
    pol = NL.batch_policy(Eng, ds, range(lo, hi), bs)
    # one untimed pass (page cache, ring allocation), then all ranks start together
    for rep in range(2):
        if rep == 1:
            while time.time() < start:
                time.sleep(0.001)
        t0 = time.perf_counter()
        ing = ingest.Ingest(ds, pol, pin=bool(pin), num_workers=workers)
        nbytes = nclips = 0
        for st in ing:
            for g in st.groups:
                nbytes += g.nbytes
                nclips += len(g.ids)
            ing.ring.release_after(st.slot, None)
        ing.close()
        dt = time.perf_counter() - t0
    out.put({'rank': rank, 'clips': nclips, 'bytes': nbytes, 'seconds': dt, 'readers': ing.workers, 'stats': {k: round(v, 4) for k, v in ing.stats.items()}})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=4096)
    ap.add_argument('--distinct', type=int, default=64)
    ap.add_argument('--bs', type=int, default=256)
    ap.add_argument('--ranks', default='1,2,4,8')
    ap.add_argument('--pin', type=int, default=0)
    ap.add_argument('--workers', type=int, default=0, help='reader threads per rank (0 = the loop default: CPU budget - 3)')
    a = ap.parse_args()
    import pandas as pd
    from nisqa_amd import synth, ingest
    res = {'what': 'ingest only, no GPU: WAV files (page cache) -> staging ring, per rank and aggregate', 'clips': a.clips,
           'bs_hint': a.bs, 'pin': a.pin, 'cpu_count': os.cpu_count(), 'cpu_budget': ingest.cpu_budget(), 'runs': []}
    with tempfile.TemporaryDirectory() as d:
        for i in range(a.distinct):
            synth.write_wav(os.path.join(d, 'c%05d.wav' % i), synth.synth_pcm16(3000 + i % 8, 10.0), 48000)
        pd.DataFrame({'deg': ['c%05d.wav' % (i % a.distinct) for i in range(a.clips)]}).to_csv(os.path.join(d, 'list.csv'), index=False)
        ctx = mp.get_context('spawn')
        for world in [int(x) for x in a.ranks.split(',')]:
            out = ctx.Queue()
            start = time.time() + 6.0 + 1.0 * world            # spawn + imports + warm pass
            ps = [ctx.Process(target=rank_main, args=(r, world, d, a.clips, a.bs, a.pin, a.workers, start, out)) for r in range(world)]
            for p in ps:
                p.start()
            rows = [out.get(timeout=600) for _ in ps]
            for p in ps:
                p.join()
            tmax = max(r['seconds'] for r in rows)
            tot_b, tot_c = sum(r['bytes'] for r in rows), sum(r['clips'] for r in rows)
            res['runs'].append({'ranks': world, 'aggregate_clips_per_s': round(tot_c / tmax, 1), 'aggregate_GBps': round(tot_b / tmax / 1e9, 2),
                                'per_rank_clips_per_s': [round(r['clips'] / r['seconds'], 1) for r in sorted(rows, key=lambda r: r['rank'])],
                                'readers_per_rank': rows[0]['readers'], 'slowest_rank_s': round(tmax, 3),
                                'rank0_stats': sorted(rows, key=lambda r: r['rank'])[0]['stats']})
            print(json.dumps(res['runs'][-1]), file=sys.stderr)
    print(json.dumps(res))


if __name__ == '__main__':
    main()

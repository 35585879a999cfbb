"""Where one batch of the predict loop spends its host time: Ingest's own counters (producer thread) next to the
consumer's (queue wait, plan, enqueue, result wait).   python tools/probe_loop.py [n_files] [bs] [workers]"""
import io, json, os, sys, tempfile, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth, ingest
from nisqa_amd import NISQA_lib as NL
from nisqa_amd.NISQA_model import nisqaModel

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 32
with tempfile.TemporaryDirectory(dir=os.environ.get('NISQA_TMP', None)) as tmp:
    args = dict(synth.DIM_ARGS)
    args.update({'pretrained_model': False, 'tr_bs_val': bs, 'tr_num_workers': 0})
    ck = os.path.join(tmp, 'rand.tar')
    torch.save({'args': args, 'model_state_dict': synth.random_state_dict(7, 'NISQA_DIM')}, ck)
    d = os.path.join(tmp, 'wavs')
    os.mkdir(d)
    clips = [synth.synth_pcm16(i, 10.0) for i in range(8)]
    n_distinct = min(n_files, 512)
    for i in range(n_distinct):
        synth.write_wav(os.path.join(d, 'c%05d.wav' % i), clips[i % 8], 48000)
    import pandas as pd
    pd.DataFrame({'deg': ['c%05d.wav' % (i % n_distinct) for i in range(n_files)]}).to_csv(os.path.join(d, 'files.csv'), index=False)
    a = {'mode': 'predict_csv', 'pretrained_model': ck, 'deg': None, 'data_dir': d, 'output_dir': None,
         'csv_file': 'files.csv', 'csv_deg': 'deg', 'num_workers': workers, 'bs': bs, 'ms_channel': None, 'tr_bs_val': bs,
         'tr_num_workers': workers}
    m = nisqaModel(a)
    with contextlib.redirect_stdout(io.StringIO()):
        m.predict()
    ds = m.ds_val
    keep = {}
    orig = ingest.Ingest

    class Spy(orig):
        def __init__(self, *a_, **k_):
            orig.__init__(self, *a_, **k_)
            keep['ing'] = self
    ingest.Ingest = Spy
    NL._ingest.Ingest = Spy
    dummies = []
    for rep in range(int(os.environ.get('REPS', 3))):
        if os.environ.get('SHIFT'):                          # a new stream generation every second repetition
            if rep % 2 == 0:
                NL._STREAMS.clear()
                dummies += [torch.cuda.Stream(m.dev) for _ in range(int(os.environ['SHIFT']))]
        t0 = time.perf_counter()
        NL._predict(m.model, ds, bs, m.dev, workers)
        dt = time.perf_counter() - t0
        st = keep['ing'].stats
        nb = max(1, st['batches'])
        print('rep %d: %.0f clips/s, %.2f ms per batch of %d; producer ms/batch: %s; consumer ms/batch: %s' % (
            rep, n_files / dt, dt / nb * 1e3, bs,
            {k: round(v / nb * 1e3, 2) for k, v in st.items() if k != 'batches'},
            {k: round(v / nb * 1e3, 2) for k, v in NL.LOOP_STATS.items()}), flush=True)

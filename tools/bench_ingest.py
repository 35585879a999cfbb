"""Where the end-to-end predict_dir time goes (host side): staging alone, staging + H2D + kernels, full predict()."""
import io, json, os, sys, tempfile, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth, ingest
from nisqa_amd import NISQA_lib as NL
from nisqa_amd.NISQA_model import nisqaModel

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
workers = [int(w) for w in sys.argv[2].split(',')] if len(sys.argv) > 2 else [8, 32, 128]
res = {'n_files': n_files, 'cpus': os.cpu_count()}
with tempfile.TemporaryDirectory() as tmp:
    args = dict(synth.DIM_ARGS)
    args.update({'pretrained_model': False, 'tr_bs_val': 64, 'tr_num_workers': 0})
    ck = os.path.join(tmp, 'rand.tar')
    torch.save({'args': args, 'model_state_dict': synth.random_state_dict(7, 'NISQA_DIM')}, ck)
    d = os.path.join(tmp, 'wavs')
    os.mkdir(d)
    clips = [synth.synth_pcm16(i, 10.0) for i in range(8)]
    for i in range(n_files):
        synth.write_wav(os.path.join(d, 'c%05d.wav' % i), clips[i % 8], 48000)
    a = {'mode': 'predict_dir', 'pretrained_model': ck, 'deg': None, 'data_dir': d, 'output_dir': None,
         'csv_file': None, 'csv_deg': None, 'num_workers': 8, 'bs': 64, 'ms_channel': None, 'tr_bs_val': 64,
         'tr_num_workers': 8}
    m = nisqaModel(a)
    with contextlib.redirect_stdout(io.StringIO()):
        m.predict()
    ds = m.ds_val
    batches = [list(range(s, min(s + 64, n_files))) for s in range(0, n_files, 64)]
    for w in workers:
        r = {}
        for pin in (False, True):
            best = 0
            for rep in range(2):
                t0 = time.perf_counter()
                ing = ingest.Ingest(ds, batches, pin=pin, num_workers=w)
                for st in ing:
                    ing.ring.release_after(st.slot, None)
                ing.close()
                best = max(best, n_files / (time.perf_counter() - t0))
            r['stage_only_pin%d' % pin] = round(best)
        best = 0
        for rep in range(2):
            t0 = time.perf_counter()
            NL._predict(m.model, ds, 64, m.dev, w)
            best = max(best, n_files / (time.perf_counter() - t0))
        r['predict_loop'] = round(best)
        m.args['tr_num_workers'] = w
        best = 0
        for rep in range(2):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                m.predict()
            best = max(best, n_files / (time.perf_counter() - t0))
        r['predict_full_quiet'] = round(best)
        res['workers_%d' % w] = r
print(json.dumps(res))

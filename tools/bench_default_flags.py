"""run_predict.py at the REFERENCE'S DEFAULT FLAGS (--bs 1 --num_workers 0, run_predict.py:16-20) against tuned flags and
against the reference's own batching (NISQA_EXACT_BS=1): nisqaModel.predict() in predict_dir mode over N synthetic 10 s /
48 kHz PCM16 WAV files in the page cache, and the same for the nisqa_tts.tar architecture on mixed 3-30 s clips.  Round 3:
the loop coalesces the default flags into work-sized, length-sorted batches (NISQA_lib.batch_policy)."""
import contextlib, io, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth
from nisqa_amd.NISQA_model import nisqaModel

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
res = {'n_files': n_files}


def run(a, reps=3, env=None):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            m = nisqaModel(a)
            best, df = 0.0, None
            for _ in range(reps):
                t0 = time.perf_counter()
                df = m.predict()
                best = max(best, len(df) / (time.perf_counter() - t0))
        assert np.isfinite(df['mos_pred'].to_numpy(dtype=float)).all()
        return round(best, 1), df['mos_pred'].to_numpy(dtype=float)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


with tempfile.TemporaryDirectory() as tmp:
    for arch, aargs, seed, durs in (('nisqa', synth.DIM_ARGS, 7, None),
                                    ('nisqa_tts', synth.TTS_ARGS, 9, np.random.default_rng(7).uniform(3, 30, n_files // 4))):
        args = dict(aargs)
        args.update({'pretrained_model': False, 'tr_bs_val': 1, 'tr_num_workers': 0})
        ck = os.path.join(tmp, arch + '.tar')
        torch.save({'args': args, 'model_state_dict': synth.random_state_dict(seed, 'NISQA_DIM' if arch == 'nisqa' else 'NISQA_TTS')}, ck)
        d = os.path.join(tmp, arch)
        os.mkdir(d)
        if durs is None:
            clips = [synth.synth_pcm16(i, 10.0) for i in range(8)]
            for i in range(n_files):
                synth.write_wav(os.path.join(d, 'c%05d.wav' % i), clips[i % 8], 48000)
        else:
            base = synth.synth_pcm16(5, 30.0)
            for i, du in enumerate(durs):
                synth.write_wav(os.path.join(d, 'c%05d.wav' % i), base[:int(du * 48000)], 48000)

        def a(bs, w):
            return {'mode': 'predict_dir', 'pretrained_model': ck, 'deg': None, 'data_dir': d, 'output_dir': None, 'csv_file': None,
                    'csv_deg': None, 'num_workers': w, 'bs': bs, 'ms_channel': None, 'tr_bs_val': bs, 'tr_num_workers': w}
        r = {}
        r['default_flags_bs1_workers0'], y0 = run(a(1, 0))
        r['tuned_bs64_workers12'], y1 = run(a(64, 12))
        r['reference_batches_bs1_workers0 (NISQA_EXACT_BS=1)'], y2 = run(a(1, 0), reps=1, env={'NISQA_EXACT_BS': '1'})
        r['reference_batches_bs64_workers12 (NISQA_EXACT_BS=1)'], y3 = run(a(64, 12), env={'NISQA_EXACT_BS': '1'})
        r['max_abs_diff_between_batchings'] = float(max(np.abs(y0 - y1).max(), np.abs(y0 - y2).max(), np.abs(y0 - y3).max()))
        res[arch] = r
print(json.dumps(res))

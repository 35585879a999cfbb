"""End-to-end rate of the drop-in surface: nisqaModel.predict() in
predict_dir mode over N synthetic 10 s / 48 kHz PCM16 WAV files that sit in the page cache -- file read + RIFF
parse + H2D + kernels + DataFrame.  Not the driver's bench contract (bench.py times the HBM-resident hot path)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth
from nisqa_amd.NISQA_model import nisqaModel

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 512
workers = [int(w) for w in sys.argv[2].split(',')] if len(sys.argv) > 2 else [8, 32]
res = {'n_files': n_files}
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
    for w in workers:
        a = {'mode': 'predict_dir', 'pretrained_model': ck, 'deg': None, 'data_dir': d, 'output_dir': None,
             'csv_file': None, 'csv_deg': None, 'num_workers': w, 'bs': 64, 'ms_channel': None, 'tr_bs_val': 64,
             'tr_num_workers': w}
        m = nisqaModel(a)
        best = 0.0
        for rep in range(3):
            t0 = time.perf_counter()
            df = m.predict()
            dt = time.perf_counter() - t0
            best = max(best, n_files / dt)
        assert len(df) == n_files and np.isfinite(df['mos_pred'].to_numpy(dtype=float)).all()
        res['workers_%d' % w] = {'clips_per_s': round(best, 1), 'GBps_of_wav': round(best * 960044 / 1e9, 2)}
print(json.dumps(res))

"""Oracle support (test infrastructure): import the reference's torch modules in
a container that has no librosa.

DEVIATION, stated wherever it is used: ``nisqa/NISQA_lib.py:10`` does
``import librosa as lb`` at module scope and librosa is not installed here, so
an EMPTY stand-in module is registered as ``sys.modules['librosa']`` before
the import.  Only torch code of the reference then runs (model classes NL:29-1417,
``segment_specs`` NL:2239-2282); ``lb.load`` / ``lb.feature.melspectrogram`` /
``lb.core.amplitude_to_db`` are never reached.  /root/reference exists only in
the build container; ``__graft_entry__.build()`` stages ``nisqa/NISQA_lib.py`` under
``oracle/_ref/nisqa/`` (git-ignored, shipped with the snapshot like the checkpoints under
``oracle/_ref/weights/``), and REFERENCE_ROOT falls back to ``oracle/_ref`` where /root/reference
does not exist -- so the reference's torch half is importable on the GPU box too (live-reference
``-m gpu`` test, ``bench.py``'s ``cpu_baseline``).  Nothing of the product imports this module.

``import_reference_lib(functional_librosa=True)`` registers a stand-in whose three entry points the
reference calls -- ``lb.load``, ``lb.feature.melspectrogram``, ``lb.core.amplitude_to_db``
(NISQA_lib.py:2299-2330) -- are served by ``oracle/mel.py`` (the restatement of librosa 0.8.1, PARITY
UNPINNED): the reference's own ``get_librosa_melspec`` / ``SpeechQualityDataset`` / ``predict_dim`` then run
unmodified end to end, with only the third-party arithmetic replaced.
"""
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
_STAGED = os.path.join(_HERE, '_ref')
REFERENCE_ROOT = os.environ.get('NISQA_REFERENCE_ROOT') or (
    '/root/reference' if os.path.isfile('/root/reference/nisqa/NISQA_lib.py') else _STAGED)


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'nisqa', 'NISQA_lib.py'))


def _functional_librosa():
    """A ``librosa`` stand-in that serves exactly the three calls of NISQA_lib.py:2299-2330 from oracle/mel.py."""
    from . import mel as omel
    import numpy as np
    lb = types.ModuleType('librosa')
    lb.feature, lb.core = types.ModuleType('librosa.feature'), types.ModuleType('librosa.core')

    def load(path, sr=None, mono=True):
        if sr is not None:
            raise NotImplementedError('oracle covers ms_sr=None (all shipped checkpoints)')
        if mono:
            return omel.load_wav(path)
        from scipy.io import wavfile                       # mono=False: (channels, n), the caller picks a row
        rate, data = wavfile.read(path)
        if data.ndim == 1:
            return omel.load_wav(path)
        chans = [omel.load_wav(path, ms_channel=c)[0] for c in range(data.shape[1])]
        return np.stack(chans), int(rate)

    def melspectrogram(y=None, sr=None, S=None, n_fft=None, hop_length=None, win_length=None, window='hann', center=True,
                       pad_mode='reflect', power=1.0, n_mels=None, fmin=0.0, fmax=None, htk=False, norm='slaney'):
        assert S is None and window == 'hann' and center and pad_mode == 'reflect' and power == 1.0 and not htk \
            and norm == 'slaney' and fmin == 0.0
        mag = omel.stft_mag(y, n_fft, hop_length, win_length)
        return np.dot(omel.mel_filterbank(sr, n_fft, n_mels, fmin, fmax), mag).astype(np.float32)

    def amplitude_to_db(S, ref=1.0, amin=1e-4, top_db=80.0):
        assert ref == 1.0
        return omel.amplitude_to_db(S, amin=amin, top_db=top_db)

    lb.load, lb.feature.melspectrogram, lb.core.amplitude_to_db = load, melspectrogram, amplitude_to_db
    return lb


def import_reference_lib(functional_librosa=False):
    """Return the reference's ``nisqa.NISQA_lib`` module (torch parts usable).  functional_librosa: the module's ``lb``
    is the oracle-backed stand-in above instead of an empty one, so its mel front end runs too (PARITY UNPINNED there)."""
    if not reference_available():
        raise ImportError('reference tree not present at ' + REFERENCE_ROOT)
    sys.dont_write_bytecode = True            # keep /root/reference clean
    standin = 'librosa' not in sys.modules
    if standin:
        sys.modules['librosa'] = types.ModuleType('librosa')   # the stand-in
    import matplotlib
    matplotlib.use('Agg')
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    try:
        mod = importlib.import_module('nisqa.NISQA_lib')
        if functional_librosa and not hasattr(getattr(mod, 'lb', None), 'load'):
            mod.lb = _functional_librosa()
        return mod
    finally:
        if standin:                            # the reference keeps its own `lb`; nobody else should see the stand-in
            del sys.modules['librosa']


MODEL_ARG_KEYS = [
    'ms_seg_length', 'ms_n_mels', 'cnn_model', 'cnn_c_out_1', 'cnn_c_out_2', 'cnn_c_out_3',
    'cnn_kernel_size', 'cnn_dropout', 'cnn_pool_1', 'cnn_pool_2', 'cnn_pool_3', 'cnn_fc_out_h',
    'td', 'td_sa_d_model', 'td_sa_nhead', 'td_sa_pos_enc', 'td_sa_num_layers', 'td_sa_h',
    'td_sa_dropout', 'td_lstm_h', 'td_lstm_num_layers', 'td_lstm_dropout', 'td_lstm_bidirectional',
    'td_2', 'td_2_sa_d_model', 'td_2_sa_nhead', 'td_2_sa_pos_enc', 'td_2_sa_num_layers', 'td_2_sa_h',
    'td_2_sa_dropout', 'td_2_lstm_h', 'td_2_lstm_num_layers', 'td_2_lstm_dropout',
    'td_2_lstm_bidirectional', 'pool', 'pool_att_h', 'pool_att_dropout']


def build_reference_model(args, state_dict):
    """The reference's own module tree with ``state_dict`` loaded strict (NISQA_model.py:1012-1023)."""
    NL = import_reference_lib()
    margs = {k: args[k] for k in MODEL_ARG_KEYS}
    model = {'NISQA': NL.NISQA, 'NISQA_DIM': NL.NISQA_DIM}[args['model']](**margs)
    model.load_state_dict(state_dict, strict=True)
    model.eval()
    return model, NL


def reference_predict(checkpoint, data_dir, files, bs, num_workers=0, ms_channel=None, timings=None):
    """The reference's predict path on CPU, its own code end to end (NISQA_model.py:745-776 dataset construction, 928-1030
    model construction, NISQA_lib.py:1420-1467 predict_mos / predict_dim: DataLoader -> SpeechQualityDataset.__getitem__
    -> get_librosa_melspec -> segment_specs padded to [B, ms_max_segments, 1, 48, 15] -> model(x, n_wins)), with librosa's
    three entry points served by oracle/mel.py (PARITY UNPINNED there).  files: basenames under data_dir.
    -> float32 [N, heads].  timings (dict, optional) receives the wall seconds of the predict call."""
    import time
    import pandas as pd
    import torch
    NL = import_reference_lib(functional_librosa=True)
    ck = torch.load(checkpoint, map_location='cpu')
    args = dict(ck['args'])
    args.setdefault('double_ended', args['model'] == 'NISQA_DE')
    args.setdefault('dim', args['model'] == 'NISQA_DIM')
    model = {'NISQA': NL.NISQA, 'NISQA_DIM': NL.NISQA_DIM}[args['model']](**{k: args[k] for k in MODEL_ARG_KEYS})
    model.load_state_dict(ck['model_state_dict'], strict=True)
    ds = NL.SpeechQualityDataset(
        pd.DataFrame(list(files), columns=['deg']), df_con=None, data_dir=data_dir, filename_column='deg',
        mos_column='predict_only', seg_length=args['ms_seg_length'], max_length=args['ms_max_segments'], to_memory=None,
        to_memory_workers=None, seg_hop_length=args['ms_seg_hop_length'], transform=None, ms_n_fft=args['ms_n_fft'],
        ms_hop_length=args['ms_hop_length'], ms_win_length=args['ms_win_length'], ms_n_mels=args['ms_n_mels'],
        ms_sr=args['ms_sr'], ms_fmax=args['ms_fmax'], ms_channel=ms_channel, double_ended=args['double_ended'],
        dim=args['dim'], filename_column_ref=None)
    t0 = time.perf_counter()
    if args['dim']:
        y_hat, _ = NL.predict_dim(model, ds, bs, 'cpu', num_workers=num_workers)
    else:
        y_hat, _ = NL.predict_mos(model, ds, bs, 'cpu', num_workers=num_workers)
    if timings is not None:
        timings['predict_s'] = time.perf_counter() - t0
    return y_hat.astype('float32')


def reference_train_step(args, state_dict, data_dir, files, labels, bs, lr=1e-3, steps=1):
    """The reference's own training step on the CPU (NISQA_model.py:96-152): its SpeechQualityDataset (get_librosa_melspec ->
    segment_specs padded to [ms_max_segments, 1, 48, 15]) behind a torch DataLoader of batch size ``bs`` (shuffle off so that the
    timed batch is the first ``bs`` files), model(x, n_wins) of its NISQA / NISQA_DIM in train mode, biasLoss.get_loss
    (NL:1879-1894; first_order mapping, before any bias update: the plain NaN-aware MSE), loss.backward(), torch.optim.Adam.step().
    librosa's three entry points are served by oracle/mel.py (PARITY UNPINNED there).  labels: [N] MOS values.
    -> dict(seconds per step incl. the DataLoader fetch, loss of the last step, segments of the last batch)."""
    import time
    import pandas as pd
    import torch
    from torch.utils.data import DataLoader
    NL = import_reference_lib(functional_librosa=True)
    args = dict(args)
    args.setdefault('double_ended', False)
    model = {'NISQA': NL.NISQA, 'NISQA_DIM': NL.NISQA_DIM}[args['model']](**{k: args[k] for k in MODEL_ARG_KEYS})
    model.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=True)
    df = pd.DataFrame({'deg': list(files), 'mos': [float(v) for v in labels], 'db': ['bench'] * len(files)})
    ds = NL.SpeechQualityDataset(
        df, df_con=None, data_dir=data_dir, filename_column='deg', mos_column='mos', seg_length=args['ms_seg_length'],
        max_length=args['ms_max_segments'], to_memory=None, to_memory_workers=None, seg_hop_length=args['ms_seg_hop_length'],
        transform=None, ms_n_fft=args['ms_n_fft'], ms_hop_length=args['ms_hop_length'], ms_win_length=args['ms_win_length'],
        ms_n_mels=args['ms_n_mels'], ms_sr=args['ms_sr'], ms_fmax=args['ms_fmax'], ms_channel=None, double_ended=False,
        dim=False, filename_column_ref=None)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    bias = NL.biasLoss(df.db, anchor_db=None, mapping='first_order', min_r=0.7, do_print=False)
    dl = DataLoader(ds, batch_size=bs, shuffle=False, drop_last=False, pin_memory=False, num_workers=0)
    model.train()
    out = {'seconds': [], 'loss': None, 'segments': None}
    it = iter(dl)
    for _ in range(steps):
        t0 = time.perf_counter()
        try:
            xb, yb, (idx, n_wins) = next(it)
        except StopIteration:                              # the next epoch over the same files, as the reference's loop would
            it = iter(dl)
            xb, yb, (idx, n_wins) = next(it)
        y_hat = model(xb, n_wins)
        loss = bias.get_loss(yb, y_hat, idx)
        loss.backward()
        opt.step()
        opt.zero_grad()
        out['seconds'].append(time.perf_counter() - t0)
        out['loss'], out['segments'] = float(loss.item()), int(n_wins.sum())
    return out

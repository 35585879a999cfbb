"""Synthetic workload of SURVEY.md section 8(d): speech-like 48 kHz clips from seeds.

Clip ``i`` uses ``numpy.random.default_rng(1000 + i)``: white noise through a
random 2-pole low-pass, times a 2-8 Hz raised-cosine envelope (so both the
``amin`` floor and the ``top_db`` clamp of the dB stage are exercised),
peak-normalised to a level drawn uniformly from [0.05, 0.9].  Used by the
tests, the golden-vector generator and bench.py; there is no dataset access
in this environment.
"""
import numpy as np


def synth_clip(i, seconds=10.0, sr=48000):
    """float64 waveform in [-0.9, 0.9] for clip index ``i`` (deterministic)."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(1000 + int(i))
    n = int(round(seconds * sr))
    noise = rng.standard_normal(n)
    r = rng.uniform(0.80, 0.98)                       # pole radius
    theta = rng.uniform(0.02, 0.5) * np.pi           # pole angle
    y = lfilter([1.0], [1.0, -2.0 * r * np.cos(theta), r * r], noise)
    f_env = rng.uniform(2.0, 8.0)
    phase = rng.uniform(0.0, 2.0 * np.pi)
    t = np.arange(n) / float(sr)
    env = 0.5 - 0.5 * np.cos(2.0 * np.pi * f_env * t + phase)
    y = y * env
    level = rng.uniform(0.05, 0.9)
    peak = np.max(np.abs(y))
    if peak > 0:
        y = y * (level / peak)
    return y


def to_pcm16(y):
    """Quantise to int16 the way a PCM16 WAV writer would."""
    return np.clip(np.round(np.asarray(y) * 32767.0), -32768, 32767).astype(np.int16)


def synth_pcm16(i, seconds=10.0, sr=48000):
    return to_pcm16(synth_clip(i, seconds, sr))


def edge_clip(kind, sr=48000):
    """Edge cases every parity set carries."""
    if kind == 'zeros':
        return np.zeros(sr * 2, dtype=np.int16)
    if kind == 'sine':                                 # full-scale 1 kHz sine
        t = np.arange(sr * 2) / float(sr)
        return to_pcm16(np.sin(2 * np.pi * 1000.0 * t))
    if kind == 'min':                                  # exactly 15 frames -> n_wins = 1
        return synth_pcm16(900, seconds=14 * 480 / float(sr) + 1e-9, sr=sr)[:14 * 480]
    if kind == 'max':                                  # 1300 segments, the cap (52 s)
        n = (1300 * 4 - 4 + 14) * 480
        return synth_pcm16(901, seconds=n / float(sr), sr=sr)[:n]
    raise KeyError(kind)


def write_wav(path, pcm, sr=48000, g711=None):
    """Minimal PCM WAV writer (mono [n] or multi-channel [n, ch], int16/int32/uint8/float32); with g711 = 'alaw' /
    'mulaw' the uint8 array holds G.711 code words (format tags 6 / 7)."""
    import struct
    pcm = np.ascontiguousarray(pcm)
    ch = 1 if pcm.ndim == 1 else pcm.shape[1]
    if g711 is not None:
        assert pcm.dtype == np.uint8
        fmt, bits = {'alaw': 6, 'mulaw': 7}[g711], 8
    elif pcm.dtype == np.float32:
        fmt, bits = 3, 32
    elif pcm.dtype == np.int16:
        fmt, bits = 1, 16
    elif pcm.dtype == np.int32:
        fmt, bits = 1, 32
    elif pcm.dtype == np.uint8:
        fmt, bits = 1, 8
    else:
        raise TypeError(pcm.dtype)
    data = pcm.tobytes()
    blk = ch * bits // 8
    hdr = b'RIFF' + struct.pack('<I', 36 + len(data)) + b'WAVE' + b'fmt ' + struct.pack(
        '<IHHIIHH', 16, fmt, ch, sr, sr * blk, blk, bits) + b'data' + struct.pack('<I', len(data))
    with open(path, 'wb') as f:
        f.write(hdr + data)


# args of weights/nisqa.tar that the hot path reads (copied VALUES, read from the checkpoint;
# used to build random-init models where the checkpoint file is absent: tests, bench.py).
DIM_ARGS = {
    'model': 'NISQA_DIM', 'name': 'rand_dim',
    'ms_sr': None, 'ms_fmax': 20000, 'ms_n_fft': 4096, 'ms_hop_length': 0.01, 'ms_win_length': 0.02,
    'ms_n_mels': 48, 'ms_seg_length': 15, 'ms_seg_hop_length': 4, 'ms_max_segments': 1300,
    'cnn_model': 'adapt', 'cnn_c_out_1': 16, 'cnn_c_out_2': 32, 'cnn_c_out_3': 64,
    'cnn_kernel_size': (3, 3), 'cnn_dropout': 0.2, 'cnn_fc_out_h': None,
    'cnn_pool_1': [24, 7], 'cnn_pool_2': [12, 5], 'cnn_pool_3': [6, 3],
    'td': 'self_att', 'td_sa_d_model': 64, 'td_sa_nhead': 1, 'td_sa_pos_enc': False,
    'td_sa_num_layers': 2, 'td_sa_h': 64, 'td_sa_dropout': 0.1,
    'td_lstm_h': None, 'td_lstm_num_layers': None, 'td_lstm_dropout': None, 'td_lstm_bidirectional': None,
    'td_2': 'skip', 'td_2_sa_d_model': None, 'td_2_sa_nhead': None, 'td_2_sa_pos_enc': None,
    'td_2_sa_num_layers': None, 'td_2_sa_h': None, 'td_2_sa_dropout': None, 'td_2_lstm_h': None,
    'td_2_lstm_num_layers': None, 'td_2_lstm_dropout': None, 'td_2_lstm_bidirectional': None,
    'pool': 'att', 'pool_att_h': 128, 'pool_att_dropout': 0, 'tr_parallel': False,
    'dim': True, 'double_ended': False,
}
MOS_ARGS = dict(DIM_ARGS, model='NISQA', name='rand_mos', dim=False)
# args of weights/nisqa_tts.tar (StandardCNN + fc 768->20 + BiLSTM(128) + last-step pooling; hop 1, fmax 8000)
TTS_ARGS = dict(DIM_ARGS, model='NISQA', name='rand_tts', dim=False, ms_fmax=8000, ms_seg_hop_length=1,
                ms_max_segments=6000, cnn_model='standard', cnn_fc_out_h=20, cnn_pool_1=None, cnn_pool_2=None,
                cnn_pool_3=None, td='lstm', td_sa_d_model=None, td_sa_nhead=None, td_sa_pos_enc=None,
                td_sa_num_layers=None, td_sa_h=None, td_sa_dropout=None, td_lstm_h=128, td_lstm_num_layers=1,
                td_lstm_dropout=0, td_lstm_bidirectional=True, pool='last_step_bi', pool_att_h=None,
                pool_att_dropout=None)




def random_state_dict(seed, model='NISQA_DIM'):
    """Seeded random-init weights with the exact key set / shapes of nisqa.tar (or nisqa_mos_only.tar).

    BatchNorm gets non-trivial running stats and a few NEGATIVE gammas (the BN fold must not
    assume positive scale); magnitudes keep activations O(1) through six conv layers.
    """
    import torch
    g = torch.Generator().manual_seed(int(seed))

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    sd = {}
    chans = [1, 16, 32, 64, 64, 64, 64]
    for i in range(1, 7):
        cin, cout = chans[i - 1], chans[i]
        sd['cnn.model.conv%d.weight' % i] = rn(cout, cin, 3, 3, std=(2.0 / (cin * 9)) ** 0.5)
        sd['cnn.model.conv%d.bias' % i] = rn(cout, std=0.1)
        gamma = 1.0 + rn(cout, std=0.2)
        gamma[::7] = -gamma[::7]
        sd['cnn.model.bn%d.weight' % i] = gamma
        sd['cnn.model.bn%d.bias' % i] = rn(cout, std=0.2)
        sd['cnn.model.bn%d.running_mean' % i] = rn(cout, std=0.3)
        sd['cnn.model.bn%d.running_var' % i] = 0.5 + torch.rand(cout, generator=g)
        sd['cnn.model.bn%d.num_batches_tracked' % i] = torch.tensor(100, dtype=torch.int64)
    # conv1 sees dB values of magnitude ~40: keep its output O(1)
    sd['cnn.model.conv1.weight'] = sd['cnn.model.conv1.weight'] * 0.05
    if model == 'NISQA_TTS':
        sd['cnn.model.fc_out.weight'] = rn(20, 768, std=768 ** -0.5)
        sd['cnn.model.fc_out.bias'] = rn(20, std=0.1)
        lp = 'time_dependency.model.lstm.'
        for sfx in ('', '_reverse'):
            sd[lp + 'weight_ih_l0' + sfx] = rn(512, 20, std=0.3)
            sd[lp + 'weight_hh_l0' + sfx] = rn(512, 128, std=0.15)
            sd[lp + 'bias_ih_l0' + sfx] = rn(512, std=0.1)
            sd[lp + 'bias_hh_l0' + sfx] = rn(512, std=0.1)
        sd['pool.model.linear.weight'] = rn(1, 256, std=0.2)
        sd['pool.model.linear.bias'] = 3.0 + rn(1, std=0.1)
        return sd
    td = 'time_dependency.model.'
    sd[td + 'norm1.weight'] = 1.0 + rn(64, std=0.1)
    sd[td + 'norm1.bias'] = rn(64, std=0.1)
    sd[td + 'linear.weight'] = rn(64, 384, std=384 ** -0.5)
    sd[td + 'linear.bias'] = rn(64, std=0.1)
    for l in range(2):
        p = td + 'layers.%d.' % l
        sd[p + 'self_attn.in_proj_weight'] = rn(192, 64, std=0.25)
        sd[p + 'self_attn.in_proj_bias'] = rn(192, std=0.1)
        sd[p + 'self_attn.out_proj.weight'] = rn(64, 64, std=0.125)
        sd[p + 'self_attn.out_proj.bias'] = rn(64, std=0.1)
        sd[p + 'linear1.weight'] = rn(64, 64, std=0.125)
        sd[p + 'linear1.bias'] = rn(64, std=0.1)
        sd[p + 'linear2.weight'] = rn(64, 64, std=0.125)
        sd[p + 'linear2.bias'] = rn(64, std=0.1)
        for n in ('norm1', 'norm2'):
            sd[p + n + '.weight'] = 1.0 + rn(64, std=0.1)
            sd[p + n + '.bias'] = rn(64, std=0.1)
    heads = ['pool_layers.%d.model.' % h for h in range(5)] if model == 'NISQA_DIM' else ['pool.model.']
    for p in heads:
        sd[p + 'linear1.weight'] = rn(128, 64, std=0.125)
        sd[p + 'linear1.bias'] = rn(128, std=0.1)
        sd[p + 'linear2.weight'] = rn(1, 128, std=0.3)
        sd[p + 'linear2.bias'] = rn(1, std=0.1)
        sd[p + 'linear3.weight'] = rn(1, 64, std=0.125)
        sd[p + 'linear3.bias'] = 3.0 + rn(1, std=0.1)
    return sd




def eval_corpus(seed, n_db=3, n_con=12, per_con=6):
    """Seeded synthetic "listening test" for the evaluation statistics (tests/golden/make_golden_eval.py):
    -> (per-file frame: db, con, filepath_deg, mos, mos_pred; per-condition frame: db, con, mos, mos_ci)."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    files, cons = [], []
    for d in range(n_db):
        db = 'DB_%c' % 'CAB'[d % 3] + ('' if d < 3 else str(d))       # not in sorted order on purpose
        gain, off = 0.8 + 0.15 * d, 0.3 - 0.2 * d                       # per-database bias of the predictor
        for c in range(1, n_con + 1):
            q = rng.uniform(1.2, 4.8)
            votes = np.clip(q + rng.normal(0, 0.45, per_con), 1, 5)
            pred = np.clip(off + gain * votes + 0.05 * (votes - 3) ** 2 + rng.normal(0, 0.25, per_con), 0.5, 5.5)
            for k in range(per_con):
                files.append({'db': db, 'con': c, 'filepath_deg': '%s/c%02d_f%d.wav' % (db, c, k), 'mos': votes[k],
                              'mos_pred': pred[k]})
            cons.append({'db': db, 'con': c, 'mos': votes.mean(),
                         'mos_ci': 1.96 * votes.std(ddof=1) / np.sqrt(per_con)})
    df = pd.DataFrame(files).sample(frac=1.0, random_state=seed).reset_index(drop=True)    # shuffled file order
    return df, pd.DataFrame(cons)

"""Host-side mirror of the reference's hot-path interface in nisqa/NISQA_lib.py: same names,
argument meaning and error behaviour, with the arithmetic moved to the HIP engine.

  SpeechQualityDataset  (reference NL:2052-2236)  file table + ms_* parameters; items are read here,
                                                   spectrograms are computed on the GPU
  predict_mos / predict_dim (NL:1420-1467)        batching loop: WAV ingest -> device -> HIP forward
  NISQA / NISQA_DIM     (NL:29-268)                parameter containers with the reference's
                                                   state_dict keys; forward() runs the HIP path

Evaluation statistics (eval_results and helpers, NL:1469-1852) are re-exported from nisqa_amd/evaluation.py.
Out of scope here (raise NotImplementedError): training, NISQA_DE,
alternative blocks no shipped checkpoint uses.
"""
import os
import sys
import time
from contextlib import nullcontext as _nullcontext

import numpy as np
import pandas as pd; pd.options.mode.chained_assignment = None
import torch
import torch.nn as nn

from . import dist as _dist
from .evaluation import (calc_eval_metrics, calc_mapped, calc_mapping, calc_rmse, calc_rmse_star, eval_results,  # noqa: F401
                         fit_first_order, fit_monotonic_third_order, fit_second_order, fit_third_order, is_const)
from . import ingest as _ingest
from . import lib as _lib_mod
from .wavio import read_wav


# ---------------------------------------------------------------------------------------------
# Parameter containers: identical module tree / state_dict keys as the reference so that
# nisqa.tar loads with load_state_dict(strict=True) (NISQA_model.py:1023).
# ---------------------------------------------------------------------------------------------
class _Params(nn.Module):
    """Leaf holding named tensors (weights as Parameters, running stats as buffers)."""

    def __init__(self, shapes, buffers=()):
        super().__init__()
        for name, shape in shapes.items():
            self.register_parameter(name, nn.Parameter(torch.zeros(shape), requires_grad=False))
        for name, shape, dtype in buffers:
            self.register_buffer(name, torch.zeros(shape, dtype=dtype))


def _conv(cout, cin, kh, kw):
    return _Params({'weight': (cout, cin, kh, kw), 'bias': (cout,)})


def _bn(c):
    return _Params({'weight': (c,), 'bias': (c,)},
                   [('running_mean', (c,), torch.float32), ('running_var', (c,), torch.float32),
                    ('num_batches_tracked', (), torch.int64)])


def _lin(cout, cin):
    return _Params({'weight': (cout, cin), 'bias': (cout,)})


class _AdaptCNNParams(nn.Module):
    def __init__(self, c1, c2, c3, kernel_size, pool_3):
        super().__init__()
        kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        chans = [1, c1, c2, c3, c3, c3, c3]
        for i in range(1, 7):
            k_w = pool_3[1] if i == 6 else kw                       # NL:625, NL:672-676
            setattr(self, 'conv%d' % i, _conv(chans[i], chans[i - 1], kh, k_w))
            setattr(self, 'bn%d' % i, _bn(chans[i]))
        self.fan_out = c3 * pool_3[0]


class _SALayerParams(nn.Module):
    def __init__(self, d, h):
        super().__init__()
        self.self_attn = nn.Module()
        self.self_attn.register_parameter('in_proj_weight', nn.Parameter(torch.zeros(3 * d, d), requires_grad=False))
        self.self_attn.register_parameter('in_proj_bias', nn.Parameter(torch.zeros(3 * d), requires_grad=False))
        self.self_attn.out_proj = _lin(d, d)
        self.linear1 = _lin(h, d)
        self.linear2 = _lin(d, h)
        self.norm1 = _Params({'weight': (d,), 'bias': (d,)})
        self.norm2 = _Params({'weight': (d,), 'bias': (d,)})


class _SelfAttentionParams(nn.Module):
    def __init__(self, input_size, d, layers, h):
        super().__init__()
        self.norm1 = _Params({'weight': (d,), 'bias': (d,)})
        self.linear = _lin(d, input_size)
        self.layers = nn.ModuleList([_SALayerParams(d, h) for _ in range(layers)])


class _PoolAttFFParams(nn.Module):
    def __init__(self, d, h):
        super().__init__()
        self.linear1 = _lin(h, d)
        self.linear2 = _lin(1, h)
        self.linear3 = _lin(1, d)


class _StandardCNNParams(nn.Module):
    def __init__(self, c1, c2, c3, fc_out_h):
        super().__init__()
        chans = [1, c1, c2, c3, c3, c3, c3]
        for i in range(1, 7):
            setattr(self, 'conv%d' % i, _conv(chans[i], chans[i - 1], 3, 3))
            setattr(self, 'bn%d' % i, _bn(chans[i]))
        self.fc_out = _lin(fc_out_h, c3 * 6 * 2)                   # NL:803-807
        self.fan_out = fc_out_h


class _LSTMParams(nn.Module):
    def __init__(self, input_size, lstm_h):
        super().__init__()
        # parameter holder with nn.LSTM's own key names (weight_ih_l0, ..., *_reverse); never called
        self.lstm = nn.LSTM(input_size=input_size, hidden_size=lstm_h, num_layers=1, batch_first=True, bidirectional=True)
        for p_ in self.lstm.parameters():
            p_.requires_grad_(False)


class _PoolLastStepBiParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.linear = _lin(1, d)


class _Wrap(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model


class _NisqaBase(nn.Module):
    """Shared by NISQA and NISQA_DIM: builds the parameter tree, owns the (lazy) HIP engine."""

    def __init__(self, n_heads, **kw):
        super().__init__()
        g = lambda k, d=None: kw.get(k, d)
        self._hp = dict(kw)
        self._engine = None
        self._engine_args = None
        if g('cnn_model') == 'standard' and g('td') == 'lstm' and g('pool') == 'last_step_bi' \
                and g('td_2', 'skip') in (None, 'skip') and n_heads == 1:
            # nisqa_tts.tar architecture (NL:712-836, NL:897-943, NL:1099-1115)
            self.cnn = _Wrap(_StandardCNNParams(g('cnn_c_out_1', 16), g('cnn_c_out_2', 32), g('cnn_c_out_3', 64),
                                                g('cnn_fc_out_h', 20)))
            self.time_dependency = _Wrap(_LSTMParams(self.cnn.model.fan_out, g('td_lstm_h', 128)))
            self.pool = _Wrap(_PoolLastStepBiParams(2 * g('td_lstm_h', 128)))
            return
        if g('cnn_model', 'adapt') != 'adapt' or g('td', 'self_att') != 'self_att' or g('pool', 'att') != 'att' \
                or g('td_2', 'skip') not in (None, 'skip') or not g('pool_att_h', 128):
            raise NotImplementedError(
                'nisqa_amd accelerates the CNN-SA-AP path (cnn_model=adapt, td=self_att, td_2=skip, pool=att with '
                'pool_att_h) and the nisqa_tts path (cnn_model=standard, td=lstm, pool=last_step_bi); got '
                'cnn_model={} td={} td_2={} pool={}'.format(g('cnn_model'), g('td'), g('td_2'), g('pool')))
        self.cnn = _Wrap(_AdaptCNNParams(g('cnn_c_out_1', 16), g('cnn_c_out_2', 32), g('cnn_c_out_3', 64),
                                         g('cnn_kernel_size', 3), g('cnn_pool_3', [6, 3])))
        d = g('td_sa_d_model', 64)
        self.time_dependency = _Wrap(_SelfAttentionParams(self.cnn.model.fan_out, d, g('td_sa_num_layers', 2),
                                                          g('td_sa_h', 64)))
        if n_heads == 5:
            self.pool_layers = nn.ModuleList([_Wrap(_PoolAttFFParams(d, g('pool_att_h', 128))) for _ in range(5)])
        else:
            self.pool = _Wrap(_PoolAttFFParams(d, g('pool_att_h', 128)))

    def bind_args(self, args):
        """Give the module the checkpoint's args (ms_* front-end parameters live there, NISQA_model.py:941-942)."""
        self._engine_args = args
        self._engine = None
        return self

    def engine(self, device=None):
        if self._engine is None:
            from .engine import HipNisqa
            if self._engine_args is None:
                raise RuntimeError('bind_args(checkpoint_args) must be called before the HIP engine is built')
            self._engine = HipNisqa(self._engine_args, self.state_dict(), device)
        return self._engine

    def forward(self, x, n_wins):
        """Reference inner operator model(x[B,L,1,48,15], n_wins[B]) -> [B, heads] (NL:137-142, NL:260-268)."""
        dev = x.device if x.is_cuda else None
        return self.engine(dev).forward_segments(x, n_wins)


def init_parameters_(model):
    """Random initialisation with the distributions the reference's modules get from PyTorch (for training from
    scratch, pretrained_model: false): Conv2d / Linear kaiming-uniform(a=sqrt 5) weights and U(+-1/sqrt(fan_in)) biases,
    BatchNorm / LayerNorm at identity with fresh running statistics, nn.MultiheadAttention's xavier-uniform in_proj and
    zero attention biases, and SelfAttention._reset_parameters (NL:981-984): xavier-uniform on every matrix of the
    time-dependency block."""
    import math
    for name, mod in model.named_modules():
        if not isinstance(mod, _Params):
            continue
        names = dict(mod.named_parameters(recurse=False))
        bufs = dict(mod.named_buffers(recurse=False))
        w = names.get('weight')
        if 'running_mean' in bufs or (w is not None and w.dim() == 1):         # BatchNorm / LayerNorm
            nn.init.ones_(w)
            nn.init.zeros_(names['bias'])
            if 'running_mean' in bufs:
                bufs['running_mean'].zero_()
                bufs['running_var'].fill_(1.0)
                bufs['num_batches_tracked'].zero_()
        elif w is not None:                                                      # Conv2d / Linear
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            fan_in = w[0].numel()
            nn.init.uniform_(names['bias'], -1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
    td = getattr(model, 'time_dependency', None)
    if td is not None and isinstance(td.model, _SelfAttentionParams):
        for layer in td.model.layers:
            nn.init.zeros_(layer.self_attn.in_proj_bias)
            nn.init.zeros_(layer.self_attn.out_proj.bias)
        for p_ in td.model.parameters():
            if p_.dim() > 1:
                nn.init.xavier_uniform_(p_)
    return model


class NISQA(_NisqaBase):
    def __init__(self, **kw):
        super().__init__(1, **kw)
        self.name = 'NISQA'


class NISQA_DIM(_NisqaBase):
    def __init__(self, **kw):
        super().__init__(5, **kw)
        self.name = 'NISQA_DIM'


# ---------------------------------------------------------------------------------------------
# Dataset mirror
# ---------------------------------------------------------------------------------------------
class SpeechQualityDataset(object):
    """File table of the reference dataset (NL:2052-2236).  Same constructor arguments; loading a
    waveform replaces ``_load_spec`` (the spectrogram itself is produced on the GPU)."""

    def __init__(self, df, df_con=None, data_dir='', folder_column='', filename_column='filename', mos_column='MOS',
                 seg_length=15, max_length=None, to_memory=False, to_memory_workers=0, transform=None,
                 seg_hop_length=1, ms_n_fft=1024, ms_hop_length=80, ms_win_length=170, ms_n_mels=32, ms_sr=48e3,
                 ms_fmax=16e3, ms_channel=None, double_ended=False, filename_column_ref=None, dim=False):
        if double_ended:
            raise NotImplementedError('double-ended (NISQA_DE) datasets are out of scope')
        if transform is not None:
            raise NotImplementedError('spectrogram transforms are not supported on the HIP path')
        self.df, self.df_con, self.data_dir = df, df_con, data_dir
        self.filename_column, self.mos_column = filename_column, mos_column
        self.seg_length, self.seg_hop_length, self.max_length = seg_length, seg_hop_length, max_length
        self.ms_n_fft, self.ms_hop_length, self.ms_win_length = ms_n_fft, ms_hop_length, ms_win_length
        self.ms_n_mels, self.ms_sr, self.ms_fmax, self.ms_channel = ms_n_mels, ms_sr, ms_fmax, ms_channel
        self.dim = dim

    def __len__(self):
        return len(self.df)

    def file_path(self, index):
        return os.path.join(self.data_dir, self.df[self.filename_column].iloc[index])   # NL:2132

    def file_paths(self, indices):
        """Paths of many items at once: the filename column is read out once per CALL (a pandas scalar lookup per item
        costs more host time than staging the item's samples; nothing is cached on the dataset, so an in-place edit of
        the column is always seen)."""
        names, d = self.df[self.filename_column].tolist(), self.data_dir
        return [os.path.join(d, names[i]) for i in indices]

    def load_audio(self, index):
        """(samples, sr) of item ``index``; raises ValueError('Could not load file ...') like NL:2305-2306."""
        return read_wav(self.file_path(index), self.ms_channel)

    def bind_engine(self, factory):
        """``factory()`` -> HipNisqa; lets __getitem__ produce spectrogram segments like the reference dataset."""
        self._engine_factory = factory
        return self

    def __getitem__(self, index):
        """(x_spec_seg [max_length,1,n_mels,seg_length], y, (index, n_wins)) like NL:2162-2233.  The spectrogram is
        computed by the HIP front end; the overlapping-window gather (NL:2266-2280) is a host-side view for callers
        that want the reference's item format -- the predict loop never materialises it."""
        assert isinstance(index, int), 'index must be integer (no slice)'
        if getattr(self, '_engine_factory', None) is None:
            raise RuntimeError('SpeechQualityDataset item access needs bind_engine(...) (spectrograms are computed on the GPU)')
        eng = self._engine_factory()
        y, sr = self.load_audio(index)
        plan = eng.audio_plan([len(y)], sr, names=[self.file_path(index)])
        pcm = eng.resample(torch.from_numpy(y).to(eng.device), [len(y)], sr)          # (ms_sr: lb.load resamples first, NL:2300-2304)
        mel, _ = eng.mel(pcm, plan, eng.rate(sr), clamp=True)                         # int16 PCM or float32 samples
        spec = mel.cpu().numpy()                                   # [T, n_mels]
        n_wins = int(plan.n_wins[0])
        idx = self.seg_hop_length * np.arange(n_wins)[:, None] + np.arange(self.seg_length)[None, :]
        x = np.transpose(spec[idx], (0, 2, 1))[:, None]           # [n_wins, 1, n_mels, seg_length]
        if self.max_length is not None:
            pad = np.zeros((self.max_length,) + x.shape[1:], np.float32)
            pad[:n_wins] = x
            x = pad
        return torch.from_numpy(np.ascontiguousarray(x)), self.label(index), (index, np.array(n_wins))

    def label(self, index):
        """y of item ``index`` like NL:2217-2231: NaN in predict_only mode, else row ``index`` of the MOS column(s)."""
        if self.mos_column == 'predict_only':
            return np.full(5 if self.dim else 1, np.nan, dtype=np.float32)
        cols = ['mos', 'noi', 'dis', 'col', 'loud'] if self.dim else [self.mos_column]
        return np.array([self.df[c].iloc[index] for c in cols], dtype=np.float32)

    def labels(self, n):
        """Labels of the first n items like NL:2217-2231: NaN rows in predict_only mode, else the csv columns."""
        if self.mos_column == 'predict_only':
            return np.full((n, 5 if self.dim else 1), np.nan, dtype=np.float32)
        cols = ['mos', 'noi', 'dis', 'col', 'loud'] if self.dim else [self.mos_column]
        return np.stack([self.df[c].to_numpy(dtype=np.float32)[:n] for c in cols], 1)


# ---------------------------------------------------------------------------------------------
# Batching loop
# ---------------------------------------------------------------------------------------------
_STREAMS = {}


def _loop_streams(device):
    """(copy stream, [kernel stream, kernel stream]) of the predict loop on ``device``, created once per process.
    The copy stream has high priority = a hardware queue of its own: streams of one priority share a small pool of
    hardware queues round-robin, and a copy stream that lands on the queue of a kernel stream is serialised behind its
    kernels."""
    key = str(device)
    if key not in _STREAMS:
        nk = int(os.environ.get('NISQA_LOOP_KERNEL_STREAMS', '2'))
        ks = [torch.cuda.Stream(device=device) for _ in range(max(1, min(nk, 2)))]
        _STREAMS[key] = (torch.cuda.Stream(device=device, priority=-1), [ks[0], ks[-1]])
    return _STREAMS[key]


LOOP_STATS = {}                 # host seconds of the last _predict call by phase (tools/probe_loop.py)


# work a launch chain should carry before a batch is closed (batch_policy): the AdaptCNN kernel runs 512 four-segment
# workgroups at a time (2 per CU), so 16 k segments are ~8 rounds -- the size of the bs = 64 x 10 s configuration the
# kernels were tuned on; the BiLSTM of the nisqa_tts.tar path runs ONE workgroup per (clip, direction) for as many
# sequential steps as the longest clip has segments: 128 clips x 2 directions fill the 256 CUs
MIN_TOKENS_SA = 16384
MIN_CLIPS_LSTM = 128
BATCH_BYTE_CAP = 256 << 20          # staged PCM per batch (the page-locked ring holds three such slots)


def tokens_of(ds, n_frames, sample_rate):
    """Segments per clip from WAV header fields alone: frames T = 1 + samples // hop (librosa centre framing, NL:2311),
    n_wins = ceil((T - (seg_length - 1)) / seg_hop) (NL:2256-2273).  hop as melbank.MelTables derives it."""
    sr = np.asarray(sample_rate, dtype=np.float64)
    n_frames = np.asarray(n_frames, dtype=np.int64)
    if getattr(ds, 'ms_sr', None) is not None:                     # lb.load(..., sr=ms_sr): ceil(n * ms_sr / sr) samples at ms_sr
        n_frames = np.ceil(n_frames.astype(np.float64) * (float(ds.ms_sr) / np.maximum(sr, 1.0))).astype(np.int64)
        sr = np.full_like(sr, float(ds.ms_sr))
    hop = np.maximum(1, (sr * float(ds.ms_hop_length)).astype(np.int64))
    T = 1 + n_frames // hop
    return np.maximum(1, -(-(T - (int(ds.seg_length) - 1)) // max(1, int(ds.seg_hop_length))))


def batch_policy(eng, ds, indices, bs):
    """Length-aware batches for the predict loop (ingest.LengthAware): --bs is a lower bound, batches are cut by
    segments / clips / staged bytes after sorting a window of items by length."""
    lstm = getattr(eng, 'arch', 0) == 1
    # Large jobs get larger batches: an H2D copy carries ~80 us that does not scale with its size (64 MB copies run at 53 GB/s inside
    # the loop where 245 MB copies run at 56), but a job needs a few hundred batches to keep the pipeline's fill and drain small --
    # so the floor grows with the job, 1 x (<= 25 k items) ... 4 x (>= 100 k items) of MIN_TOKENS_SA.  Measured on a directory of
    # 98 304 ten-second files at --bs 64: 48.5 k clips/s (1 x) / 50.0-50.9 k (2 x) / 50.3-51.2 k (4 x); on 32 768 files 4 x is the
    # slowest (128 batches).  Rows do not depend on the batch composition (tested).
    n_items = len(indices)
    scale = min(4, max(1, n_items // (384 * 66)))
    return _ingest.LengthAware(indices, bs, lambda f, r: tokens_of(ds, f, r),
                               min_tokens=0 if lstm else int(os.environ.get('NISQA_MIN_TOKENS', MIN_TOKENS_SA * scale)),
                               min_clips=int(os.environ.get('NISQA_MIN_CLIPS', MIN_CLIPS_LSTM)) if lstm else 1,
                               byte_cap=int(os.environ.get('NISQA_BATCH_BYTES', BATCH_BYTE_CAP)))


def _predict(model, ds, bs, dev, num_workers, on_rows=None):
    """Shared body of predict_mos / predict_dim: returns y_hat [N, heads] float32 for ALL items of ds
    (clip-sharded over ranks when torch.distributed is initialised, then gathered).
    on_rows(ids, rows): optional callback with every batch's item indices and [B, heads] float32 rows as they come back from
    the device (nisqaModel.predict formats the table cells there, under the next batches' transfers)."""
    dev = torch.device(dev)
    if model._engine is None and dev.type != 'cuda':
        raise RuntimeError('nisqa_amd has no CPU path: device {} requested but the hot path runs only as HIP kernels '
                           'on an MI355X'.format(dev))
    eng = model.engine(dev if dev.index is not None else None)
    on_gpu = eng.device.type == 'cuda'
    n = len(ds)
    lo, hi = _dist.shard_range(n)
    bounds = None
    rank, world = _dist.world()
    if world > 1 and os.environ.get('NISQA_SHARD_BY_COUNT') != '1':
        # work-balanced contiguous shards: every rank probes the RIFF headers of its count share (native threads, headers
        # only), the per-clip segment counts are summed into one vector on all ranks, and the shard boundaries follow its
        # prefix sum -- a length-sorted list of mixed 3-30 s clips otherwise gives the last rank several times the first
        # one's work.  An unreadable header counts as one segment here; the error itself is raised where the reference
        # raises it, when the file is loaded.
        # A rank whose probe itself fails (ingest library missing, no filename column ...) still enters both collectives:
        # the others would otherwise wait in all_reduce for ever (dist.raise_together).
        tok = np.zeros(n, dtype=np.int64)
        probe_err = None
        try:
            if hi > lo:
                info = _ingest.probe_headers(ds, range(lo, hi), num_workers)
                ok = info['status'] == _lib_mod.WAV_OK
                tok[lo:hi] = np.where(ok, tokens_of(ds, np.where(ok, info['n_frames'], 0), np.where(ok, info['sample_rate'], 48000)), 1)
        except Exception as e:                                      # noqa: BLE001 (re-raised on every rank below)
            probe_err = e
            tok[lo:hi] = 0
        tok = _dist.all_reduce_sum_i64(tok)
        _dist.raise_together(probe_err)
        bounds = _dist.balanced_bounds(tok, world)
        lo, hi = bounds[rank]
    loop_err, ing, copy_events = None, None, []
    # Three Python threads share the interpreter lock here: this one (enqueue, results, table cells), the producer (staging) and the
    # helper that probes and cuts the NEXT window of 16 384 files -- 20 ms of pure Python per window.  With CPython's default 5 ms
    # switch interval the helper holds the lock in 5 ms slices while the other two need it for microseconds every batch: the copy
    # stream ran dry for 2 x 6-8 ms at every window boundary (rocprofv3 --memory-copy-trace: 85 of 1 735 ms on the csv leg, 135 of
    # 1 794 on the directory leg).  A 0.1 ms interval for the duration of the loop hands the lock over before a copy's worth of
    # time has passed.
    switch_interval = sys.getswitchinterval()
    sys.setswitchinterval(min(switch_interval, float(os.environ.get('NISQA_LOOP_SWITCH_INTERVAL', '1e-4'))))
    T = {'queue_wait': 0.0, 'enqueue': 0.0, 'result_wait': 0.0}
    try:
        bs = max(1, int(bs))
        heads = eng.n_heads
        y_local = np.zeros((hi - lo, heads), dtype=np.float32)
        if os.environ.get('NISQA_EXACT_BS') == '1':                     # the reference's batches: index order, exactly bs clips
            batches = [list(range(s, min(s + bs, hi))) for s in range(lo, hi, bs)]
        else:
            batches = batch_policy(eng, ds, range(lo, hi), bs)
        # host side (ingest.py): a producer thread + num_workers readers stage batches two ahead in page-locked buffers;
        # device side: ONE stream carries nothing but the H2D copies (a stream that also carries kernels gets its copies done
        # by a shader blit that competes with them instead of the SDMA engine: copy and kernels of neighbouring batches then
        # do not overlap at all, tools/probe_overlap.py: 7.5 ms per 256-clip batch against 4.5), two streams take the
        # kernels of alternate batches behind an event, a staging slot is recycled as soon as ITS copy is done, and the D2H
        # of batch i's [B, heads] rows is waited for one batch late (no device-wide sync in the loop)
        # three batches staged ahead (four page-locked slots): the consumer below keeps TWO batches in flight, so the copy of
        # batch k + 1 is queued a whole batch time before the link needs it, not 0.8 ms before (kernels 3 ms + enqueue 0.5 ms
        # against a 4.3 ms copy: with one batch in flight any jitter left the link idle, 89 % of its rate over 98 304 rows)
        ing = _ingest.Ingest(ds, batches, pin=on_gpu, num_workers=num_workers, depth=int(os.environ.get('NISQA_LOOP_DEPTH', '3')),
                             device=eng.device if on_gpu else None)
        copy_stream, streams = _loop_streams(eng.device) if on_gpu else (None, [None, None])
        inflight = []                                                   # (ids, host rows, event behind them)
        keep_inflight = max(1, int(os.environ.get('NISQA_LOOP_INFLIGHT', '2')))
        time_copies = on_gpu and os.environ.get('NISQA_LOOP_TIME_COPIES') == '1'     # tools: HIP events around every batch's H2D copies
        clock = time.perf_counter

        def drain(keep):
            # waits on the EVENT behind a batch's rows, never on its stream: hipStreamSynchronize also waits for whatever was
            # queued on the stream (or on another stream that shares its hardware queue) after that batch, i.e. for the batch
            # just enqueued -- the loop then runs copy and kernels strictly one after the other (8.2 instead of 4.6 ms per
            # 256 clips; which it was depended on the order streams were created in)
            t0 = clock()
            while len(inflight) > keep:
                ids, rows, done = inflight.pop(0)
                if done is not None:
                    done.synchronize()
                y_local[np.asarray(ids) - lo] = rows.numpy()
                if on_rows is not None:
                    on_rows(ids, rows.numpy())
            T['result_wait'] += clock() - t0

        t_it = clock()
        for bi, staged in enumerate(ing):
            t_got = clock()
            T['queue_wait'] += t_got - t_it
            st = streams[bi % 2]
            raw = ing.ring.buf[staged.slot]
            sent, ev = [], None
            try:
                with (torch.cuda.stream(copy_stream) if on_gpu else _nullcontext()):
                    if time_copies:
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record(copy_stream)
                    for g in staged.groups:                          # files of one rate share the mel tables
                        plan = eng.audio_plan(g.lengths, g.sr, names=g.names)
                        tables = plan.to(eng.device)
                        host = raw[g.offset:g.offset + g.nbytes].view(torch.int16 if g.is_i16 else torch.float32)
                        # PCM16 stays int16: 2 bytes/sample over PCIe.  (Round 5 sent the halves of a batch through TWO copy-only
                        # streams so that one engine's set-up would be covered by the other's transfer: 48.6 / 47.7 k clips/s against
                        # 50.4 / 52.3 k with one stream on the same box -- two SDMA queues share the link worse than one fills it.)
                        pcm = host.to(eng.device, non_blocking=True)
                        sent.append((g, plan, tables, pcm))
                    if on_gpu:
                        ev = torch.cuda.Event(enable_timing=time_copies)
                        ev.record(copy_stream)
                        if time_copies:
                            copy_events.append((e0, ev))
            finally:
                ing.ring.release_after(staged.slot, ev)
            if on_gpu:
                st.wait_event(ev)
            with (torch.cuda.stream(st) if on_gpu else _nullcontext()):
                for g, plan, tables, pcm in sent:
                    out = eng.forward_audio(pcm, g.lengths, g.sr, plan)      # (resampled to ms_sr first when the checkpoint sets it)
                    if on_gpu:
                        pcm.record_stream(st)                        # allocated on the copy stream, consumed on this one
                        tables['_buf'].record_stream(st)
                        rows = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
                        rows.copy_(out, non_blocking=True)           # [B, heads] floats, behind the kernels
                        done = torch.cuda.Event()
                        done.record(st)
                        inflight.append((g.ids, rows, done))
                    else:
                        inflight.append((g.ids, out, None))
            T['enqueue'] += clock() - t_got
            drain(keep=keep_inflight * len(staged.groups))           # results of the batch before the previous one
            t_it = clock()
        drain(keep=0)
    except Exception as e:                                          # noqa: BLE001
        # the reference's errors (unreadable file, too short, too many segments: NL:2305-2306, 2259-2263, 2276-2277) are
        # raised on EVERY rank below, not only on the one whose shard holds the file -- the others would block in the
        # closing all_gather
        loop_err = e
    finally:
        sys.setswitchinterval(switch_interval)
        LOOP_STATS.clear()
        LOOP_STATS.update(T)
        if ing is not None:
            ing.close()
            LOOP_STATS.update({'producer_' + k: v for k, v in ing.stats.items()})
            LOOP_STATS['readers'] = ing.workers
        if copy_events:
            torch.cuda.synchronize()
            LOOP_STATS['copy_busy_s'] = sum(a.elapsed_time(b) for a, b in copy_events) * 1e-3
            LOOP_STATS['copy_span_s'] = copy_events[0][0].elapsed_time(copy_events[-1][1]) * 1e-3
    _dist.raise_together(loop_err)
    return _dist.gather_rows(y_local, n, lo, hi, dev, bounds)


def predict_mos(model, ds, bs, dev, num_workers=0, on_rows=None):
    """predict_mos (NL:1420-1439): fills ds.df['mos_pred'] (float64 like NL:1438), returns (y_hat, y)."""
    y_hat = _predict(model, ds, bs, dev, num_workers, on_rows)[:, :1]
    y = ds.labels(len(ds))[:, :1]
    ds.df['mos_pred'] = y_hat.astype(dtype=float)
    return y_hat, y


def predict_dim(model, ds, bs, dev, num_workers=0, on_rows=None):
    """predict_dim (NL:1441-1467): fills mos/noi/dis/col/loud_pred columns, returns (y_hat, y)."""
    y_hat = _predict(model, ds, bs, dev, num_workers, on_rows)
    y = ds.labels(len(ds))
    ds.df['mos_pred'] = y_hat[:, 0].reshape(-1, 1)
    ds.df['noi_pred'] = y_hat[:, 1].reshape(-1, 1)
    ds.df['dis_pred'] = y_hat[:, 2].reshape(-1, 1)
    ds.df['col_pred'] = y_hat[:, 3].reshape(-1, 1)
    ds.df['loud_pred'] = y_hat[:, 4].reshape(-1, 1)
    return y_hat, y

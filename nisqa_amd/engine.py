"""Device engine: owns the packed weights / tables on one GPU and drives libnisqa_hip.so.

PyTorch is used only for device memory, streams and H2D/D2H copies; all arithmetic of the hot
path (mel front end, AdaptCNN, self-attention, attention pooling) runs in the HIP kernels behind
the C ABI (include/nisqa_hip.h).  No fallback: constructing the engine without a GPU or without
the built library raises.
"""
import ctypes
import os

import numpy as np
import torch

from . import lib as _lib
from . import weights as _w
from .melbank import KAISER_BEST_PRECISION, MelTables, kaiser_best_table, resampled_lengths

SEG_LEN = 15
TOK_PAD = 64          # tokens of a clip are padded to whole 64-token workgroups (csrc/td16_bf16x6.hip); the other kernels need 32
# The precision every GEMM of the path runs in unless the caller (or NISQA_HIP_PRECISION) says otherwise.  'bf16x6' carries the
# reference's fp32 operands EXACTLY (three bf16 terms, six MFMA products per term pair, fp32 accumulation): reference-grade
# arithmetic, as far from a float64 evaluation as the exact-fp32 kernels and the reference's own CPU float32
# (tests/test_gpu_parity.py::test_rounding_error_of_the_precision_modes_against_float64).  'bf16x3' (16 of the 24 operand mantissa
# bits, |dMOS| <= 5e-5, 1.5 x faster) was the default until round 4 and stays selectable; 'f32' is the exact fp32-MFMA path.
#
# Round 5 added 'f16x4' / 'f16x3': the AdaptCNN with every fp32 operand as TWO f16 terms of the power-of-two-scaled tensor (11 + 11
# significand bits and the low term's sign: the fp32 value itself for ~75 % of the values, one fp32 ulp off for the rest) and all four /
# three term products; self-attention and pooling run as in 'bf16x6'.  Measured against float64 they are as close as 'f32' and 'bf16x6'
# (the same test), at 4 / 3 instead of 6 MFMA products -- but an operand may lose its last bit, so they are opt-in, not the default.
DEFAULT_PRECISION = 'bf16x6'
PRECISIONS = ('f32', 'bf16x3', 'bf16x6', 'f16x3', 'f16x4')
CNN_MODE = {'f32': 0, 'bf16x3': 1, 'bf16x6': 2, 'f16x3': 3, 'f16x4': 4}


class BatchPlan(object):
    """Host-side shape metadata of one batch (the only thing the host computes per batch)."""

    def __init__(self, lengths, hop, seg_hop, max_segments, names=None):
        lengths = np.asarray(lengths, dtype=np.int64).reshape(-1)
        self.n_clips = int(len(lengths))
        if self.n_clips == 0:
            raise ValueError('empty batch')
        self.lengths = lengths
        self.T = (1 + lengths // hop).astype(np.int64)             # librosa centre framing
        n_full = self.T - (SEG_LEN - 1)                              # NISQA_lib.py:2256
        for i in np.nonzero(n_full < 1)[0]:
            raise ValueError(
                'Sample too short. Only {} windows available but seg_length={}. '
                'Consider zero padding the audio sample. File: {}'.format(
                    int(self.T[i]), SEG_LEN, names[i] if names is not None else i))
        n = -(-n_full // seg_hop) if seg_hop > 1 else n_full        # NISQA_lib.py:2271-2273
        if max_segments is not None:
            for i in np.nonzero(n > max_segments)[0]:
                raise ValueError('n_wins {} > max_length {} --- {}. Increase max window length ms_max_segments!'.format(
                    int(n[i]), max_segments, names[i] if names is not None else i))
        self.n_wins = n.astype(np.int32)
        self.clip_off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        self.frame_off = np.concatenate([[0], np.cumsum(self.T)]).astype(np.int32)
        npad = (n + TOK_PAD - 1) // TOK_PAD * TOK_PAD
        self.tok_off = np.concatenate([[0], np.cumsum(npad)]).astype(np.int32)
        self.total_samples = int(self.clip_off[-1])
        self.total_frames = int(self.frame_off[-1])
        self.total_tok = int(self.tok_off[-1])
        self.dev = None

    @classmethod
    def from_n_wins(cls, n_wins):
        """Plan for callers that already hold segments (model.forward): only the token layout is needed."""
        self = cls.__new__(cls)
        n = np.asarray(n_wins, dtype=np.int64).reshape(-1)
        self.n_clips = int(len(n))
        self.lengths = np.zeros(self.n_clips, np.int64)
        self.T = np.zeros(self.n_clips, np.int64)
        self.n_wins = n.astype(np.int32)
        self.clip_off = np.zeros(self.n_clips + 1, np.int64)
        self.frame_off = np.zeros(self.n_clips + 1, np.int32)
        npad = (n + TOK_PAD - 1) // TOK_PAD * TOK_PAD
        self.tok_off = np.concatenate([[0], np.cumsum(npad)]).astype(np.int32)
        self.total_samples, self.total_frames, self.total_tok = 0, 0, int(self.tok_off[-1])
        self.dev = None
        return self

    def to(self, device):
        """The four index tables on ``device``: packed into ONE page-locked buffer and sent with one asynchronous copy.
        (Four pageable ``.to(device)`` calls each block the host until everything queued before them on the stream has
        run -- behind a batch's 245 MB PCM copy that is 4 ms of the predict loop.)"""
        if self.dev is None or self.dev['device'] != device:
            parts = (('clip_off', self.clip_off), ('frame_off', self.frame_off), ('tok_off', self.tok_off), ('n_wins', self.n_wins))
            offs, total = [], 0
            for _, a in parts:
                offs.append(total)
                total += (a.nbytes + 15) // 16 * 16
            pin = torch.device(device).type == 'cuda'
            host = torch.empty(max(total, 16), dtype=torch.uint8, pin_memory=pin)
            hv = host.numpy()
            for (_, a), o in zip(parts, offs):
                hv[o:o + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
            buf = host.to(device, non_blocking=pin)
            self.dev = {'device': device, '_buf': buf, '_host': host}
            for (k, a), o in zip(parts, offs):
                self.dev[k] = buf[o:o + a.nbytes].view(torch.from_numpy(a[:0]).dtype)
        return self.dev

    def token_index(self):
        """Indices (into the padded token axis) of the valid tokens, clip by clip."""
        return np.concatenate([self.tok_off[b] + np.arange(self.n_wins[b]) for b in range(self.n_clips)])


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class HipNisqa(object):
    """nisqa.tar / nisqa_mos_only.tar (CNN-SA-AP) on one MI355X."""

    def __init__(self, args, state_dict, device=None, precision=None):
        """precision -- one of PRECISIONS, for every GEMM of the path (AdaptCNN / StandardCNN, self-attention, pooling):
        'bf16x6' (DEFAULT_PRECISION, the bench contract line): fp32 operands as three bf16 terms, an exact split, six MFMA products per
                 term pair -- the reference's operands bit for bit at 1.8 x the rate of 'f32';
        'f32'    exact fp32 MFMA, the reference's own arithmetic;
        'f16x4' / 'f16x3' (opt-in): the CNN on two f16 terms of the power-of-two-scaled tensors, four / three products (one fp32 ulp
                 off for ~25 % of the operands; as close to float64 as 'f32' by measurement); self-attention and pooling as in 'bf16x6';
        'bf16x3' (opt-in): two bf16 terms, three products: 16 of the 24 operand mantissa bits, |dMOS| <= 5e-5, the fast mode.
        The environment variable NISQA_HIP_PRECISION overrides the default."""
        if not torch.cuda.is_available():
            raise RuntimeError('nisqa_amd: no GPU visible (torch.cuda.is_available() is False); '
                               'the HIP engine has no CPU fallback')
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        self.args = args
        a = args
        sa = a.get('cnn_model') == 'adapt' and a.get('td') == 'self_att' and a.get('pool') == 'att'
        tts = a.get('cnn_model') == 'standard' and a.get('td') == 'lstm' and a.get('pool') == 'last_step_bi'
        if not (sa or tts) or a.get('td_2') not in (None, 'skip') or a.get('td_sa_pos_enc'):
            raise NotImplementedError(
                'HIP engine covers cnn_model=adapt / td=self_att / pool=att (nisqa.tar, nisqa_mos_only.tar) and '
                'cnn_model=standard / td=lstm / pool=last_step_bi (nisqa_tts.tar); got cnn_model={} td={} td_2={} pool={}'
                .format(a.get('cnn_model'), a.get('td'), a.get('td_2'), a.get('pool')))
        if a['ms_seg_length'] != SEG_LEN or a['ms_n_mels'] != 48:
            raise NotImplementedError('HIP engine is built for 15-frame segments of 48 mel bands')
        if sa and (list(a['cnn_pool_1']) != [24, 7] or list(a['cnn_pool_2']) != [12, 5] or list(a['cnn_pool_3']) != [6, 3]
                   or a['td_sa_nhead'] != 1 or a['td_sa_d_model'] != 64 or not a.get('pool_att_h')):
            raise NotImplementedError('HIP engine is built for the nisqa.tar geometry (pools 24x7/12x5/6x3, 1 head, '
                                      'd_model 64, pool_att_h)')
        if tts and (a.get('cnn_fc_out_h') != 20 or a.get('td_lstm_h') != 128 or a.get('td_lstm_num_layers') != 1
                    or not a.get('td_lstm_bidirectional') or a['model'] != 'NISQA'):
            raise NotImplementedError('HIP engine is built for the nisqa_tts.tar geometry (fc 20, BiLSTM 128 x 1 layer)')
        self.arch = 1 if tts else 0
        # ms_sr: lb.load(path, sr=ms_sr) resamples every file to that rate first (NISQA_lib.py:2300, 2304); None in every shipped checkpoint
        self.ms_sr = int(a['ms_sr']) if a.get('ms_sr') is not None else None
        self._resample_tables = {}
        self.seg_hop = int(a['ms_seg_hop_length'])
        self.max_segments = a['ms_max_segments']
        self.dim = a['model'] == 'NISQA_DIM'
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(self.device)
        self.precision = precision or os.environ.get('NISQA_HIP_PRECISION') or DEFAULT_PRECISION
        if self.precision not in PRECISIONS:
            raise ValueError('precision must be one of {}, got {}'.format(', '.join(PRECISIONS), self.precision))
        # the operand format of self-attention / pooling: the f16 CNN modes pair with the three-term kernels
        self.td_precision = 'bf16x6' if self.precision in ('f16x3', 'f16x4') else self.precision
        if self.arch == 1:
            # StandardCNN (split-bf16 or exact-fp32 MFMA) + BiLSTM + last-step pooling (fp32 VALU)
            self.n_layers, self.n_heads = 0, 1
            self.cnn_w = up(_w.pack_standard_cnn(state_dict))
            self.td_w = up(_w.pack_lstm_laststep(state_dict))
            self.pool_w = torch.zeros(4, dtype=torch.float32, device=self.device)
            self.cnn_wb = up(_w.pack_adapt_cnn_bf16(state_dict, conv1_pairs=True).view(np.int16)) if self.precision == 'bf16x3' else None
            if self.precision == 'bf16x6':               # three-term fragments (the BiLSTM and the pooling are fp32 in every mode)
                self.cnn_wb = up(_w.pack_adapt_cnn_bf16(state_dict, conv1_pairs=True, terms=3).view(np.int16))
            if self.precision in ('f16x3', 'f16x4'):     # two-term f16 fragments of the scaled weights + per-layer constants
                self.cnn_wb = up(_w.pack_adapt_cnn_f16(state_dict).view(np.int16))
            self.td_wb = self.pool_wb = None
            self._mel = {}
            self._ws = {}
            return
        self.n_layers = int(a['td_sa_num_layers'])
        heads = ['pool_layers.%d.model.' % h for h in range(5)] if self.dim else ['pool.model.']
        self.n_heads = len(heads)
        self.cnn_w = up(_w.pack_adapt_cnn(state_dict))
        bf = self.precision == 'bf16x3'
        self.cnn_wb = up(_w.pack_adapt_cnn_bf16(state_dict, conv1_pairs=True).view(np.int16)) if bf else None
        if self.precision == 'bf16x6':
            self.cnn_wb = up(_w.pack_adapt_cnn_bf16(state_dict, conv1_pairs=True, terms=3).view(np.int16))
        if self.precision in ('f16x3', 'f16x4'):
            self.cnn_wb = up(_w.pack_adapt_cnn_f16(state_dict).view(np.int16))
        self.td_w = up(_w.pack_self_att(state_dict, self.n_layers))
        self.pool_w = up(_w.pack_pool_att(state_dict, heads))
        self.td_wb = up(_w.pack_self_att_bf16(state_dict, self.n_layers).view(np.int16)) if bf else None
        self.pool_wb = up(_w.pack_pool_att_bf16(state_dict, heads).view(np.int16)) if bf else None
        if self.td_precision == 'bf16x6':                    # three-term fragments for self-attention and pooling as well
            self.td_wb = up(_w.pack_self_att_bf16(state_dict, self.n_layers, terms=3).view(np.int16))
            self.pool_wb = up(_w.pack_pool_att_bf16(state_dict, heads, terms=3).view(np.int16))
        self._mel = {}
        self._ws = {}                      # one workspace per stream (batches may be in flight on several)

    # -- tables -----------------------------------------------------------------------------
    def mel_tables(self, sr):
        sr = int(sr)
        if sr not in self._mel:
            a = self.args
            t = MelTables(sr, a['ms_n_fft'], a['ms_hop_length'], a['ms_win_length'], a['ms_n_mels'], a['ms_fmax'])
            up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(self.device)
            d = {'host': t, 'window': up(t.window), 'twiddle': up(t.twiddle), 'band_start': up(t.band_start),
                 'band_len': up(t.band_len), 'band_woff': up(t.band_woff), 'band_w': up(t.band_w)}
            d['cfg'] = _lib.MelCfg(t.n_fft, t.hop, t.win, t.n_mels, t.n_bins, int(t.band_w.size), 1e-8, 80.0)
            d['model'] = _lib.ModelDev(_ptr(d['window']), _ptr(d['twiddle']), _ptr(d['band_start']), _ptr(d['band_len']),
                                       _ptr(d['band_woff']), _ptr(d['band_w']), _ptr(self.cnn_w), _ptr(self.td_w),
                                       _ptr(self.pool_w), self.n_layers, self.n_heads, self.seg_hop, None,
                                       _ptr(self.cnn_wb) if self.cnn_wb is not None else None,
                                       CNN_MODE[self.precision],
                                       _ptr(self.td_wb) if self.td_wb is not None else None,
                                       _ptr(self.pool_wb) if self.pool_wb is not None else None, self.arch)
            self._mel[sr] = d
        return self._mel[sr]

    def plan(self, lengths, sr, names=None):
        return BatchPlan(lengths, self.mel_tables(sr)['host'].hop, self.seg_hop, self.max_segments, names)

    # -- ms_sr: what the batch looks like after lb.load(path, sr=ms_sr) -------------------------------------
    def rate(self, sr):
        """The rate the spectrogram is computed at for a file of rate ``sr``: ms_sr when the checkpoint sets it."""
        return int(sr) if self.ms_sr is None else self.ms_sr

    def audio_plan(self, lengths, sr, names=None):
        """plan() of a batch of files of rate ``sr`` as the network sees them (resampled to ms_sr when that is set)."""
        if self.ms_sr is None or int(sr) == self.ms_sr:
            return self.plan(lengths, sr, names)
        return self.plan(resampled_lengths(lengths, sr, self.ms_sr)[0], self.ms_sr, names)

    def resample(self, pcm, lengths, sr):
        """pcm: device tensor, the clips of ``lengths`` samples back to back at rate ``sr`` (float32 or int16 PCM) -> float32 device
        tensor, the clips at ms_sr back to back (librosa.resample(..., res_type='kaiser_best') + fix_length: nisqa_resample)."""
        sr = int(sr)
        if self.ms_sr is None or sr == self.ms_sr:
            return pcm
        ratio = float(self.ms_sr) / float(sr)
        if sr not in self._resample_tables:
            # built once per source rate, on the default stream, and complete before any stream of the predict loop reads it (the
            # loop alternates two kernel streams: a table uploaded on one of them would be unordered against the other)
            ds = torch.cuda.default_stream(self.device)
            with torch.cuda.stream(ds):
                t = torch.from_numpy(kaiser_best_table(ratio)).to(self.device)
            ds.synchronize()
            self._resample_tables[sr] = t
        table = self._resample_tables[sr]
        lengths = np.asarray(lengths, dtype=np.int64).reshape(-1)
        n_out, valid = resampled_lengths(lengths, sr, self.ms_sr)
        offs = np.concatenate([np.concatenate([[0], np.cumsum(lengths)]), np.concatenate([[0], np.cumsum(n_out)]), valid]).astype(np.int64)
        assert pcm.numel() == int(lengths.sum()) and pcm.dtype in (torch.float32, torch.int16)
        host = torch.empty(offs.shape, dtype=torch.int64, pin_memory=True)       # page-locked + non-blocking: a pageable copy would block the host
        host.numpy()[...] = offs                                                   # until the stream has drained (caching host allocator: reuse is stream-safe)
        dev = host.to(self.device, non_blocking=True)
        b = len(lengths)
        out = torch.empty(int(n_out.sum()), dtype=torch.float32, device=self.device)
        ws = torch.empty(max(8, self.lib.nisqa_resample_workspace_bytes(b, int(n_out.max()))), dtype=torch.uint8, device=self.device)
        rc = self.lib.nisqa_resample(_ptr(pcm), 1 if pcm.dtype == torch.int16 else 0, _ptr(dev[:b + 1]), _ptr(dev[b + 1:2 * b + 2]),
                                     _ptr(dev[2 * b + 2:]), b, int(n_out.max()), ratio, _ptr(table), table.shape[0],
                                     1 << KAISER_BEST_PRECISION, _ptr(ws), ws.numel(), _ptr(out), self._stream())
        _lib.check(rc, 'nisqa_resample')
        return out

    def forward_audio(self, pcm, lengths, sr, plan, stage_events=None):
        """forward_pcm for a batch as the files hold it: resampled to ms_sr first when the checkpoint asks for that (``plan`` from
        audio_plan(lengths, sr))."""
        return self.forward_pcm(self.resample(pcm, lengths, sr), plan, self.rate(sr), stage_events)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- whole forward --------------------------------------------------------------------------
    def forward_pcm(self, pcm, plan, sr, stage_events=None):
        """pcm: device tensor [plan.total_samples], float32 samples or int16 PCM (scaled by 1/32768 inside the mel
        kernel, as soundfile does for lb.load) -> device tensor [B, n_heads].

        stage_events: optional list of 6 recorded-once torch.cuda.Event(enable_timing=True); they are
        re-recorded at the stage boundaries (profiling hook of nisqa_model_dev)."""
        assert pcm.dtype in (torch.float32, torch.int16) and pcm.is_cuda and pcm.numel() == plan.total_samples
        mt = self.mel_tables(sr)
        d = plan.to(self.device)
        need = self.lib.nisqa_workspace_bytes(plan.n_clips, plan.total_frames, plan.total_tok)
        skey = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(skey)
        if ws is None or ws.numel() < need:
            ws = self._ws[skey] = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=self.device)
        out = torch.empty((plan.n_clips, self.n_heads), dtype=torch.float32, device=self.device)
        model = _lib.ModelDev.from_buffer_copy(mt['model'])
        if stage_events is not None:
            arr = (ctypes.c_void_p * 6)(*[ctypes.c_void_p(e.cuda_event if e is not None else None) for e in stage_events])
            model.stage_events = ctypes.cast(arr, ctypes.c_void_p)
        entry = self.lib.nisqa_predict_batch_pcm16 if pcm.dtype == torch.int16 else self.lib.nisqa_predict_batch
        rc = entry(_ptr(pcm), _ptr(d['clip_off']), _ptr(d['frame_off']), _ptr(d['tok_off']),
                   _ptr(d['n_wins']), plan.n_clips, plan.total_frames, plan.total_tok,
                   ctypes.byref(mt['cfg']), ctypes.byref(model), _ptr(ws),
                   ws.numel(), _ptr(out), self._stream())
        _lib.check(rc, 'nisqa_predict_batch')
        return out

    def pcm16_to_f32(self, pcm16):
        out = torch.empty(pcm16.numel(), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.nisqa_pcm16_to_f32(_ptr(pcm16), _ptr(out), pcm16.numel(), self._stream()), 'nisqa_pcm16_to_f32')
        return out

    # -- stage-by-stage API (parity tests, profiling) ---------------------------------------------
    def mel(self, pcm, plan, sr, clamp=True):
        """-> (mel_tm [TT,48] dB, clip_floor [B]); with clamp the top_db floor is applied in place."""
        mt = self.mel_tables(sr)
        d = plan.to(self.device)
        mel = torch.empty((plan.total_frames, 48), dtype=torch.float32, device=self.device)
        cmax = torch.zeros(plan.n_clips, dtype=torch.int32, device=self.device)
        floor = torch.empty(plan.n_clips, dtype=torch.float32, device=self.device)
        entry = self.lib.nisqa_mel_db_pcm16 if pcm.dtype == torch.int16 else self.lib.nisqa_mel_db
        _lib.check(entry(_ptr(pcm), _ptr(d['clip_off']), _ptr(d['frame_off']), plan.n_clips,
                         plan.total_frames, ctypes.byref(mt['cfg']), _ptr(mt['window']),
                         _ptr(mt['twiddle']), _ptr(mt['band_start']), _ptr(mt['band_len']),
                         _ptr(mt['band_woff']), _ptr(mt['band_w']), _ptr(mel), _ptr(cmax),
                         self._stream()), 'nisqa_mel_db')
        _lib.check(self.lib.nisqa_mel_finalize(_ptr(mel), _ptr(d['frame_off']), plan.n_clips, plan.total_frames,
                                               _ptr(cmax), 80.0, _ptr(floor), 1 if clamp else 0, self._stream()),
                   'nisqa_mel_finalize')
        return mel, floor

    def cnn(self, mel_tm, clip_floor, plan):
        d = plan.to(self.device)
        p3 = torch.empty((plan.total_tok, 18, 64), dtype=torch.float32, device=self.device)
        feat = torch.zeros((plan.total_tok, 384), dtype=torch.float32, device=self.device)
        if self.precision == 'bf16x3':
            _lib.check(self.lib.nisqa_cnn_adapt_bf16(_ptr(mel_tm), _ptr(d['frame_off']), _ptr(d['tok_off']),
                                                     _ptr(d['n_wins']), _ptr(clip_floor), plan.n_clips, plan.total_tok,
                                                     self.seg_hop, _ptr(self.cnn_w), _ptr(self.cnn_wb), _ptr(p3),
                                                     _ptr(feat), self._stream()), 'nisqa_cnn_adapt_bf16')
        elif self.precision == 'bf16x6':
            _lib.check(self.lib.nisqa_cnn_adapt_bf16x6(_ptr(mel_tm), _ptr(d['frame_off']), _ptr(d['tok_off']),
                                                       _ptr(d['n_wins']), _ptr(clip_floor), plan.n_clips, plan.total_tok,
                                                       self.seg_hop, _ptr(self.cnn_w), _ptr(self.cnn_wb), _ptr(feat),
                                                       self._stream()), 'nisqa_cnn_adapt_bf16x6')
            p3 = None                      # (the one-launch kernel has no pooled conv4 tensor in memory)
        elif self.precision in ('f16x3', 'f16x4'):
            _lib.check(self.lib.nisqa_cnn_adapt_f16(_ptr(mel_tm), _ptr(d['frame_off']), _ptr(d['tok_off']), _ptr(d['n_wins']),
                                                    _ptr(clip_floor), plan.n_clips, plan.total_tok, self.seg_hop, _ptr(self.cnn_w),
                                                    _ptr(self.cnn_wb), int(self.precision[-1]), _ptr(feat), self._stream()),
                       'nisqa_cnn_adapt_f16')
            p3 = None
        else:
            _lib.check(self.lib.nisqa_cnn_adapt(_ptr(mel_tm), _ptr(d['frame_off']), _ptr(d['tok_off']), _ptr(d['n_wins']),
                                                _ptr(clip_floor), plan.n_clips, plan.total_tok, self.seg_hop,
                                                _ptr(self.cnn_w), _ptr(p3), _ptr(feat), self._stream()), 'nisqa_cnn_adapt')
        return feat, p3

    def forward_segments(self, x, n_wins):
        """Reference inner operator model(x[B,L,1,48,15], n_wins[B]) -> [B, heads] (NL:137-142, NL:260-268)."""
        if self.arch != 0:
            raise NotImplementedError('segment-tensor forward is implemented for the CNN-SA-AP architecture only')
        if x.dim() != 5 or tuple(x.shape[2:]) != (1, 48, SEG_LEN):
            raise ValueError('expected x of shape [B, L, 1, 48, 15], got {}'.format(tuple(x.shape)))
        x = x.to(self.device, dtype=torch.float32).contiguous()
        n = np.asarray(n_wins.detach().cpu().numpy() if torch.is_tensor(n_wins) else n_wins, dtype=np.int64).reshape(-1)
        B, L = x.shape[0], x.shape[1]
        if len(n) != B or (n < 1).any() or (n > L).any():
            raise ValueError('n_wins must hold one count in [1, L] per clip')
        plan = BatchPlan.from_n_wins(n)
        d = plan.to(self.device)
        p3 = torch.empty((plan.total_tok, 18, 64), dtype=torch.float32, device=self.device)
        feat = torch.empty((plan.total_tok, 384), dtype=torch.float32, device=self.device)
        if self.precision in ('bf16x3', 'bf16x6'):
            fn = self.lib.nisqa_cnn_adapt_segments_bf16 if self.precision == 'bf16x3' else self.lib.nisqa_cnn_adapt_segments_bf16x6
            _lib.check(fn(_ptr(x), L, _ptr(d['tok_off']), _ptr(d['n_wins']), B, plan.total_tok, _ptr(self.cnn_w), _ptr(self.cnn_wb),
                          _ptr(feat), self._stream()), 'nisqa_cnn_adapt_segments_' + self.precision)
        elif self.precision in ('f16x3', 'f16x4'):
            _lib.check(self.lib.nisqa_cnn_adapt_segments_f16(_ptr(x), L, _ptr(d['tok_off']), _ptr(d['n_wins']), B, plan.total_tok,
                                                             _ptr(self.cnn_w), _ptr(self.cnn_wb), int(self.precision[-1]), _ptr(feat),
                                                             self._stream()), 'nisqa_cnn_adapt_segments_f16')
        else:
            _lib.check(self.lib.nisqa_cnn_adapt_segments(_ptr(x), L, _ptr(d['tok_off']), _ptr(d['n_wins']), B,
                                                         plan.total_tok, _ptr(self.cnn_w), _ptr(p3), _ptr(feat),
                                                         self._stream()), 'nisqa_cnn_adapt_segments')
        return self.td_pool(feat, plan)

    # -- nisqa_tts.tar stages ------------------------------------------------------------------------
    def cnn_std(self, mel_tm, clip_floor, plan):
        """StandardCNN + fc_out -> feat20 [NP, 20]"""
        d = plan.to(self.device)
        feat = torch.zeros((plan.total_tok, 20), dtype=torch.float32, device=self.device)
        if self.precision in ('f16x3', 'f16x4'):
            _lib.check(self.lib.nisqa_cnn_standard_f16(_ptr(mel_tm), _ptr(d['frame_off']), _ptr(d['tok_off']), _ptr(d['n_wins']),
                                                       _ptr(clip_floor), plan.n_clips, plan.total_tok, self.seg_hop, _ptr(self.cnn_w),
                                                       _ptr(self.cnn_wb), int(self.precision[-1]), _ptr(feat), self._stream()),
                       'nisqa_cnn_standard_f16')
            return feat
        if self.precision in ('bf16x3', 'bf16x6'):
            fn = self.lib.nisqa_cnn_standard_bf16 if self.precision == 'bf16x3' else self.lib.nisqa_cnn_standard_bf16x6
            _lib.check(fn(_ptr(mel_tm), _ptr(d['frame_off']), _ptr(d['tok_off']), _ptr(d['n_wins']), _ptr(clip_floor), plan.n_clips,
                          plan.total_tok, self.seg_hop, _ptr(self.cnn_w), _ptr(self.cnn_wb), _ptr(feat), self._stream()),
                       'nisqa_cnn_standard_' + self.precision)
            return feat
        p3 = torch.empty((plan.total_tok, 12, 64), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.nisqa_cnn_standard(_ptr(mel_tm), _ptr(d['frame_off']), _ptr(d['tok_off']), _ptr(d['n_wins']),
                                               _ptr(clip_floor), plan.n_clips, plan.total_tok, self.seg_hop,
                                               _ptr(self.cnn_w), _ptr(p3), _ptr(feat), self._stream()), 'nisqa_cnn_standard')
        return feat

    def lstm(self, feat20, plan, want_seq=False):
        """BiLSTM + PoolLastStepBi -> (out [B,1], seq [NP,256] or None)"""
        d = plan.to(self.device)
        hfin = torch.empty((plan.n_clips, 256), dtype=torch.float32, device=self.device)
        seq = torch.zeros((plan.total_tok, 256), dtype=torch.float32, device=self.device) if want_seq else None
        out = torch.empty((plan.n_clips, 1), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.nisqa_lstm_laststep(_ptr(feat20), _ptr(d['tok_off']), _ptr(d['n_wins']), plan.n_clips,
                                                _ptr(self.td_w), _ptr(hfin), _ptr(seq) if want_seq else None, _ptr(out),
                                                self._stream()), 'nisqa_lstm_laststep')
        return out, seq

    def td(self, feat, plan):
        d = plan.to(self.device)
        ws = torch.empty(plan.total_tok * 64 * 9, dtype=torch.float32, device=self.device)
        x = torch.zeros((plan.total_tok, 64), dtype=torch.float32, device=self.device)
        if self.td_precision in ('bf16x3', 'bf16x6'):
            fn = self.lib.nisqa_td_selfatt_bf16 if self.td_precision == 'bf16x3' else self.lib.nisqa_td_selfatt_bf16x6
            _lib.check(fn(_ptr(feat), _ptr(d['tok_off']), _ptr(d['n_wins']), plan.n_clips, plan.total_tok, self.n_layers,
                          _ptr(self.td_w), _ptr(self.td_wb), _ptr(ws), _ptr(x), self._stream()), 'nisqa_td_selfatt_' + self.td_precision)
        else:
            _lib.check(self.lib.nisqa_td_selfatt(_ptr(feat), _ptr(d['tok_off']), _ptr(d['n_wins']), plan.n_clips,
                                                 plan.total_tok, self.n_layers, _ptr(self.td_w), _ptr(ws), _ptr(x),
                                                 self._stream()), 'nisqa_td_selfatt')
        return x

    def td_pool(self, feat, plan):
        """self-attention + attention pooling as the whole-batch forward runs them: in the three-term modes ONE chain of
        n_layers + 1 launches (nisqa_td_pool_bf16x6), otherwise td() then pool()"""
        if self.td_precision != 'bf16x6':
            return self.pool(self.td(feat, plan), plan)
        d = plan.to(self.device)
        ws = torch.empty(plan.total_tok * 64 * 9, dtype=torch.float32, device=self.device)
        wsp = torch.empty(plan.total_tok * 16 + plan.n_clips, dtype=torch.float32, device=self.device)
        x = torch.zeros((plan.total_tok, 64), dtype=torch.float32, device=self.device)
        out = torch.empty((plan.n_clips, self.n_heads), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.nisqa_td_pool_bf16x6(_ptr(feat), _ptr(d['tok_off']), _ptr(d['n_wins']), plan.n_clips, plan.total_tok,
                                                 self.n_layers, _ptr(self.td_w), _ptr(self.td_wb), self.n_heads, _ptr(self.pool_wb),
                                                 _ptr(ws), _ptr(x), _ptr(wsp), _ptr(out), self._stream()), 'nisqa_td_pool_bf16x6')
        return out

    def pool(self, x, plan):
        d = plan.to(self.device)
        ws = torch.empty(plan.total_tok * 16, dtype=torch.float32, device=self.device)
        out = torch.empty((plan.n_clips, self.n_heads), dtype=torch.float32, device=self.device)
        if self.td_precision in ('bf16x3', 'bf16x6'):
            fn = self.lib.nisqa_pool_att_bf16 if self.td_precision == 'bf16x3' else self.lib.nisqa_pool_att_bf16x6
            _lib.check(fn(_ptr(x), _ptr(d['tok_off']), _ptr(d['n_wins']), plan.n_clips, plan.total_tok, self.n_heads,
                          _ptr(self.pool_w), _ptr(self.pool_wb), _ptr(ws), _ptr(out), self._stream()), 'nisqa_pool_att_' + self.td_precision)
        else:
            _lib.check(self.lib.nisqa_pool_att(_ptr(x), _ptr(d['tok_off']), _ptr(d['n_wins']), plan.n_clips, plan.total_tok,
                                               self.n_heads, _ptr(self.pool_w), _ptr(ws), _ptr(out), self._stream()),
                       'nisqa_pool_att')
        return out

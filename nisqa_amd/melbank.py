"""Host-side constant tables of the mel front end (product code; numpy, float64 then float32).

Mirrors what librosa 0.8.1 builds inside ``lb.feature.melspectrogram`` for the reference's call
(nisqa/NISQA_lib.py:2311-2328): periodic hann window, slaney-normalised triangular mel filterbank
(htk=False) -- here exported in sparse row form because each FFT bin touches at most two bands --
plus the 4096-point twiddle table the FFT kernel indexes.
"""
import os

import numpy as np


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        big = f >= min_log_hz
        mels[big] = min_log_mel + np.log(f[big] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = m >= min_log_mel
    freqs[big] = min_log_hz * np.exp(logstep * (m[big] - min_log_mel))
    return freqs


def slaney_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Dense [n_mels, 1+n_fft//2] float32 filterbank (librosa.filters.mel, htk=False, norm='slaney')."""
    n_bins = 1 + n_fft // 2
    fb = np.zeros((n_mels, n_bins), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = np.subtract.outer(edges, fftfreqs)
    for m in range(n_mels):
        fb[m] = np.maximum(0, np.minimum(-ramps[m] / width[m], ramps[m + 2] / width[m + 1]))
    fb *= (2.0 / (edges[2:n_mels + 2] - edges[:n_mels]))[:, np.newaxis]
    return fb


class MelTables(object):
    """All constant arrays nisqa_mel_db needs, as numpy (upload once per model)."""

    def __init__(self, sr, n_fft, hop_s, win_s, n_mels, fmax):
        if int(n_fft) != 4096:
            raise NotImplementedError('HIP mel front end supports ms_n_fft=4096 only (got {})'.format(n_fft))
        if int(n_mels) != 48:
            raise NotImplementedError('HIP mel front end supports ms_n_mels=48 only (got {})'.format(n_mels))
        self.sr = int(sr)
        self.n_fft = int(n_fft)
        self.hop = int(sr * hop_s)                 # NISQA_lib.py:2308
        self.win = int(sr * win_s)                 # NISQA_lib.py:2309
        if not (2 <= self.win <= self.n_fft):
            raise NotImplementedError(
                'HIP mel front end needs 2 <= win_length <= n_fft samples (sr {} gives {})'.format(sr, self.win))
        self.n_mels = int(n_mels)
        n = np.arange(self.win, dtype=np.float64)
        self.window = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / self.win)).astype(np.float32)
        k = np.arange(4096, dtype=np.float64)
        self.twiddle = np.stack([np.cos(2 * np.pi * k / 4096), -np.sin(2 * np.pi * k / 4096)], 1).astype(np.float32)
        fb = slaney_filterbank(self.sr, self.n_fft, self.n_mels, 0.0, float(fmax))
        # Sparse rows for the kernel: band m covers bins [start[m], start[m] + len[m]) with weights
        # band_w[woff[m] : woff[m] + len[m]].  The kernel sums four consecutive bands per pass (one per 16-lane
        # row) with a wave-uniform trip count, so len is PADDED (zero weights) to the pass maximum rounded up
        # to 16 and is identical for the four bands of a pass.
        start = np.zeros(self.n_mels, np.int32)
        true_len = np.zeros(self.n_mels, np.int32)
        for m in range(self.n_mels):
            nz = np.nonzero(fb[m])[0]
            if len(nz):
                start[m], true_len[m] = nz[0], nz[-1] - nz[0] + 1
        length = np.zeros(self.n_mels, np.int32)
        for ps in range(self.n_mels // 4):
            length[4 * ps:4 * ps + 4] = max(16, -(-int(true_len[4 * ps:4 * ps + 4].max()) // 16) * 16)
        # The four bands of a pass are read by the four 16-lane rows of a wave in the same instruction: their weight
        # runs start 16 floats apart modulo the 64 LDS banks (a pass length that is a multiple of 64 would put all four
        # rows on the same 16 banks), at the price of <= 48 floats of padding per band.
        woff = np.zeros(self.n_mels, np.int32)
        pos = 0
        for m in range(self.n_mels):
            if os.environ.get('NISQA_MEL_BANK_PAD', '1') != '0':
                pos += (16 * (m % 4) - pos) % 64
            woff[m] = pos
            pos += int(length[m])
        w = np.zeros(pos, np.float32)
        for m in range(self.n_mels):
            w[woff[m]:woff[m] + true_len[m]] = fb[m, start[m]:start[m] + true_len[m]]
        self.band_start, self.band_len, self.band_woff, self.band_w = start, length, woff, w
        self.true_len = true_len
        self.n_bins = int((start + true_len).max()) if true_len.any() else 1
        self.dense = fb


# ---- lb.load(path, sr=ms_sr): the table of librosa's default resampler (res_type='kaiser_best' -> resampy) ----------------------
KAISER_BEST_ZEROS, KAISER_BEST_PRECISION = 64, 9
KAISER_BEST_ROLLOFF, KAISER_BEST_BETA = 0.9475937167399596, 14.769656459379492


def kaiser_best_table(ratio):
    """float32 [64 * 512 + 1, 2] for csrc/resample.hip (nisqa_resample): column 0 the 'kaiser_best' half window of resampy --
    rolloff * sinc(rolloff * t) under the right half of a Kaiser window (beta 14.77), 64 zero crossings at 512 entries each,
    multiplied by ``ratio`` when downsampling -- column 1 its forward differences (the linear interpolation between
    entries).  Parameters by recollection of resampy 0.2.2 (a floating dependency of librosa 0.8.1, reference env.yml:16)."""
    n = (1 << KAISER_BEST_PRECISION) * KAISER_BEST_ZEROS
    win = KAISER_BEST_ROLLOFF * np.sinc(KAISER_BEST_ROLLOFF * np.linspace(0, KAISER_BEST_ZEROS, num=n + 1, endpoint=True))
    win = win * np.kaiser(2 * n + 1, KAISER_BEST_BETA)[n:]
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    return np.ascontiguousarray(np.stack([win, delta], axis=1), dtype=np.float32)


def resampled_lengths(lengths, sr, target_sr):
    """(out, valid): samples per clip after lb.load(..., sr=target_sr) -- librosa fixes the length to ceil(n * ratio), resampy
    computes int(n * ratio) of them (both in float64 like the originals)."""
    n = np.asarray(lengths, dtype=np.int64)
    ratio = float(target_sr) / float(sr)
    prod = n.astype(np.float64) * ratio
    return np.ceil(prod).astype(np.int64), prod.astype(np.int64)

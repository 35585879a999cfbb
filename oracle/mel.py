"""Oracle (test infrastructure, NOT product code): CPU restatement of the
reference mel front end, ``get_librosa_melspec`` (NISQA_lib.py:2284-2331).

**parity unpinned**: the arithmetic the reference runs here lives in
librosa==0.8.1 / soundfile (env.yml:8,16), a third-party dependency that is
not vendored under /root/reference and is not installable in the build
container.  Every function below restates the published librosa 0.8.1
algorithm for the exact call the reference makes and cites both the reference
call site and the librosa routine it follows.  The reference ships no golden
vector for this stage, so nothing pins these numbers
against the real library.  What narrows the gap: tests/test_oracle.py
cross-checks this file against transformers.audio_utils (an independently
written slaney filter bank + centred reflect-padded STFT + amplitude_to_db
that upstream tests against librosa): filter bank equal to 1e-9, dB
spectrogram equal to 1.2e-5 dB at 48 kHz and 16 kHz.  That is agreement of
two restatements, not a run of librosa 0.8.1 -- the stage stays "unpinned".

Dtype discipline follows librosa 0.8.1: audio is float32; the STFT is taken in
float64 (numpy.fft upcasts) and stored as complex64; magnitude, mel projection
and dB are float32.
"""
import numpy as np

# --------------------------------------------------------------------------
# lb.load  (NISQA_lib.py:2299-2304)
# --------------------------------------------------------------------------

def load_wav(path, ms_channel=None):
    """librosa.load(path, sr=None[, mono=False]) as called at NISQA_lib.py:2300/2304.

    soundfile semantics: integer PCM is scaled by 1/2**(bits-1) to float32,
    float WAVs pass through, the result is (channels, n).  ``mono=True`` (the
    default branch, NISQA_lib.py:2304) averages channels (librosa.to_mono);
    with ``ms_channel`` the reference loads mono=False and picks one row
    (NISQA_lib.py:2300-2302).  ``ms_sr`` is None in every shipped checkpoint so
    no resampling happens; the native rate is returned.

    The decoder here is scipy.io.wavfile (independent of the product's own
    RIFF parser, so the two cross-check each other).
    """
    from scipy.io import wavfile
    try:
        sr, data = wavfile.read(path)
    except Exception:
        raise ValueError('Could not load file {}'.format(path))
    if data.dtype == np.int16:
        y = data.astype(np.float32) / np.float32(32768.0)
    elif data.dtype == np.int32:
        # scipy left-justifies 24-bit PCM into int32, so 1/2**31 is right for both
        y = (data.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif data.dtype == np.uint8:
        y = (data.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif data.dtype in (np.float32, np.float64):
        y = data.astype(np.float32)
    else:
        raise ValueError('Could not load file {}'.format(path))
    if y.ndim == 2:
        y = y.T                              # (channels, n) like soundfile(...).T
        if ms_channel is not None:
            y = y[ms_channel, :]
        else:
            y = np.mean(y, axis=0, dtype=np.float32)
    return np.ascontiguousarray(y), int(sr)


# --------------------------------------------------------------------------
# librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney')
# (called through lb.feature.melspectrogram, NISQA_lib.py:2311-2328)
# --------------------------------------------------------------------------

def _hz_to_mel(freq):
    """librosa.core.convert.hz_to_mel, htk=False (Slaney auditory-toolbox scale)."""
    freq = np.asanyarray(freq, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = freq / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if freq.ndim:
        log_t = freq >= min_log_hz
        mels[log_t] = min_log_mel + np.log(freq[log_t] / min_log_hz) / logstep
    elif freq >= min_log_hz:
        mels = min_log_mel + np.log(freq / min_log_hz) / logstep
    return mels


def _mel_to_hz(mels):
    """librosa.core.convert.mel_to_hz, htk=False."""
    mels = np.asanyarray(mels, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    log_t = mels >= min_log_mel
    freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    return freqs


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(..., htk=False, norm='slaney', dtype=float32) -> [n_mels, 1+n_fft//2]."""
    n_bins = 1 + n_fft // 2
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


# --------------------------------------------------------------------------
# librosa.stft / melspectrogram / amplitude_to_db
# --------------------------------------------------------------------------

def hann_periodic(win_length):
    """scipy.signal.get_window('hann', M, fftbins=True) (float64)."""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def stft_mag(y, n_fft, hop_length, win_length):
    """|librosa.stft(y, n_fft, hop, win, window='hann', center=True, pad_mode='reflect')|.

    librosa 0.8.1: window centre-padded to n_fft, y reflect-padded by n_fft//2,
    frames = 1 + len(y)//hop, FFT in float64, stored complex64, abs -> float32.
    """
    y = np.asarray(y, dtype=np.float32)
    win = hann_periodic(win_length)
    lpad = (n_fft - win_length) // 2
    fft_window = np.zeros(n_fft, dtype=np.float64)
    fft_window[lpad:lpad + win_length] = win
    ypad = np.pad(y, n_fft // 2, mode='reflect')
    n_frames = 1 + (len(ypad) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    out = np.empty((1 + n_fft // 2, n_frames), dtype=np.complex64)
    blk = 256
    for s in range(0, n_frames, blk):
        fr = ypad[idx[:, s:s + blk]].astype(np.float64) * fft_window[:, None]
        out[:, s:s + blk] = np.fft.rfft(fr, axis=0)
    return np.abs(out)            # float32, power=1.0 (NISQA_lib.py:2321)


def amplitude_to_db(S, amin=1e-4, top_db=80.0):
    """librosa.amplitude_to_db(S, ref=1.0, amin=1e-4, top_db=80) (NISQA_lib.py:2330).

    = power_to_db(S**2, ref=1, amin=amin**2): 10*log10(max(amin^2, S^2)) - 0,
    then floor at (global max - top_db).  float32 throughout.
    """
    S = np.asarray(S, dtype=np.float32)
    power = np.square(np.abs(S))
    log_spec = np.float32(10.0) * np.log10(np.maximum(np.float32(amin * amin), power))
    log_spec = log_spec - np.float32(10.0 * np.log10(max(amin * amin, 1.0)))
    return np.maximum(log_spec, log_spec.max() - np.float32(top_db)).astype(np.float32)


def melspec_db_from_audio(y, sr, n_fft=4096, hop_length=0.01, win_length=0.02,
                          n_mels=48, fmax=20000.0, return_unclamped=False):
    """get_librosa_melspec after lb.load (NISQA_lib.py:2308-2331) -> [n_mels, T] float32 dB."""
    hop = int(sr * hop_length)          # NISQA_lib.py:2308
    win = int(sr * win_length)          # NISQA_lib.py:2309
    S = stft_mag(y, n_fft, hop, win)
    fb = mel_filterbank(sr, n_fft, n_mels, 0.0, fmax)
    M = np.dot(fb, S).astype(np.float32)
    if return_unclamped:
        return amplitude_to_db(M, top_db=1e30)
    return amplitude_to_db(M)


# --------------------------------------------------------------------------------------------------------------
# librosa.load(path, sr=ms_sr) -> librosa.resample(y, sr_native, ms_sr, res_type='kaiser_best') -> resampy.resample
# (NISQA_lib.py:2300, 2304 pass sr=ms_sr; None in every shipped checkpoint).  resampy is a dependency of librosa 0.8.1
# (>= 0.2.2, not pinned in env.yml, not in /root/reference): its published algorithm -- Smith's band-limited
# interpolation with a precomputed half window and linear interpolation between table entries -- is restated here BY
# RECOLLECTION of resampy 0.2.2 (resampy/filters.py sinc_window + the 'kaiser_best' parameters; resampy/interpn.py
# resample_f).  PARITY UNPINNED; tests/test_oracle.py checks it as a resampler (band-limited signals against their
# analytic values at the new rate), which holds whatever the exact table constants.
# --------------------------------------------------------------------------------------------------------------
KAISER_BEST = {'num_zeros': 64, 'precision': 9, 'rolloff': 0.9475937167399596, 'beta': 14.769656459379492}


def kaiser_best_half_window():
    """resampy.filters.sinc_window(num_zeros=64, precision=9, window=kaiser(beta), rolloff): float64 [64 * 512 + 1]."""
    from scipy.signal.windows import kaiser
    p = KAISER_BEST
    n = (2 ** p['precision']) * p['num_zeros']
    sinc_win = p['rolloff'] * np.sinc(p['rolloff'] * np.linspace(0, p['num_zeros'], num=n + 1, endpoint=True))
    taper = kaiser(2 * n + 1, p['beta'])[n:]
    return taper * sinc_win


def resample_kaiser_best(y, sr_orig, sr_new):
    """librosa.resample(y, sr_orig, sr_new) with its defaults (res_type='kaiser_best', fix=True, scale=False) for a mono
    float32 signal: resampy's loop (one output sample = left wing + right wing of the interpolated window, accumulated in
    the float32 output element one tap at a time), then util.fix_length to ceil(len * ratio)."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    if sr_orig == sr_new:
        return y
    ratio = float(sr_new) / sr_orig
    n_orig = len(y)
    n_out = int(n_orig * ratio)
    win = kaiser_best_half_window()
    num_table = 2 ** KAISER_BEST['precision']
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    nwin = len(win)
    # time_register += time_increment, one addition per output sample (cumsum adds sequentially)
    time = np.concatenate(([0.0], np.cumsum(np.full(max(0, n_out - 1), 1.0 / ratio))))[:n_out]
    n = time.astype(np.int64)
    out = np.zeros(n_out, dtype=np.float32)
    xp = np.concatenate((y.astype(np.float64), [0.0]))               # (index n_orig: masked taps read a zero)
    for wing in (0, 1):
        frac = scale * (time - n) if wing == 0 else scale - scale * (time - n)
        index_frac = frac * num_table
        offset = index_frac.astype(np.int64)
        eta = index_frac - offset
        limit = (n + 1) if wing == 0 else (n_orig - n - 1)
        taps = np.minimum(limit, (nwin - offset) // index_step)
        for i in range(int(taps.max()) if n_out else 0):
            live = i < taps
            idx = np.where(live, offset + i * index_step, 0)
            weight = win[idx] + eta * delta[idx]
            src = np.where(live, (n - i) if wing == 0 else (n + i + 1), n_orig)
            out = np.where(live, (out.astype(np.float64) + weight * xp[src]).astype(np.float32), out)
    want = int(np.ceil(n_orig * ratio))                              # librosa.util.fix_length
    if want > n_out:
        out = np.concatenate((out, np.zeros(want - n_out, dtype=np.float32)))
    return np.ascontiguousarray(out[:want])


def get_melspec(path, sr=None, n_fft=4096, hop_length=0.01, win_length=0.02,
                n_mels=48, fmax=20000.0, ms_channel=None):
    """Whole of get_librosa_melspec (NISQA_lib.py:2284-2331); sr = ms_sr (None: the file's rate, as in every shipped
    checkpoint; a number: lb.load resamples to it first)."""
    y, sr_file = load_wav(path, ms_channel)
    if sr is not None and int(sr) != sr_file:
        y, sr_file = resample_kaiser_best(y, sr_file, int(sr)), int(sr)
    return melspec_db_from_audio(y, sr_file, n_fft, hop_length, win_length, n_mels, fmax)

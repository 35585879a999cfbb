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


def write_wav(path, pcm, sr=48000):
    """Minimal PCM WAV writer (mono [n] or multi-channel [n, ch], int16/int32/uint8/float32)."""
    import struct
    pcm = np.ascontiguousarray(pcm)
    ch = 1 if pcm.ndim == 1 else pcm.shape[1]
    if pcm.dtype == np.float32:
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

"""WAV ingest for the predict path: what ``lb.load(path, sr=None[, mono=False])`` does at
reference nisqa/NISQA_lib.py:2299-2304, for RIFF/WAVE files, without librosa/soundfile.

soundfile semantics: integer PCM -> float32 scaled by 1/2**(bits-1) (8-bit is unsigned, offset
128); float WAVs pass through; multi-channel audio is averaged (librosa.to_mono) unless
``ms_channel`` selects one channel.  The native sample rate is returned (ms_sr=None).
Mono PCM16 -- the common case -- is returned as int16 so that only 2 bytes/sample cross PCIe; the
1/32768 scaling then happens on the GPU (nisqa_pcm16_to_f32), bit-identical to the host scaling.
"""
import struct

import numpy as np

_PCM, _FLOAT, _EXT = 1, 3, 0xFFFE


def _parse(path):
    with open(path, 'rb') as f:
        raw = f.read()
    if len(raw) < 12 or raw[0:4] not in (b'RIFF', b'RF64') or raw[8:12] != b'WAVE':
        raise ValueError('not a RIFF/WAVE file')
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, size = raw[pos:pos + 4], struct.unpack('<I', raw[pos + 4:pos + 8])[0]
        body = pos + 8
        if cid == b'fmt ':
            tag, ch, sr, _, blk, bits = struct.unpack('<HHIIHH', raw[body:body + 16])
            if tag == _EXT and size >= 26:
                tag = struct.unpack('<H', raw[body + 24:body + 26])[0]
            fmt = (tag, ch, sr, blk, bits)
        elif cid == b'data':
            if size == 0xFFFFFFFF or body + size > len(raw):
                size = len(raw) - body
            data = raw[body:body + size]
            break
        pos = body + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError('missing fmt/data chunk')
    return fmt, data


def read_wav(path, ms_channel=None):
    """-> (samples, sr); samples is int16 [n] (mono PCM16) or float32 [n]."""
    try:
        (tag, ch, sr, blk, bits), data = _parse(path)
        if ch < 1 or blk != ch * ((bits + 7) // 8):
            raise ValueError('bad block align')
        n = len(data) // blk
        data = data[:n * blk]
        if tag == _PCM and bits == 16:
            x = np.frombuffer(data, dtype='<i2').reshape(n, ch)
            if ch == 1:
                return np.ascontiguousarray(x[:, 0]), int(sr)
            y = x.astype(np.float32) / np.float32(32768.0)
        elif tag == _PCM and bits == 8:
            y = (np.frombuffer(data, dtype=np.uint8).reshape(n, ch).astype(np.float32) - np.float32(128.0)) \
                / np.float32(128.0)
        elif tag == _PCM and bits == 24:
            b = np.frombuffer(data, dtype=np.uint8).reshape(n, ch, 3).astype(np.int32)
            v = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
            v = np.where(v >= (1 << 23), v - (1 << 24), v)
            y = (v.astype(np.float64) / 8388608.0).astype(np.float32)
        elif tag == _PCM and bits == 32:
            y = (np.frombuffer(data, dtype='<i4').reshape(n, ch).astype(np.float64) / 2147483648.0).astype(np.float32)
        elif tag == _FLOAT and bits == 32:
            y = np.frombuffer(data, dtype='<f4').reshape(n, ch).astype(np.float32)
        elif tag == _FLOAT and bits == 64:
            y = np.frombuffer(data, dtype='<f8').reshape(n, ch).astype(np.float32)
        else:
            raise ValueError('unsupported WAV encoding tag={} bits={}'.format(tag, bits))
        if ch == 1:
            y = y[:, 0]
        elif ms_channel is not None:
            y = y[:, ms_channel]                       # NISQA_lib.py:2300-2302
        else:
            y = np.mean(y.T, axis=0, dtype=np.float32)  # librosa.to_mono
        return np.ascontiguousarray(y, dtype=np.float32), int(sr)
    except Exception:
        raise ValueError('Could not load file {}'.format(path))      # NISQA_lib.py:2305-2306

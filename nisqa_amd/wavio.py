"""WAV ingest for the predict path: what ``lb.load(path, sr=None[, mono=False])`` does at
reference nisqa/NISQA_lib.py:2299-2304, for RIFF / RF64 / RIFX (big-endian) WAVE files and for FLAC files, without
librosa/soundfile.  (FLAC streams are decoded by libnisqa_ingest.so -- csrc/flac.hpp, every frame CRC and the stream's MD5
verified -- and scaled here like soundfile does: integer sample / 2**(bits - 1).)

soundfile semantics: integer PCM -> float32 scaled by 1/2**(8*bytes-1) of its CONTAINER (1-4 bytes per sample = bits
rounded up; 12- or 20-bit samples sit left-justified in 2 / 3 bytes; a 1-byte container is unsigned, offset
128); G.711 A-law / mu-law WAVs expand to their 16-bit table values, then 1/32768; float WAVs pass through; multi-channel audio is averaged (librosa.to_mono) unless
``ms_channel`` selects one channel.  The native sample rate is returned (ms_sr=None).
Mono PCM16 -- the common case -- is returned as int16 so that only 2 bytes/sample cross PCIe; the
1/32768 scaling then happens on the GPU (nisqa_pcm16_to_f32), bit-identical to the host scaling.
"""
import os
import struct

import numpy as np

_PCM, _FLOAT, _ALAW, _MULAW, _EXT = 1, 3, 6, 7, 0xFFFE
_FLAC = 0xF1AC                                     # lib.WAV_TAG_FLAC: not a WAVE tag, a FLAC stream


def _g711_tables():
    """G.711 expansion tables (ITU-T G.711; the 16-bit values libsndfile's alaw.c / ulaw.c hold and soundfile then scales
    by 1/32768 like any 16-bit PCM): mu-law -> +-(((m << 3) + 0x84) << e) - 0x84), A-law -> +-((m << 4) + 8) for e = 0,
    +-(((m << 4) + 0x108) << (e - 1)) otherwise, on the bit-inverted (mu) / 0x55-toggled (A) code word."""
    code = np.arange(256, dtype=np.int32)
    u = ~code & 0xFF
    mag = ((((u & 0x0F) << 3) + 0x84) << ((u >> 4) & 7)) - 0x84
    mulaw = np.where(u & 0x80, -mag, mag).astype(np.int16)
    a = code ^ 0x55
    e, m = (a >> 4) & 7, a & 0x0F
    mag = np.where(e == 0, (m << 4) + 8, ((m << 4) + 0x108) << np.maximum(e - 1, 0))
    alaw = np.where(a & 0x80, mag, -mag).astype(np.int16)
    return alaw, mulaw


_ALAW_TAB, _MULAW_TAB = _g711_tables()


class Header(object):
    """An opened WAV file whose RIFF chunks have been walked: format fields, position and frame count of the data
    chunk.  ``fast`` marks mono PCM16, whose data chunk can be copied verbatim (2 bytes/sample cross PCIe)."""
    __slots__ = ('path', 'fd', 'tag', 'ch', 'sr', 'blk', 'bits', 'data_off', 'n', 'fast', 'ms_channel', 'be', 'info')

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None


def _walk(fd):
    size_file = os.fstat(fd).st_size
    head = os.pread(fd, 4096, 0)
    if len(head) < 12 or head[0:4] not in (b'RIFF', b'RF64', b'RIFX') or head[8:12] != b'WAVE':
        raise ValueError('not a RIFF/WAVE file')
    be = head[0:4] == b'RIFX'                          # big-endian variant: every header field and every sample
    E = '>' if be else '<'
    pos, fmt = 12, None
    while pos + 8 <= size_file:
        hdr = head[pos:pos + 8] if pos + 8 <= len(head) else os.pread(fd, 8, pos)
        if len(hdr) < 8:
            break
        cid, size = hdr[0:4], struct.unpack(E + 'I', hdr[4:8])[0]
        body = pos + 8
        if cid == b'fmt ':
            raw = head[body:body + 28] if body + 28 <= len(head) else os.pread(fd, 28, body)
            tag, ch, sr, _, blk, bits = struct.unpack(E + 'HHIIHH', raw[:16])
            if tag == _EXT and size >= 28 and len(raw) >= 28:
                tag = struct.unpack(E + 'I', raw[24:28])[0] & 0xFFFF      # Data1 of the sub-format GUID (a 32-bit field)
            fmt = (tag, ch, sr, blk, bits, be)
        elif cid == b'data':
            if size == 0xFFFFFFFF or body + size > size_file:
                size = size_file - body
            if fmt is None:
                break
            return fmt, body, size
        pos = body + size + (size & 1)
    raise ValueError('missing fmt/data chunk')


def _probe_flac(path, fd, ms_channel):
    """STREAMINFO of a FLAC file through the native ingest library -> Header (tag _FLAC; ``info`` keeps the native record)."""
    import ctypes
    from . import lib as _lib
    L = _lib.load_ingest()
    info = (_lib.WavInfo * 1)()
    paths = (ctypes.c_char_p * 1)(os.fsencode(path))
    if L.nisqa_ingest_probe(paths, 1, info, 1) != 0 or info[0].tag != _FLAC:
        raise ValueError('not a FLAC stream this decoder reads')
    h = Header()
    h.path, h.fd, h.tag, h.ch, h.sr, h.blk, h.bits, h.be = path, fd, _FLAC, info[0].channels, int(info[0].sample_rate), \
        info[0].block_align, info[0].bits, False
    h.data_off, h.n, h.ms_channel, h.fast, h.info = info[0].data_offset, int(info[0].n_frames), ms_channel, False, info
    return h


def _decode_flac(h):
    """FLAC -> int16 [n] (mono 16-bit, like mono PCM16) or float32 [n]: integer sample / 2**(bits - 1), as libsndfile hands a
    FLAC stream to soundfile's float32 read; channels then like any other file."""
    import ctypes
    from . import lib as _lib
    v = np.empty((h.n, h.ch), dtype=np.int32)
    rc = _lib.load_ingest().nisqa_ingest_decode_flac(os.fsencode(h.path), h.info, ctypes.c_void_p(v.ctypes.data))
    if rc != 0:
        raise ValueError('FLAC stream does not decode (status {})'.format(rc))
    if h.ch == 1 and h.bits == 16:
        return v[:, 0].astype(np.int16)
    y = (v.astype(np.float64) / float(1 << (h.bits - 1))).astype(np.float32)
    if h.ch == 1:
        y = y[:, 0]
    elif h.ms_channel is not None:
        y = y[:, h.ms_channel]                     # NISQA_lib.py:2300-2302
    else:
        y = np.mean(y.T, axis=0, dtype=np.float32)  # librosa.to_mono
    return np.ascontiguousarray(y, dtype=np.float32)


def probe(path, ms_channel=None):
    """Open ``path`` and parse its header -> Header (caller closes).  Raises the reference's load error."""
    fd = None
    try:
        fd = os.open(path, os.O_RDONLY)
        magic = os.pread(fd, 4, 0)
        if magic == b'fLaC' or magic[:3] == b'ID3':
            return _probe_flac(path, fd, ms_channel)
        (tag, ch, sr, blk, bits, be), off, size = _walk(fd)
        if ch < 1 or blk != ch * ((bits + 7) // 8):
            raise ValueError('bad block align')
        if not ((tag == _PCM and 1 <= bits <= 32) or (tag == _FLOAT and bits in (32, 64))
                or (tag in (_ALAW, _MULAW) and bits == 8)):
            raise ValueError('unsupported WAV encoding tag={} bits={}'.format(tag, bits))
        h = Header()
        h.path, h.fd, h.tag, h.ch, h.sr, h.blk, h.bits, h.be = path, fd, tag, ch, int(sr), blk, bits, be
        h.data_off, h.n, h.ms_channel = off, size // blk, ms_channel
        h.fast = tag == _PCM and bits == 16 and ch == 1 and not be
        h.info = None
        return h
    except Exception:
        if fd is not None:
            os.close(fd)
        raise ValueError('Could not load file {}'.format(path))      # NISQA_lib.py:2305-2306


def read_data_into(h, out):
    """Copy the data chunk of ``h`` (h.n * h.blk bytes) into the writable buffer ``out`` with preadv."""
    try:
        mv = memoryview(out).cast('B')
        want, got = h.n * h.blk, 0
        if len(mv) != want:
            raise ValueError('staging slice has the wrong size')
        while got < want:
            r = os.preadv(h.fd, [mv[got:]], h.data_off + got)
            if r <= 0:
                raise ValueError('short read')
            got += r
    except Exception:
        raise ValueError('Could not load file {}'.format(h.path))


def _decode(h, data):
    """bytes of the data chunk -> int16 [n] (mono PCM16) or float32 [n], with lb.load's semantics."""
    tag, ch, bits, n = h.tag, h.ch, h.bits, h.n
    E = '>' if h.be else '<'
    cont = h.blk // ch                             # bytes per sample: libsndfile reads the container, whatever `bits` says
    if tag == _PCM and cont == 2:
        x = np.frombuffer(data, dtype=E + 'i2').reshape(n, ch)
        if ch == 1:
            return x[:, 0].astype(np.int16)        # (native byte order)
        y = x.astype(np.float32) / np.float32(32768.0)
    elif tag in (_ALAW, _MULAW):                   # 8-bit companded -> the 16-bit value of the G.711 table -> / 32768
        tab = _ALAW_TAB if tag == _ALAW else _MULAW_TAB
        y = tab[np.frombuffer(data, dtype=np.uint8).reshape(n, ch)].astype(np.float32) / np.float32(32768.0)
    elif tag == _PCM and cont == 1:
        y = (np.frombuffer(data, dtype=np.uint8).reshape(n, ch).astype(np.float32) - np.float32(128.0)) \
            / np.float32(128.0)
    elif tag == _PCM and cont == 3:
        b = np.frombuffer(data, dtype=np.uint8).reshape(n, ch, 3).astype(np.int32)
        if h.be:
            b = b[..., ::-1]
        v = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
        v = np.where(v >= (1 << 23), v - (1 << 24), v)
        y = (v.astype(np.float64) / 8388608.0).astype(np.float32)
    elif tag == _PCM and cont == 4:
        y = (np.frombuffer(data, dtype=E + 'i4').reshape(n, ch).astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == _FLOAT and bits == 32:
        y = np.frombuffer(data, dtype=E + 'f4').reshape(n, ch).astype(np.float32)
    else:
        y = np.frombuffer(data, dtype=E + 'f8').reshape(n, ch).astype(np.float32)
    if ch == 1:
        y = y[:, 0]
    elif h.ms_channel is not None:
        y = y[:, h.ms_channel]                     # NISQA_lib.py:2300-2302
    else:
        y = np.mean(y.T, axis=0, dtype=np.float32)  # librosa.to_mono
    return np.ascontiguousarray(y, dtype=np.float32)


def _read_decoded(h):
    if h.tag == _FLAC:
        return _decode_flac(h)
    data = bytearray(h.n * h.blk)
    read_data_into(h, data)
    return _decode(h, data)


def decode_f32(h):
    """float32 [h.n] samples of an opened file (any supported encoding)."""
    try:
        y = _read_decoded(h)
        if y.dtype == np.int16:
            y = y.astype(np.float32) / np.float32(32768.0)
        return y
    except Exception:
        raise ValueError('Could not load file {}'.format(h.path))


def read_wav(path, ms_channel=None):
    """-> (samples, sr); samples is int16 [n] (mono PCM16) or float32 [n]."""
    h = probe(path, ms_channel)
    try:
        return np.ascontiguousarray(_read_decoded(h)), h.sr
    except Exception:
        raise ValueError('Could not load file {}'.format(path))      # NISQA_lib.py:2305-2306
    finally:
        h.close()

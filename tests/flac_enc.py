"""A small FLAC ENCODER for the tests (test infrastructure only: nothing under nisqa_amd/ imports it).

There is no FLAC tool, library or file in this image, so the streams the decoder (nisqa_amd/csrc/flac.hpp) is tested on are
written here, from the same published format description, but as an independent piece of code in another language: big-int bit
packing, bit-by-bit CRCs, hashlib's MD5.  Every feature of the format the decoder claims is exercisable: CONSTANT / VERBATIM /
FIXED 0-4 / LPC subframes, Rice partitions with 4- or 5-bit parameters and escape partitions, wasted bits, left-side /
right-side / mid-side stereo, 8- / 16-bit explicit block sizes, long (multi-byte) frame numbers, extra metadata blocks, an ID3v2
tag in front, a STREAMINFO without length or without MD5.
"""
import hashlib

import numpy as np


class BitWriter(object):
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, val, bits):
        if bits:
            self.v = (self.v << bits) | (int(val) & ((1 << bits) - 1))
            self.n += bits

    def unary(self, q):
        self.v = (self.v << (q + 1)) | 1
        self.n += q + 1

    def align(self):
        pad = -self.n % 8
        self.v <<= pad
        self.n += pad

    def bytes(self):
        assert self.n % 8 == 0
        return self.v.to_bytes(self.n // 8, 'big')


def crc_bitwise(data, poly, width):
    top, mask, c = 1 << (width - 1), (1 << width) - 1, 0
    for byte in data:
        c ^= byte << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


def utf8_number(n):
    if n < 0x80:
        return bytes([n])
    cont = []
    for lead_bits, lead in ((5, 0xC0), (4, 0xE0), (3, 0xF0), (2, 0xF8), (1, 0xFC), (0, 0xFE)):
        cont.append(0x80 | (n & 0x3F))
        n >>= 6
        if n < (1 << lead_bits):
            return bytes([lead | n] + cont[::-1])
    raise ValueError('number too large')


def _rice_bits(res, k):
    u = np.where(res >= 0, 2 * res, -2 * res - 1).astype(np.int64)
    return int((u >> k).sum()) + (k + 1) * len(u)


def _write_residual(w, res, order, bs, porder, method, escape):
    """res: residuals of samples [order, bs).  method 0 / 1 = 4- / 5-bit Rice parameters; escape: the set of partition indices
    written raw (escape code + 5-bit width)."""
    pbits = 5 if method else 4
    w.put(method, 2)
    w.put(porder, 4)
    parts = 1 << porder
    assert porder == 0 or bs % parts == 0
    per, at = bs >> porder, 0
    for p in range(parts):
        cnt = per - (order if p == 0 else 0)
        r = np.asarray(res[at:at + cnt], dtype=np.int64)
        at += cnt
        if p in escape:
            nb = 0 if not len(r) or not r.any() else max(int(abs(int(x))).bit_length() for x in r) + 1
            w.put((1 << pbits) - 1, pbits)
            w.put(nb, 5)
            for x in r:
                w.put(int(x), nb)
            continue
        kmax = (1 << pbits) - 2
        best = min(range(0, min(kmax, 20) + 1), key=lambda k: _rice_bits(r, k)) if len(r) else 0
        w.put(best, pbits)
        for x in r:
            x = int(x)
            u = 2 * x if x >= 0 else -2 * x - 1
            w.unary(u >> best)
            w.put(u & ((1 << best) - 1), best)
    assert at == len(res)


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _subframe(w, x, bits, kind):
    """x: the block's samples of one channel (Python ints / int64), bits: their width in this subframe.  kind: dict with 'type' in
    'constant' | 'verbatim' | 'fixed' | 'lpc' and options order, porder, method, escape, precision."""
    x = [int(v) for v in x]
    bs = len(x)
    wasted = 0
    if kind.get('wasted', True) and any(x):
        while all((v >> wasted) & 1 == 0 for v in x):
            wasted += 1
    if wasted:
        x = [v >> wasted for v in x]
    b = bits - wasted
    t = kind['type']
    order = kind.get('order', 0)
    w.put(0, 1)
    w.put({'constant': 0, 'verbatim': 1, 'fixed': 8 + order, 'lpc': 32 + order - 1}[t], 6)
    if wasted:
        w.put(1, 1)
        w.unary(wasted - 1)
    else:
        w.put(0, 1)
    if t == 'constant':
        assert all(v == x[0] for v in x)
        w.put(x[0], b)
        return
    if t == 'verbatim':
        for v in x:
            w.put(v, b)
        return
    for v in x[:order]:
        w.put(v, b)
    if t == 'fixed':
        coef, shift = _FIXED[order], 0
    else:
        prec = kind.get('precision', 12)
        a = np.asarray(x, dtype=np.float64)
        rows = np.stack([a[order - 1 - j:bs - 1 - j] for j in range(order)], axis=1)
        sol = np.linalg.lstsq(rows, a[order:], rcond=None)[0] if bs > 2 * order else np.zeros(order)
        peak = max(1e-9, float(np.abs(sol).max()))
        shift = int(max(0, min(15, np.floor(np.log2(((1 << (prec - 1)) - 1) / peak)))))
        coef = [int(np.clip(np.round(c * (1 << shift)), -(1 << (prec - 1)), (1 << (prec - 1)) - 1)) for c in sol]
        w.put(prec - 1, 4)
        w.put(shift, 5)
        for c in coef:
            w.put(c, prec)
    res = []
    for i in range(order, bs):
        pred = sum(c * x[i - 1 - j] for j, c in enumerate(coef))
        res.append(x[i] - (pred >> shift))
    porder = kind.get('porder', 0)
    while porder and (bs % (1 << porder) or (bs >> porder) < order):      # (a short last block)
        porder -= 1
    _write_residual(w, res, order, bs, porder, kind.get('method', 0), set(kind.get('escape', ())))


_BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
_SR_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
_SS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}


def encode(samples, sr, bits, blocksize=4096, kinds=None, stereo=0, header_from_streaminfo=False, variable=False, id3=0,
           extra_blocks=(), with_total=True, with_md5=True, first_frame_number=0):
    """samples: int array [n] or [n, ch]; kinds: a subframe description (dict) or a function (frame, channel) -> dict; stereo:
    channel assignment for two channels (0 independent, 8 left-side, 9 right-side, 10 mid-side) or a function frame -> one of
    them; variable: the variable-blocksize stream form (frames carry their first SAMPLE number)."""
    x = np.asarray(samples, dtype=np.int64)
    if x.ndim == 1:
        x = x[:, None]
    n, ch = x.shape
    kinds = kinds or {'type': 'fixed', 'order': 2}
    frames, at, k = [], 0, 0
    min_frame, max_frame = 1 << 24, 0
    while at < n:
        bs = min(blocksize, n - at)
        blk = x[at:at + bs]
        ca = ((stereo(k) if callable(stereo) else stereo) or 1) if ch == 2 else ch - 1      # 1 = two independent channels
        hdr = BitWriter()
        hdr.put(0x3FFE, 14)
        hdr.put(0, 1)
        hdr.put(1 if variable else 0, 1)
        if bs in _BS_CODES and not (k % 5 == 4 and bs <= 65536):          # every fifth frame spells its block size out
            bs_code, bs_extra = _BS_CODES[bs], None
        elif bs <= 256:
            bs_code, bs_extra = 6, (bs - 1, 8)
        else:
            bs_code, bs_extra = 7, (bs - 1, 16)
        if header_from_streaminfo:
            sr_code, sr_extra = 0, None
        elif sr in _SR_CODES:
            sr_code, sr_extra = _SR_CODES[sr], None
        elif sr % 1000 == 0 and sr // 1000 < 256:
            sr_code, sr_extra = 12, (sr // 1000, 8)
        elif sr < 65536:
            sr_code, sr_extra = 13, (sr, 16)
        else:
            sr_code, sr_extra = 14, (sr // 10, 16)
        hdr.put(bs_code, 4)
        hdr.put(sr_code, 4)
        hdr.put(ca, 4)
        hdr.put(0 if header_from_streaminfo else _SS_CODES[bits], 3)
        hdr.put(0, 1)
        head = hdr.bytes() + utf8_number(at if variable else k + first_frame_number)
        tail = BitWriter()
        if bs_extra:
            tail.put(*bs_extra)
        if sr_extra:
            tail.put(*sr_extra)
        head += tail.bytes()
        head += bytes([crc_bitwise(head, 0x07, 8)])
        w = BitWriter()
        if ca == 8:
            chans, widths = [blk[:, 0], blk[:, 0] - blk[:, 1]], [bits, bits + 1]
        elif ca == 9:
            chans, widths = [blk[:, 0] - blk[:, 1], blk[:, 1]], [bits + 1, bits]
        elif ca == 10:
            chans, widths = [(blk[:, 0] + blk[:, 1]) >> 1, blk[:, 0] - blk[:, 1]], [bits, bits + 1]
        else:
            chans, widths = [blk[:, c] for c in range(ch)], [bits] * ch
        for c in range(ch):
            kind = kinds(k, c) if callable(kinds) else kinds
            if kind.get('order', 0) > bs:
                kind = {'type': 'verbatim'}
            _subframe(w, chans[c], widths[c], kind)
        w.align()
        body = head + w.bytes()
        frame = body + crc_bitwise(body, 0x8005, 16).to_bytes(2, 'big')
        frames.append(frame)
        min_frame, max_frame = min(min_frame, len(frame)), max(max_frame, len(frame))
        at += bs
        k += 1
    nbytes = (bits + 7) // 8
    pcm = b''.join(int(v).to_bytes(nbytes, 'little', signed=True) for v in x.reshape(-1)) if nbytes == 3 else \
        x.astype({1: '<i1', 2: '<i2'}[nbytes]).tobytes()
    si = BitWriter()
    si.put(blocksize if n >= blocksize else max(16, n), 16)
    si.put(blocksize, 16)
    si.put(min_frame if frames else 0, 24)
    si.put(max_frame, 24)
    si.put(sr, 20)
    si.put(ch - 1, 3)
    si.put(bits - 1, 5)
    si.put(n if with_total else 0, 36)
    streaminfo = si.bytes() + (hashlib.md5(pcm).digest() if with_md5 else bytes(16))
    assert len(streaminfo) == 34
    blocks = [(0, streaminfo)] + list(extra_blocks)
    out = b''
    if id3:
        size = id3
        out += b'ID3\x04\x00\x00' + bytes([(size >> 21) & 0x7F, (size >> 14) & 0x7F, (size >> 7) & 0x7F, size & 0x7F]) + bytes(size)
    out += b'fLaC'
    for i, (t, body) in enumerate(blocks):
        out += bytes([(0x80 if i == len(blocks) - 1 else 0) | t]) + len(body).to_bytes(3, 'big') + body
    return out + b''.join(frames)

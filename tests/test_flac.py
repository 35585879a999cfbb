"""CPU tests of the FLAC decoder of libnisqa_ingest.so (nisqa_amd/csrc/flac.hpp; include/nisqa_ingest.h): what lb.load reads
through soundfile for a .flac file (reference nisqa/NISQA_lib.py:2299-2306).

PARITY UNPINNED against libFLAC: the image holds no FLAC encoder, library or file, so the streams come from tests/flac_enc.py
-- an independent writer of the same published format (big-int bit packing, bit-serial CRCs, hashlib MD5).  What pins the
decoder beyond that: it accepts a stream only if every frame's CRC-8 / CRC-16, the stream length and STREAMINFO's MD5 of the
samples hold, so a file it misreads raises the reference's 'Could not load file' instead of yielding wrong audio.
"""
import ctypes
import os
import struct
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import flac_enc                      # noqa: E402
from nisqa_amd import lib, synth, wavio  # noqa: E402

FLAC = 0xF1AC


def _signal(n, ch=1, bits=16, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    amp = (1 << (bits - 1)) * 0.3
    cols = []
    for c in range(ch):
        y = amp * (0.6 * np.sin(2 * np.pi * t / (97.0 + 5 * c)) + 0.3 * np.sin(2 * np.pi * t / 13.1 + c)) + rng.normal(0, amp * 0.02, n)
        cols.append(np.round(y).astype(np.int64))
    if ch == 2:
        cols[1] = cols[0] + np.round(rng.normal(0, amp * 0.01, n)).astype(np.int64)      # correlated: the side channel is small
    x = np.stack(cols, axis=1)
    return np.clip(x, -(1 << (bits - 1)), (1 << (bits - 1)) - 1)


def _probe(path):
    L = lib.load_ingest()
    info = (lib.WavInfo * 1)()
    paths = (ctypes.c_char_p * 1)(os.fsencode(path))
    bad = L.nisqa_ingest_probe(paths, 1, info, 1)
    return bad, info, paths


def _decode(path):
    bad, info, _ = _probe(path)
    assert bad == 0, 'probe status %d' % info[0].status
    v = np.full((info[0].n_frames, info[0].channels), 0x55555555, dtype=np.int32)
    rc = lib.load_ingest().nisqa_ingest_decode_flac(os.fsencode(path), info, ctypes.c_void_p(v.ctypes.data))
    return rc, info[0], v


def _roundtrip(tmp_path, name, x, sr, bits, **kw):
    p = str(tmp_path / (name + '.flac'))
    with open(p, 'wb') as f:
        f.write(flac_enc.encode(x, sr, bits, **kw))
    rc, info, v = _decode(p)
    assert rc == 0, '%s: decode status %d' % (name, rc)
    x2 = np.asarray(x).reshape(len(x), -1)
    assert (info.tag, info.sample_rate, info.bits, info.channels, info.n_frames) == (FLAC, sr, bits, x2.shape[1], len(x2)), name
    assert info.block_align == x2.shape[1] * ((bits + 7) // 8)
    np.testing.assert_array_equal(v, x2, err_msg=name)
    return p


def test_crc_and_md5_of_the_decoder_agree_with_independent_implementations(tmp_path):
    # (through a stream: a one-frame verbatim file decodes only if CRC-8, CRC-16 and MD5 all agree with flac_enc / hashlib;
    #  known answers of the two CRCs for b'123456789' in their catalogue forms: CRC-8 0xF4, CRC-16/UMTS 0xFEE8)
    assert flac_enc.crc_bitwise(b'123456789', 0x07, 8) == 0xF4
    assert flac_enc.crc_bitwise(b'123456789', 0x8005, 16) == 0xFEE8
    x = _signal(777, seed=1)
    _roundtrip(tmp_path, 'one', x, 16000, 16, kinds={'type': 'verbatim'}, blocksize=1024)


@pytest.mark.parametrize('order', [0, 1, 2, 3, 4])
def test_fixed_predictors(tmp_path, order):
    _roundtrip(tmp_path, 'fixed%d' % order, _signal(5000, seed=order), 48000, 16, kinds={'type': 'fixed', 'order': order})


@pytest.mark.parametrize('order,precision', [(1, 12), (8, 12), (12, 15), (32, 10), (3, 5)])
def test_lpc_subframes(tmp_path, order, precision):
    _roundtrip(tmp_path, 'lpc', _signal(4200, seed=order), 44100, 16, kinds={'type': 'lpc', 'order': order, 'precision': precision},
               blocksize=1152)


def test_rice_partitions_parameter_widths_and_escapes(tmp_path):
    x = _signal(8192, seed=3)
    x[3000:3300] = 0                                                 # a silent stretch: partitions of zero residuals (k = 0; escape width 0)
    for name, kind in [('p3', {'type': 'fixed', 'order': 2, 'porder': 3}),
                       ('p8', {'type': 'fixed', 'order': 1, 'porder': 8}),
                       ('rice2', {'type': 'lpc', 'order': 6, 'porder': 2, 'method': 1}),
                       ('esc', {'type': 'fixed', 'order': 2, 'porder': 2, 'escape': (0, 2)}),
                       ('esc5', {'type': 'fixed', 'order': 3, 'porder': 4, 'method': 1, 'escape': (1, 11, 15)}),
                       ('esc0', {'type': 'fixed', 'order': 0, 'porder': 0, 'escape': (0,)})]:
        _roundtrip(tmp_path, name, x, 48000, 16, kinds=kind, blocksize=4096)
    loud = (np.random.default_rng(9).integers(-32768, 32767, 4096)).astype(np.int64)      # white full-scale noise: large Rice parameters
    _roundtrip(tmp_path, 'loud', loud, 48000, 16, kinds={'type': 'fixed', 'order': 4, 'porder': 1}, blocksize=4096)


def test_constant_verbatim_and_wasted_bits(tmp_path):
    x = np.zeros(3000, dtype=np.int64)
    x[1000:2000] = -1234
    kinds = lambda k, c: {'type': 'constant'}                       # noqa: E731
    _roundtrip(tmp_path, 'const', x, 8000, 16, kinds=kinds, blocksize=1000)
    y = _signal(4096, seed=4) // 8 * 8                               # three wasted bits in every block
    assert (y & 7).max() == 0
    for kind in ({'type': 'verbatim'}, {'type': 'fixed', 'order': 2}, {'type': 'lpc', 'order': 4}):
        _roundtrip(tmp_path, 'wasted', y, 16000, 16, kinds=kind, blocksize=1024)
    z = np.full(600, 64, dtype=np.int64)                             # a constant with wasted bits
    _roundtrip(tmp_path, 'constw', z, 16000, 16, kinds={'type': 'constant'}, blocksize=600)


@pytest.mark.parametrize('assignment', [0, 8, 9, 10])
def test_stereo_decorrelations(tmp_path, assignment):
    x = _signal(6000, ch=2, seed=assignment)
    x[17] = [32767, -32768]                                          # the widest side value: 17 bits
    x[18] = [-32768, 32767]
    _roundtrip(tmp_path, 'st', x, 44100, 16, stereo=assignment, kinds={'type': 'fixed', 'order': 2, 'porder': 2})
    _roundtrip(tmp_path, 'stv', x, 44100, 16, stereo=assignment, kinds={'type': 'verbatim'})
    # an encoder picks the assignment per frame
    _roundtrip(tmp_path, 'mix', x, 44100, 16, stereo=lambda k: (0, 8, 9, 10)[k % 4], kinds={'type': 'lpc', 'order': 8}, blocksize=576)


@pytest.mark.parametrize('bits', [8, 12, 16, 20, 24])
def test_sample_widths(tmp_path, bits):
    x = _signal(3000, ch=2, bits=bits, seed=bits)
    x[5] = [(1 << (bits - 1)) - 1, -(1 << (bits - 1))]
    _roundtrip(tmp_path, 'w%d' % bits, x, 48000, bits, stereo=10, kinds={'type': 'lpc', 'order': 8, 'precision': 14}, blocksize=1024)
    _roundtrip(tmp_path, 'm%d' % bits, x[:, 0], 48000, bits, kinds={'type': 'fixed', 'order': 3})
    # the frame header may leave width and rate to STREAMINFO (the only way to carry e.g. 18 bits)
    _roundtrip(tmp_path, 'h%d' % bits, x[:, 0], 48000, bits, header_from_streaminfo=True)
    if bits == 16:
        _roundtrip(tmp_path, 'w18', _signal(2000, bits=18, seed=2), 48000, 18, header_from_streaminfo=True)


def test_block_sizes_frame_numbers_and_sample_rates(tmp_path):
    x = _signal(30000, seed=7)
    _roundtrip(tmp_path, 'b192', x, 16000, 16, blocksize=192)                    # 157 frames: two-byte frame numbers
    _roundtrip(tmp_path, 'b100', x[:1234], 16000, 16, blocksize=100)             # 8-bit explicit block size, short last block
    _roundtrip(tmp_path, 'b1000', x[:4321], 16000, 16, blocksize=1000)           # 16-bit explicit block size
    _roundtrip(tmp_path, 'b16', x[:100], 16000, 16, blocksize=16)
    _roundtrip(tmp_path, 'big', x[:20000], 16000, 16, blocksize=16384)
    _roundtrip(tmp_path, 'late', x[:3000], 16000, 16, blocksize=256, first_frame_number=70000)   # three- / four-byte numbers
    _roundtrip(tmp_path, 'var', x[:9000], 16000, 16, blocksize=1024, variable=True)              # sample numbers instead
    for sr in (8000, 11025, 12000, 22050, 37800, 88200, 96000, 192000, 352800):
        _roundtrip(tmp_path, 'sr%d' % sr, x[:700], sr, 16, blocksize=512)
    _roundtrip(tmp_path, 'one', x[:1], 16000, 16, blocksize=4096)                # a one-sample stream


def test_metadata_id3_and_open_streaminfo(tmp_path):
    x = _signal(5000, ch=2, seed=8)
    padding = (1, bytes(8192))                                                   # what the stock encoder leaves behind STREAMINFO
    comment = (4, struct.pack('<I', 4) + b'test' + struct.pack('<I', 0))
    _roundtrip(tmp_path, 'meta', x, 48000, 16, extra_blocks=(comment, padding), stereo=10)
    _roundtrip(tmp_path, 'id3', x, 48000, 16, id3=300)
    _roundtrip(tmp_path, 'id3far', x, 48000, 16, id3=6000, extra_blocks=(padding,))   # STREAMINFO beyond the probe's first read
    _roundtrip(tmp_path, 'nomd5', x, 48000, 16, with_md5=False)
    p = _roundtrip(tmp_path, 'nototal', x, 48000, 16, with_total=False)          # counted by decoding at probe time
    y, sr = wavio.read_wav(p)
    assert sr == 48000 and len(y) == 5000


def _write(tmp_path, name, blob):
    p = str(tmp_path / name)
    with open(p, 'wb') as f:
        f.write(blob)
    return p


def test_damaged_streams_are_refused_not_decoded_wrongly(tmp_path):
    x = _signal(6000, seed=11)
    blob = flac_enc.encode(x, 48000, 16, blocksize=1024, kinds={'type': 'fixed', 'order': 2})
    first = 4 + 4 + 34
    rng = np.random.default_rng(0)
    refused = 0
    for pos in [first + 10, first + 200, len(blob) // 2, len(blob) - 3] + rng.integers(first, len(blob), 40).tolist():
        bad = bytearray(blob)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        p = _write(tmp_path, 'flip.flac', bytes(bad))
        nbad, info, _ = _probe(p)
        assert nbad == 0                                                         # (STREAMINFO itself is intact)
        v = np.zeros((6000, 1), dtype=np.int32)
        rc = lib.load_ingest().nisqa_ingest_decode_flac(os.fsencode(p), info, ctypes.c_void_p(v.ctypes.data))
        assert rc in (lib.WAV_ERR_FORMAT, lib.WAV_ERR_READ), 'a flipped bit at byte %d went unnoticed' % pos
        with pytest.raises(ValueError, match='Could not load file'):
            wavio.read_wav(p)
        refused += 1
    assert refused == 44
    # a flipped bit in a stream WITHOUT frame damage is impossible (every byte of a frame is under its CRC-16); the MD5 guards the
    # decoder itself: a stream whose STREAMINFO carries another MD5 is refused although every frame checks out
    wrong = bytearray(blob)
    wrong[first - 1] ^= 0xFF
    p = _write(tmp_path, 'md5.flac', bytes(wrong))
    assert _decode(p)[0] == lib.WAV_ERR_FORMAT
    # truncated files, a STREAMINFO that promises more or fewer samples than the frames hold
    for cut in (len(blob) - 1, len(blob) - 700, first + 3, first):
        p = _write(tmp_path, 'cut.flac', blob[:cut])
        nbad, info, _ = _probe(p)
        v = np.zeros((6000, 1), dtype=np.int32)
        # (a file cut right behind STREAMINFO cannot hold the promised samples: refused at probe time already)
        assert nbad == 1 or lib.load_ingest().nisqa_ingest_decode_flac(os.fsencode(p), info, ctypes.c_void_p(v.ctypes.data)) != 0
        assert nbad == (1 if cut <= first + 3 else 0)
    for total in (6001, 5999, 1024, 7000):
        lie = bytearray(blob)
        lie[8 + 13:8 + 18] = bytes([(lie[8 + 13] & 0xF0) | ((total >> 32) & 0x0F)]) + (total & 0xFFFFFFFF).to_bytes(4, 'big')
        p = _write(tmp_path, 'lie.flac', bytes(lie))
        nbad, info, _ = _probe(p)
        assert nbad == 0 and info[0].n_frames == total
        v = np.zeros((total + 16, 1), dtype=np.int32)
        assert lib.load_ingest().nisqa_ingest_decode_flac(os.fsencode(p), info, ctypes.c_void_p(v.ctypes.data)) != 0
        assert not v[total:].any()                                               # and nothing was written beyond the promised length
    # a total the file cannot hold (a ~100-byte file claiming 2**36 - 1 samples must not size a 100 GB staging buffer):
    # refused at probe time, like any other malformed header
    for total in ((1 << 36) - 1, (len(blob) // 10 + 2) * 1024):
        lie = bytearray(blob)
        lie[8 + 13:8 + 18] = bytes([(lie[8 + 13] & 0xF0) | ((total >> 32) & 0x0F)]) + (total & 0xFFFFFFFF).to_bytes(4, 'big')
        p = _write(tmp_path, 'huge.flac', bytes(lie))
        assert _probe(p)[0] == 1
        with pytest.raises(ValueError, match='Could not load file'):
            wavio.read_wav(p)
    # not FLAC at all, an unsupported width, garbage behind the marker
    for name, data in [('short.flac', b'fLaC'), ('junk.flac', b'fLaC' + bytes(range(200))), ('id3only.flac', b'ID3\x04\x00\x00\x00\x00\x00\x10' + bytes(64))]:
        p = _write(tmp_path, name, data)
        assert _probe(p)[0] == 1
        with pytest.raises(ValueError, match='Could not load file'):
            wavio.read_wav(p)
    wide = bytearray(flac_enc.encode(x[:100], 48000, 24, header_from_streaminfo=True))
    wide[8 + 12] |= 1                                                            # bits - 1 = 0b1xxxx: more than 24 bits
    wide[8 + 13] |= 0xF0
    assert _probe(_write(tmp_path, 'wide.flac', bytes(wide)))[0] == 1


def test_read_wav_gives_a_flac_file_the_samples_of_the_same_wav(tmp_path):
    rng = np.random.default_rng(12)
    mono = (rng.standard_normal(4000) * 3000).astype(np.int16)
    st = (rng.standard_normal((4000, 2)) * 3000).astype(np.int16)
    for name, data in (('m', mono), ('s', st)):
        w, f = str(tmp_path / (name + '.wav')), str(tmp_path / (name + '.flac'))
        synth.write_wav(w, data, 48000)
        with open(f, 'wb') as fh:
            fh.write(flac_enc.encode(data, 48000, 16, stereo=10 if data.ndim == 2 else 0))
        for channel in (None, 1) if data.ndim == 2 else (None,):
            yw, srw = wavio.read_wav(w, channel)
            yf, srf = wavio.read_wav(f, channel)
            assert srw == srf and yw.dtype == yf.dtype
            np.testing.assert_array_equal(yw, yf)
    # 24 bits: the WAVE container is scaled by 2**23, the FLAC sample by 2**(bits - 1): the same float32
    v = (rng.standard_normal(3000) * 1e6).astype(np.int32)
    raw = b''.join(int(x & 0xFFFFFF).to_bytes(3, 'little') for x in v)
    hdr = b'RIFF' + struct.pack('<I', 36 + len(raw)) + b'WAVE' + b'fmt ' + struct.pack('<IHHIIHH', 16, 1, 1, 48000, 144000, 3, 24) \
        + b'data' + struct.pack('<I', len(raw))
    w = _write(tmp_path, 'p24.wav', hdr + raw)
    f = _write(tmp_path, 'p24.flac', flac_enc.encode(v, 48000, 24, kinds={'type': 'lpc', 'order': 8}))
    np.testing.assert_array_equal(wavio.read_wav(w)[0], wavio.read_wav(f)[0])


class _ListDataset(object):
    ms_channel = None

    def __init__(self, paths):
        self.paths = paths

    def file_path(self, i):
        return self.paths[i]


def test_staging_iterator_decodes_flac_into_the_slots_wav_files_are_copied_to(tmp_path):
    """A batch of mono 16-bit FLAC files is staged byte for byte like the batch of the same clips as PCM16 WAV files (int16,
    2 bytes per sample over the link); stereo / 24-bit FLAC takes the float32 route of stereo / 24-bit WAV."""
    from nisqa_amd import ingest
    rng = np.random.default_rng(13)
    wavs, flacs = [], []
    for i in range(9):
        n = 900 + 131 * i
        data = (rng.standard_normal(n) * 2000).astype(np.int16) if i != 4 else (rng.standard_normal((n, 2)) * 2000).astype(np.int16)
        sr = 48000 if i % 3 else 16000
        w, f = str(tmp_path / ('c%d.wav' % i)), str(tmp_path / ('c%d.flac' % i))
        synth.write_wav(w, data, sr)
        with open(f, 'wb') as fh:
            fh.write(flac_enc.encode(data, sr, 16, blocksize=256, kinds={'type': 'fixed', 'order': 1 + i % 4, 'porder': i % 3}))
        wavs.append(w)
        flacs.append(f)
    batches = [[0, 1, 2, 3], [4, 5, 6], [7, 8, 0]]
    staged_bytes = {}
    for kind, paths in (('wav', wavs), ('flac', flacs), ('mixed', [p if i % 2 else q for i, (p, q) in enumerate(zip(wavs, flacs))])):
        ing = ingest.Ingest(_ListDataset(paths), batches, pin=False, num_workers=3)
        out = []
        try:
            for staged in ing:
                raw = ing.ring.buf[staged.slot]
                out.append([(g.ids, g.lengths, g.sr, g.is_i16, bytes(raw[g.offset:g.offset + g.nbytes].numpy())) for g in staged.groups])
                ing.ring.release_after(staged.slot, None)
        finally:
            ing.close()
        staged_bytes[kind] = out
    assert staged_bytes['flac'] == staged_bytes['wav'] and staged_bytes['mixed'] == staged_bytes['wav']
    assert any(g[3] for b in staged_bytes['flac'] for g in b) and any(not g[3] for b in staged_bytes['flac'] for g in b)
    # a damaged FLAC file in a batch: the reference's error, naming the file
    bad = bytearray(open(flacs[2], 'rb').read())
    bad[len(bad) // 2] ^= 0x10
    p = _write(tmp_path, 'broken.flac', bytes(bad))
    ing = ingest.Ingest(_ListDataset([flacs[0], p]), [[0, 1]], pin=False, num_workers=2)
    try:
        with pytest.raises(ValueError, match='Could not load file .*broken.flac'):
            next(iter(ing))
    finally:
        ing.close()


@pytest.mark.gpu
def test_predict_csv_on_flac_files_equals_the_same_clips_as_wav_bit_for_bit(tmp_path):
    """nisqaModel(...).predict() in predict_csv mode on FLAC files (mono 16-bit through the int16 staging slots, one stereo and one
    24-bit file through the float route) against the same clips as WAV files: the GPU sees the same bytes, so every output is equal."""
    import pandas as pd
    import helpers
    from nisqa_amd.NISQA_model import nisqaModel
    args = dict(helpers.DIM_ARGS)
    args.update({'pretrained_model': False, 'tr_bs_val': 4, 'tr_num_workers': 0})
    path = str(tmp_path / 'rand.tar')
    torch.save({'args': args, 'model_state_dict': helpers.random_state_dict(7)}, path)
    rng = np.random.default_rng(31)
    names = []
    for i in range(7):
        pcm = synth.synth_pcm16(500 + i, float(rng.uniform(0.4, 1.6)))
        bits, data = 16, pcm
        if i == 3:
            data = np.stack([pcm, np.roll(pcm, 7)], axis=1)
        synth_bits_24 = i == 5
        w, f = str(tmp_path / ('c%d.wav' % i)), str(tmp_path / ('c%d.flac' % i))
        if synth_bits_24:
            v = pcm.astype(np.int32) * 256 + 37
            raw = b''.join(int(x & 0xFFFFFF).to_bytes(3, 'little') for x in v)
            hdr = b'RIFF' + struct.pack('<I', 36 + len(raw)) + b'WAVE' + b'fmt ' + struct.pack('<IHHIIHH', 16, 1, 1, 48000, 144000, 3, 24) \
                + b'data' + struct.pack('<I', len(raw))
            with open(w, 'wb') as fh:
                fh.write(hdr + raw)
            bits, data = 24, v
        else:
            synth.write_wav(w, data, 48000)
        with open(f, 'wb') as fh:
            fh.write(flac_enc.encode(data, 48000, bits, blocksize=4096, stereo=10 if np.ndim(data) == 2 else 0,
                                     kinds={'type': 'fixed', 'order': 2, 'porder': 2}))
        names.append('c%d' % i)
    out = {}
    for ext in ('wav', 'flac'):
        pd.DataFrame({'deg': [n + '.' + ext for n in names]}).to_csv(tmp_path / (ext + '.csv'), index=False)
        a = {'mode': 'predict_csv', 'pretrained_model': path, 'deg': None, 'data_dir': str(tmp_path), 'output_dir': None,
             'csv_file': ext + '.csv', 'csv_deg': 'deg', 'num_workers': 0, 'bs': 4, 'ms_channel': None, 'tr_bs_val': 4, 'tr_num_workers': 0}
        out[ext] = nisqaModel(a).predict()
    cols = ['mos_pred', 'noi_pred', 'dis_pred', 'col_pred', 'loud_pred']
    print('predict_csv on FLAC vs WAV: max |d| %.3g' % np.abs(out['wav'][cols].to_numpy() - out['flac'][cols].to_numpy()).max())
    np.testing.assert_array_equal(out['wav'][cols].to_numpy(), out['flac'][cols].to_numpy())
    # predict_file mode on one FLAC file
    a = {'mode': 'predict_file', 'pretrained_model': path, 'deg': str(tmp_path / 'c0.flac'), 'data_dir': None, 'output_dir': None,
         'csv_file': None, 'csv_deg': None, 'num_workers': 0, 'bs': 1, 'ms_channel': None, 'tr_bs_val': 1, 'tr_num_workers': 0}
    one = nisqaModel(a).predict()
    np.testing.assert_allclose(one[cols].to_numpy()[0], out['wav'][cols].to_numpy()[0], atol=1e-5)

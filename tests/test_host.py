"""CPU tests of the host side: packed layouts vs the device header, plan arithmetic, WAV ingest vs the
oracle's decoder, the C-ABI library (loads, exports every declared symbol -- no compute without a GPU),
the nisqaModel / run_predict plumbing, and the clip-sharded predict loop under gloo (world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest
import torch

import helpers
from nisqa_amd import synth, weights as W, wavio
from nisqa_amd.engine import BatchPlan
from nisqa_amd.melbank import MelTables
from oracle import mel as omel, net as onet

ROOT = helpers.ROOT


# ---- layouts ---------------------------------------------------------------------------------------
def _header_defines(path):
    env = {}
    for line in open(path):
        m = re.match(r'#define\s+(\w+)\s+(.+?)\s*(/\*.*)?$', line)
        if m and not m.group(2).startswith('"'):
            try:
                env[m.group(1)] = int(eval(m.group(2), {}, env))
            except Exception:
                pass
    return env


def test_layout_header_matches_python():
    env = _header_defines(os.path.join(ROOT, 'nisqa_amd', 'csrc', 'layout.hpp'))
    names = [n for n in dir(W) if re.match(r'(CNN|CNNB|CNNX|CNNH|CNNS|LSTM|TD|TDL|TDB|TDBL|TDX|TDXL|PL|PLB|PLX)_', n)]
    assert len(names) > 30
    for n in names:
        assert env[n] == getattr(W, n), n
        assert getattr(W, n) % (8 if n.startswith(('CNNB', 'CNNX', 'CNNH', 'TDB', 'TDX', 'PLB', 'PLX')) else 4) == 0, n   # 16-byte aligned


def test_conv_fragments_roundtrip():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((64, 32, 3, 3))
    fr = W.conv_b_fragments(w).reshape(9, 4, 2, 64, 4)
    for tap, s, nt, lane, kk in [(0, 0, 0, 0, 0), (8, 3, 1, 63, 3), (4, 2, 1, 37, 1), (5, 1, 0, 31, 2)]:
        n, c = (lane & 31) + 32 * nt, 8 * s + 4 * (lane >> 5) + kk
        assert fr[tap, s, nt, lane, kk] == np.float32(w[n, c, tap // 3, tap % 3])
    # emulate the MFMA k-pairing: step s, mfma kk pairs k = 8s+kk (h=0) with 8s+4+kk (h=1) -> every k once
    seen = sorted(8 * s + 4 * h + kk for s in range(4) for h in range(2) for kk in range(4))
    assert seen == list(range(32))


def test_three_bf16_terms_hold_an_fp32_value_exactly_and_the_x6_fragments_are_the_weights():
    """precision 'bf16x6' (csrc/cnn_bf16x6.hip): bf16_split(x, 3) is an EXACT split of any finite fp32 value (8 + 8 + 8 significant
    bits, each term rounded to nearest), two terms are not; the three-term fragment blob of pack_adapt_cnn_bf16 holds every
    BatchNorm-folded weight bit for bit at the layout offsets of layout.hpp (CNNX_*)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(20000) * 10.0 ** rng.uniform(-6, 4, 20000), [0.0, 1.0, -1.0, 65504.0, 1e-30, -3.0e38,
                        np.float32(1) + np.float32(2) ** -23]]).astype(np.float32)
    val = lambda b: (b.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    t3 = W.bf16_split(x, 3)
    assert np.array_equal((val(t3[0]) + val(t3[1]) + val(t3[2])).astype(np.float32), x)
    assert np.all(val(t3[0]) + val(t3[1]) + val(t3[2]) == x.astype(np.float64))          # exact even before the final rounding
    t2 = W.bf16_split(x, 2)
    assert np.abs(val(t2[0]) + val(t2[1]) - x).max() > 0                                # 16 bits are not enough
    nz = x != 0
    assert np.all(np.abs(val(t3[1])[nz]) <= 2.0 ** -8 * np.abs(x[nz]) * (1 + 2.0 ** -7))  # |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|
    assert np.all(np.abs(val(t3[2])[nz]) <= 2.0 ** -16 * np.abs(x[nz]) * (1 + 2.0 ** -6))
    # fragments: conv3 (32 -> 64 channels), step g = (tap, 16-channel group), N tile nt, term t, lane, k-slot e
    sd = synth.random_state_dict(7, 'NISQA_DIM')
    blob = W.pack_adapt_cnn_bf16(sd, conv1_pairs=True, terms=3)
    assert blob.size == W.CNNX_U16S and W.pack_adapt_cnn_bf16(sd, conv1_pairs=True).size == W.CNNB_U16S
    w3, _ = W.fold_bn(sd, 'cnn.model.', 3)
    fr = blob[W.CNNX_W3:W.CNNX_W4].reshape(18, 2, 3, 64, 8)
    for g, nt, lane, e in [(0, 0, 0, 0), (17, 1, 63, 7), (5, 1, 37, 3), (10, 0, 31, 4)]:
        n, c, tap = (lane & 31) + 32 * nt, 16 * (g % 2) + 8 * (lane >> 5) + e, g // 2
        want = np.float32(w3.astype(np.float32).reshape(64, 32, 9)[n, c, tap])
        got = sum(val(fr[g, nt, t, lane, e:e + 1])[0] for t in range(3))
        assert np.float32(got) == want and got == np.float64(want)


def test_two_f16_terms_hold_an_fp32_value_to_its_last_bit_or_one_ulp_and_the_f16_blob_carries_the_layer_constants():
    """precision 'f16x4' / 'f16x3' (csrc/cnn_bf16.hip, formats F16X3 / F16X4): f16_split(x, 2) of a value scaled into f16's normal range is
    the fp32 value itself or exactly one fp32 ulp off (an odd remainder beyond 2048 ulp32, about a quarter of random values); the CNNH
    blob holds W * 2^kw with the layer's largest weight in [2^14, 2^15) and, behind the fragments, kw / G = max_c sum |W_c| / T = max
    |shift| per layer -- the constants the kernel bounds every layer output with (|y| <= max|x| * G + T)."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-3, 3, 200000))).astype(np.float32)
    x = np.ldexp(x / np.abs(x).max(), 15).astype(np.float32)            # max in [2^14, 2^15) like every scaled tensor
    x = x[np.abs(x) >= 0.25]                                             # (below 2^-2 the low term is subnormal: absolute precision 2^-25)
    hi, lo = W.f16_split(x, 2)
    rec = W.f16_val(hi).astype(np.float64) + W.f16_val(lo).astype(np.float64)
    ulps = np.abs(rec - x.astype(np.float64)) / np.spacing(np.abs(x)).astype(np.float64)
    assert set(np.unique(ulps)) <= {0.0, 1.0} and 0.70 < (ulps == 0).mean() < 0.80
    small = np.float32(3e-4)                                              # subnormal low term: absolute error <= 2^-25
    h2, l2 = W.f16_split(np.array([small]), 2)
    assert abs(float(W.f16_val(h2)[0]) + float(W.f16_val(l2)[0]) - float(small)) <= 2.0 ** -25
    sd = synth.random_state_dict(7)
    blob = W.pack_adapt_cnn_f16(sd)
    assert blob.dtype == np.uint16 and blob.size == W.CNNH_U16S
    mi, mf = blob[W.CNNH_META:].view(np.int32), blob[W.CNNH_META:].view(np.float32)
    for l in range(1, 7):
        w, t = W.fold_bn(sd, 'cnn.model.', l)
        w32 = w.astype(np.float32)
        kw = int(mi[l - 1])
        assert 2.0 ** 14 <= np.abs(w32).max() * 2.0 ** kw < 2.0 ** 15
        assert mf[8 + l - 1] >= np.abs(w32.astype(np.float64)).reshape(w32.shape[0], -1).sum(1).max() and mf[8 + l - 1] < 1.0001 * np.abs(w).reshape(w.shape[0], -1).sum(1).max()
        assert mf[16 + l - 1] >= np.abs(t.astype(np.float32)).max()
    # conv3 / conv5 fragments: hi + lo = W * 2^kw to the pair's precision, in the two-term bf16 blob's layout
    w3, _ = W.fold_bn(sd, 'cnn.model.', 3)
    fr = blob[W.CNNB_W3:W.CNNB_W4].reshape(18, 2, 2, 64, 8)
    val = (W.f16_val(fr[:, :, 0]).astype(np.float64) + W.f16_val(fr[:, :, 1])) * 2.0 ** -int(mi[2])
    for g, nt, lane, e in [(0, 0, 0, 0), (17, 1, 63, 7), (9, 1, 37, 3)]:
        ref = np.float32(w3[(lane & 31) + 32 * nt, 16 * (g % 2) + 8 * (lane >> 5) + e, (g // 2) // 3, (g // 2) % 3])
        assert abs(val[g, nt, lane, e] - ref) <= max(np.spacing(np.abs(ref)), 2.0 ** (-25 - int(mi[2])))
    w5, _ = W.fold_bn(sd, 'cnn.model.', 5)
    fr = blob[W.CNNB_W5:W.CNNB_W6].reshape(4, 18, 2, 64, 8)
    val = (W.f16_val(fr[:, :, 0]).astype(np.float64) + W.f16_val(fr[:, :, 1])) * 2.0 ** -int(mi[4])
    for wv, g, lane, e in [(0, 0, 0, 0), (3, 17, 63, 7), (2, 5, 21, 2)]:
        ref = np.float32(w5[16 * wv + (lane & 15), 32 * (g & 1) + 8 * (lane >> 4) + e, (g >> 1) // 3, (g >> 1) % 3])
        assert abs(val[wv, g, lane, e] - ref) <= max(np.spacing(np.abs(ref)), 2.0 ** (-25 - int(mi[4])))


def test_three_term_linear_fragments_hold_the_weights_exactly():
    """td_bf16x6.hip: the chain-order fragments [K/16][rows/32][3][64][8] of a linear layer sum to the fp32 weight bit for bit at the
    k-slot map of the two-term form; blob sizes follow layout.hpp (TDX_*, PLX_*)."""
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((192, 64)) * 10.0 ** rng.uniform(-3, 1, (192, 64))).astype(np.float32)
    val = lambda b: (b.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    for chain in (False, True):
        fr3 = W.linear_a_fragments_bf16(w, chain=chain, terms=3).reshape(4, 6, 3, 64, 8)
        fr2 = W.linear_a_fragments_bf16(w, chain=chain).reshape(4, 6, 2, 64, 8)
        assert np.array_equal(fr3[:, :, 0], fr2[:, :, 0])                                    # the same leading term, the same slots
        total = val(fr3[:, :, 0]) + val(fr3[:, :, 1]) + val(fr3[:, :, 2])
        for s_, mt, lane, e in [(0, 0, 0, 0), (3, 5, 63, 7), (2, 1, 37, 5)]:
            col = 16 * s_ + ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) if chain else 8 * (lane >> 5) + e)
            assert total[s_, mt, lane, e] == np.float64(w[(lane & 31) + 32 * mt, col])
    # td16_bf16x6.hip: 16 x 32 fragments [rows/64][K/32][4][3][64][8] for 16-token tiles (lane = (i = l & 15, g = l >> 4))
    for chain in (False, True):
        fr = W.linear_a_fragments_bf16_t16(w, chain=chain).reshape(3, 2, 4, 3, 64, 8)
        total = val(fr[:, :, :, 0]) + val(fr[:, :, :, 1]) + val(fr[:, :, :, 2])
        for blk, s_, mt, lane, e in [(0, 0, 0, 0, 0), (2, 1, 3, 63, 7), (1, 0, 2, 37, 5), (1, 1, 1, 16, 4)]:
            g = lane >> 4
            col = 16 * (2 * s_ + (e >> 2)) + 4 * g + (e & 3) if chain else 32 * s_ + 8 * g + e
            assert total[blk, s_, mt, lane, e] == np.float64(w[64 * blk + 16 * mt + (lane & 15), col])
        cols = {(int(s_), int(16 * (2 * s_ + (e >> 2)) + 4 * g + (e & 3)) if chain else int(32 * s_ + 8 * g + e))
                for s_ in range(2) for g in range(4) for e in range(8)}
        assert len(cols) == 64 and {c_ for _, c_ in cols} == set(range(64))                # every column exactly once per row
    sd = synth.random_state_dict(7, 'NISQA_DIM')
    heads = ['pool_layers.%d.model.' % h for h in range(5)]
    # the pooling tail of td16_layer_kernel<true>: per head two 64-row blocks of 24 fragments, then one 1 KB float block per 64-row block
    t16 = W.pack_pool_att_t16(sd, heads)
    assert t16.size == 10 * 24 * 512 + 10 * 512
    par = t16[10 * 24 * 512:].view(np.float32).reshape(10, 256)
    for h in (0, 3):
        p_ = heads[h]
        for j in range(2):
            np.testing.assert_array_equal(par[2 * h + j, 0:64], sd[p_ + 'linear1.bias'].numpy()[64 * j:64 * j + 64])
            np.testing.assert_array_equal(par[2 * h + j, 64:128], sd[p_ + 'linear2.weight'].numpy().reshape(-1)[64 * j:64 * j + 64])
            np.testing.assert_array_equal(par[2 * h + j, 128:192], sd[p_ + 'linear3.weight'].numpy().reshape(-1))
            assert par[2 * h + j, 192] == sd[p_ + 'linear2.bias'].numpy().reshape(-1)[0] and par[2 * h + j, 193] == sd[p_ + 'linear3.bias'].numpy().reshape(-1)[0]
        fr = t16[:10 * 24 * 512].reshape(10, 2, 4, 3, 64, 8)          # [block][s][mt][term][lane][e]
        w1 = sd[p_ + 'linear1.weight'].numpy()
        tot = val(fr[:, :, :, 0]) + val(fr[:, :, :, 1]) + val(fr[:, :, :, 2])
        for j, s_, mt, lane, e in [(0, 0, 0, 0, 0), (1, 1, 3, 63, 7), (1, 0, 2, 37, 5)]:
            col = 16 * (2 * s_ + (e >> 2)) + 4 * (lane >> 4) + (e & 3)
            assert tot[2 * h + j, s_, mt, lane, e] == np.float64(w1[64 * j + 16 * mt + (lane & 15), col])
    assert W.pack_self_att_bf16(sd, 2, terms=3).size == W.TDX_LAYER0 + 2 * W.TDXL_U16S
    assert W.pack_self_att_bf16(sd, 2).size == W.TDB_LAYER0 + 2 * W.TDBL_U16S
    assert W.pack_pool_att_bf16(sd, heads, terms=3).size == 5 * W.PLX_U16S + 10 * (24 * 512 + 512) and W.pack_pool_att_bf16(sd, heads).size == 5 * W.PLB_U16S


def test_linear_fragments_roundtrip():
    rng = np.random.default_rng(1)
    w = rng.standard_normal((192, 64))
    fr = W.linear_a_fragments(w).reshape(8, 6, 64, 4)
    x = rng.standard_normal(64)
    y = np.zeros(192)
    for s in range(8):
        for mt in range(6):
            for lane in range(64):
                for kk in range(4):
                    y[(lane & 31) + 32 * mt] += fr[s, mt, lane, kk] * x[8 * s + 4 * (lane >> 5) + kk]
    np.testing.assert_allclose(y, w.astype(np.float32).astype(np.float64) @ x, atol=1e-5)


def test_bn_fold_matches_batchnorm():
    sd = synth.random_state_dict(3)
    wf, t = W.fold_bn(sd, 'cnn.model.', 2)
    x = torch.randn(2, 16, 8, 5)
    ref = torch.nn.functional.batch_norm(
        torch.nn.functional.conv2d(x, sd['cnn.model.conv2.weight'], sd['cnn.model.conv2.bias'], padding=1),
        sd['cnn.model.bn2.running_mean'], sd['cnn.model.bn2.running_var'], sd['cnn.model.bn2.weight'],
        sd['cnn.model.bn2.bias'], False, 0.0, 1e-5)
    got = torch.nn.functional.conv2d(x, torch.from_numpy(wf).float(), torch.from_numpy(t).float(), padding=1)
    assert (ref - got).abs().max() < 1e-5
    assert (sd['cnn.model.bn2.weight'] < 0).any()           # negative gammas are exercised


def test_pack_sizes_and_unsupported_geometry():
    sd = synth.random_state_dict(7)
    assert W.pack_adapt_cnn(sd).shape == (W.CNN_W_FLOATS,)
    assert W.pack_self_att(sd, 2).shape == (W.TD_LAYER0 + 2 * W.TDL_FLOATS,)
    assert W.pack_pool_att(sd, ['pool_layers.%d.model.' % h for h in range(5)]).shape == (5 * W.PL_FLOATS,)
    bad = dict(sd)
    bad['cnn.model.conv3.weight'] = torch.zeros(48, 32, 3, 3)
    with pytest.raises(NotImplementedError):
        W.pack_adapt_cnn(bad)


# ---- mel tables -------------------------------------------------------------------------------------
@pytest.mark.parametrize('sr', [48000, 44100, 16000, 8000])
def test_mel_tables_match_oracle_filterbank(sr):
    t = MelTables(sr, 4096, 0.01, 0.02, 48, 20000)
    fb = omel.mel_filterbank(sr, 4096, 48, 0.0, 20000.0)
    np.testing.assert_array_equal(t.dense, fb)
    assert t.hop == int(sr * 0.01) and t.win == int(sr * 0.02)
    for m in range(48):
        row = np.zeros(2049 + 400, np.float32)
        row[t.band_start[m]:t.band_start[m] + t.band_len[m]] = t.band_w[t.band_woff[m]:t.band_woff[m] + t.band_len[m]]
        np.testing.assert_array_equal(row[:2049], fb[m])
        assert t.band_len[m] % 16 == 0 and t.band_len[m] == t.band_len[4 * (m // 4)]
    np.testing.assert_allclose(t.window, omel.hann_periodic(t.win).astype(np.float32))
    assert t.n_bins == (np.nonzero(fb.any(0))[0].max() + 1)


def test_mel_tables_reject_unsupported():
    with pytest.raises(NotImplementedError):
        MelTables(48000, 2048, 0.01, 0.02, 48, 20000)
    with pytest.raises(NotImplementedError):
        MelTables(240000, 4096, 0.01, 0.02, 48, 20000)      # 4800-sample window > n_fft
    assert MelTables(96000, 4096, 0.01, 0.02, 48, 20000).win == 1920     # long windows are supported (NQ = 2)


# ---- plan ---------------------------------------------------------------------------------------------
def test_batch_plan_counts_and_errors():
    p = BatchPlan([480000, 14 * 480, 144000], 480, 4, 1300)
    assert list(p.T) == [1001, 15, 301]
    assert list(p.n_wins) == [247, 1, 72] == [onet.n_wins_of(int(t)) for t in p.T]
    assert list(p.tok_off) == [0, 256, 320, 448] and p.total_tok == 448              # whole 64-token workgroups per clip
    assert list(p.frame_off) == [0, 1001, 1016, 1317]
    assert len(p.token_index()) == 247 + 1 + 72
    with pytest.raises(ValueError, match='Sample too short'):
        BatchPlan([14 * 480 - 1], 480, 4, 1300, names=['x.wav'])
    with pytest.raises(ValueError, match='Increase max window length ms_max_segments'):
        BatchPlan([(1301 * 4 + 14) * 480], 480, 4, 1300)
    p1 = BatchPlan([20 * 480], 480, 1, 6000)                 # seg_hop 1 (nisqa_tts geometry)
    assert list(p1.n_wins) == [7]


# ---- WAV ingest -----------------------------------------------------------------------------------------
def _cases(tmp_path):
    rng = np.random.default_rng(4)
    mono16 = (rng.standard_normal(5000) * 3000).astype(np.int16)
    st16 = (rng.standard_normal((5000, 2)) * 3000).astype(np.int16)
    f32 = (rng.standard_normal((4000, 2)) * 0.2).astype(np.float32)
    i32 = (rng.standard_normal(3000) * 1e8).astype(np.int32)
    u8 = rng.integers(0, 255, 3000).astype(np.uint8)
    out = {}
    for name, data, sr in [('mono16', mono16, 48000), ('st16', st16, 44100), ('f32', f32, 16000),
                           ('i32', i32, 8000), ('u8', u8, 22050)]:
        p = str(tmp_path / (name + '.wav'))
        synth.write_wav(p, data, sr)
        out[name] = p
    return out


def test_wav_ingest_matches_oracle_decoder(tmp_path):
    files = _cases(tmp_path)
    for name, path in files.items():
        y, sr = wavio.read_wav(path)
        yo, sro = omel.load_wav(path)
        assert sr == sro
        if y.dtype == np.int16:
            assert name == 'mono16'
            y = y.astype(np.float32) / np.float32(32768.0)
        np.testing.assert_array_equal(y, yo)
    y, _ = wavio.read_wav(files['st16'], ms_channel=1)
    yo, _ = omel.load_wav(files['st16'], ms_channel=1)
    np.testing.assert_array_equal(y, yo)


def test_wav_24bit_and_errors(tmp_path):
    v = np.array([0, 1, -1, 8388607, -8388608, 123456], dtype=np.int32)
    raw = b''.join(int(x & 0xFFFFFF).to_bytes(3, 'little') for x in v)
    import struct
    hdr = b'RIFF' + struct.pack('<I', 36 + len(raw)) + b'WAVE' + b'fmt ' + struct.pack('<IHHIIHH', 16, 1, 1, 48000, 144000, 3, 24) \
        + b'data' + struct.pack('<I', len(raw))
    p = tmp_path / 'p24.wav'
    p.write_bytes(hdr + raw)
    y, sr = wavio.read_wav(str(p))
    np.testing.assert_allclose(y, v / 8388608.0, atol=1e-7)
    bad = tmp_path / 'bad.wav'
    bad.write_bytes(b'not a wav')
    with pytest.raises(ValueError, match='Could not load file'):
        wavio.read_wav(str(bad))
    with pytest.raises(ValueError, match='Could not load file'):
        wavio.read_wav(str(tmp_path / 'missing.wav'))


# ---- native ingest (include/nisqa_ingest.h) and the staging iterator ------------------------------------------
def _ingest_lib():
    from nisqa_amd import lib
    if not os.path.isfile(lib.INGEST_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return lib, lib.load_ingest()


def test_ingest_library_exports_every_declared_symbol():
    lib, L = _ingest_lib()
    hdr = open(os.path.join(ROOT, 'include', 'nisqa_ingest.h')).read()
    declared = set(re.findall(r'^\s*int\s+(nisqa_ingest_[a-z0-9_]+)\s*\(', hdr, re.M))
    assert declared == set(lib.INGEST_SYMBOLS), declared ^ set(lib.INGEST_SYMBOLS)
    out = subprocess.check_output(['nm', '-D', '--defined-only', lib.INGEST_PATH]).decode()
    assert declared <= set(re.findall(r' T (nisqa_\w+)', out))
    assert L.nisqa_ingest_abi_version() == 2 and ctypes.sizeof(lib.WavInfo) == 40


def _riff(fmt_body, data, extra_before=b'', data_size=None, riff=b'RIFF'):
    import struct
    chunks = b'fmt ' + struct.pack('<I', len(fmt_body)) + fmt_body + extra_before \
        + b'data' + struct.pack('<I', len(data) if data_size is None else data_size) + data
    return riff + struct.pack('<I', 4 + len(chunks)) + b'WAVE' + chunks


def test_native_probe_and_read_match_the_python_decoder(tmp_path):
    import struct
    lib, L = _ingest_lib()
    files = _cases(tmp_path)
    rng = np.random.default_rng(5)
    pcm = (rng.standard_normal(3001) * 2000).astype(np.int16)
    fmt16 = struct.pack('<HHIIHH', 1, 1, 32000, 64000, 2, 16)
    odd = b'LIST' + struct.pack('<I', 5) + b'abcde' + b'\0'                  # odd-sized chunk + pad byte before data
    big = b'junk' + struct.pack('<I', 6000) + bytes(6000)                     # pushes 'data' past the 4 KiB header read
    ext = struct.pack('<HHIIHHHHIH', 0xFFFE, 1, 32000, 64000, 2, 16, 22, 16, 4, 1) + bytes(14)   # WAVE_FORMAT_EXTENSIBLE
    special = {
        'odd': _riff(fmt16, pcm.tobytes(), odd),
        'far': _riff(fmt16, pcm.tobytes(), big),
        'ext': _riff(ext, pcm.tobytes()),
        'stream': _riff(fmt16, pcm.tobytes(), data_size=0xFFFFFFFF),          # size unknown: to end of file
        'trunc': _riff(fmt16, pcm.tobytes()[:4001], data_size=len(pcm) * 2),  # data size overruns the file
    }
    for k, blob in special.items():
        (tmp_path / (k + '.wav')).write_bytes(blob)
        files[k] = str(tmp_path / (k + '.wav'))
    names = sorted(files)
    n = len(names)
    paths = (ctypes.c_char_p * n)(*[os.fsencode(files[k]) for k in names])
    infos = (lib.WavInfo * n)()
    assert L.nisqa_ingest_probe(paths, n, infos, 4) == 0
    off, total = [], 0
    for i in range(n):
        off.append(total)
        total += infos[i].n_frames * infos[i].block_align
    dst = np.zeros(total, np.uint8)
    assert L.nisqa_ingest_read(paths, n, infos, dst.ctypes.data, (ctypes.c_int64 * n)(*off), 3) == 0
    for i, k in enumerate(names):
        h = wavio.probe(files[k])
        try:
            assert (infos[i].tag, infos[i].channels, infos[i].bits, infos[i].block_align, infos[i].sample_rate,
                    infos[i].data_offset, infos[i].n_frames) == (h.tag, h.ch, h.bits, h.blk, h.sr, h.data_off, h.n), k
            want = bytearray(h.n * h.blk)
            wavio.read_data_into(h, want)
        finally:
            h.close()
        assert dst[off[i]:off[i] + len(want)].tobytes() == bytes(want), k
    for k in ('odd', 'far', 'ext', 'stream'):
        y, sr = wavio.read_wav(files[k])
        assert sr == 32000 and np.array_equal(y, pcm), k
    assert len(wavio.read_wav(files['trunc'])[0]) == 2000


def test_native_probe_reports_unusable_files(tmp_path):
    import struct
    lib, L = _ingest_lib()
    fmt16 = struct.pack('<HHIIHH', 1, 1, 32000, 64000, 2, 16)
    cases = {
        'notriff': b'not a wav at all, just text',
        'nodata': b'RIFF' + struct.pack('<I', 28) + b'WAVE' + b'fmt ' + struct.pack('<I', 16) + fmt16,
        'badalign': _riff(struct.pack('<HHIIHH', 1, 2, 32000, 64000, 2, 16), bytes(64)),
        'adpcm': _riff(struct.pack('<HHIIHH', 2, 1, 32000, 16000, 1, 4), bytes(64)),
        'datafirst': b'RIFF' + struct.pack('<I', 20) + b'WAVE' + b'data' + struct.pack('<I', 8) + bytes(8),
        'empty': b'',
    }
    names = sorted(cases) + ['missing']
    for k, blob in cases.items():
        (tmp_path / (k + '.wav')).write_bytes(blob)
    n = len(names)
    paths = (ctypes.c_char_p * n)(*[os.fsencode(str(tmp_path / (k + '.wav'))) for k in names])
    infos = (lib.WavInfo * n)()
    assert L.nisqa_ingest_probe(paths, n, infos, 2) == n
    for i, k in enumerate(names):
        assert infos[i].status == (lib.WAV_ERR_OPEN if k == 'missing' else lib.WAV_ERR_FORMAT), k
        with pytest.raises(ValueError, match='Could not load file'):
            wavio.read_wav(str(tmp_path / (k + '.wav')))
    assert L.nisqa_ingest_probe(None, 0, None, 1) == 0 and L.nisqa_ingest_probe(None, 3, None, 1) == -1


class _ListDataset(object):
    ms_channel = None

    def __init__(self, paths):
        self.paths = paths

    def file_path(self, i):
        return self.paths[i]


@pytest.mark.parametrize('workers', [0, 5])
def test_staging_iterator_lays_batches_out_like_concatenated_read_wav(tmp_path, workers):
    from nisqa_amd import ingest
    files = _cases(tmp_path)
    rng = np.random.default_rng(6)
    paths = []
    for i in range(11):                                            # mono PCM16 at two rates, ragged lengths
        p = str(tmp_path / ('m%02d.wav' % i))
        synth.write_wav(p, (rng.standard_normal(700 + 37 * i) * 1000).astype(np.int16), 48000 if i % 3 else 16000)
        paths.append(p)
    paths[4:4] = [files['st16'], files['f32'], files['u8']]         # 44.1 k stereo, 16 k float stereo, 22.05 k 8-bit
    ds = _ListDataset(paths)
    batches = [list(range(s, min(s + 4, len(paths)))) for s in range(0, len(paths), 4)] * 2   # slots are recycled
    ing = ingest.Ingest(ds, batches, pin=False, num_workers=workers)
    seen = []
    try:
        for staged in ing:
            raw = ing.ring.buf[staged.slot]
            got = []
            for g in staged.groups:
                host = raw[g.offset:g.offset + g.nbytes].view(torch.int16 if g.is_i16 else torch.float32).numpy()
                ref = [wavio.read_wav(paths[i]) for i in g.ids]
                assert all(sr == g.sr for _, sr in ref) and g.lengths == [len(y) for y, _ in ref]
                assert g.is_i16 == all(y.dtype == np.int16 for y, _ in ref)
                want = np.concatenate([y if g.is_i16 or y.dtype != np.int16 else y.astype(np.float32) / np.float32(32768.0)
                                       for y, _ in ref])
                np.testing.assert_array_equal(host, want)
                got += g.ids
            seen.append(sorted(got))
            ing.ring.release_after(staged.slot, None)
    finally:
        ing.close()
    assert seen == [sorted(b) for b in batches]


def test_staging_iterator_surfaces_load_errors_in_order(tmp_path):
    from nisqa_amd import ingest
    good = str(tmp_path / 'g.wav')
    synth.write_wav(good, np.zeros(500, np.int16), 48000)
    ds = _ListDataset([good, good, str(tmp_path / 'gone.wav'), good])
    ing = ingest.Ingest(ds, [[0, 1], [2, 3]], pin=False, num_workers=2)
    try:
        it = iter(ing)
        first = next(it)
        assert [g.ids for g in first.groups] == [[0, 1]]
        ing.ring.release_after(first.slot, None)
        with pytest.raises(ValueError, match='Could not load file .*gone.wav'):
            next(it)
    finally:
        ing.close()


# ---- C ABI ------------------------------------------------------------------------------------------------
def test_struct_layouts_match_the_c_headers(tmp_path):
    """The ctypes mirrors in nisqa_amd/lib.py against the structs a C compiler sees in include/*.h: size and every
    field offset (a maintainer binding the C ABI from another language gets the same numbers from the header)."""
    import subprocess
    from nisqa_amd import lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {'nisqa_mel_cfg': lib.MelCfg, 'nisqa_model_dev': lib.ModelDev, 'nisqa_wav_info': lib.WavInfo}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "nisqa_hip.h"', '#include "nisqa_ingest.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c99', '-I', os.path.join(root, 'include'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split('\n')
    seen = 0
    for line in out:
        if not line.strip():
            continue
        cname, fname, val = line.split()
        cls = structs[cname]
        if fname == 'size':
            assert ctypes.sizeof(cls) == int(val), (cname, ctypes.sizeof(cls), val)
        else:
            assert getattr(cls, fname).offset == int(val), (cname, fname, getattr(cls, fname).offset, val)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in structs.values())


def test_library_exports_every_declared_symbol():
    from nisqa_amd import lib
    hdr = open(os.path.join(ROOT, 'include', 'nisqa_hip.h')).read()
    declared = set(re.findall(r'\b(nisqa_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    if not os.path.isfile(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = lib.load()                                           # loads, binds every symbol, checks the ABI version
    assert L.nisqa_abi_version() == 2
    assert L.nisqa_workspace_bytes(64, 64064, 16384) > 64064 * 48 * 4
    assert L.nisqa_workspace_bytes(0, 1, 1) == 0
    out = subprocess.check_output(['nm', '-D', '--defined-only', lib.LIB_PATH]).decode()
    exported = set(re.findall(r' T (nisqa_\w+)', out))
    assert declared <= exported


def test_product_library_exports_no_debug_symbol_and_an_instrumented_one_is_refused(tmp_path):
    """experiment scaffolding lives in csrc/experimental.hpp, which only -DNQ_EXPERIMENTAL builds include; such a
    build exports nisqa_debug_* readers and lib.load() refuses it unless NISQA_ALLOW_DEBUG_LIB=1."""
    from nisqa_amd import lib
    assert lib.exported_symbols(lib.LIB_PATH, 'nisqa_debug_') == []
    mine = set(lib.exported_symbols(lib.LIB_PATH, 'nisqa_'))
    nm = subprocess.check_output(['nm', '-D', '--defined-only', lib.LIB_PATH]).decode()
    assert mine == set(re.findall(r' T (nisqa_\w+)', nm))                       # the ELF reader agrees with binutils
    csrc = os.path.join(ROOT, 'nisqa_amd', 'csrc')
    for f in sorted(os.listdir(csrc)):                                          # no experiment #ifdef in any product kernel source
        if f.endswith('.hip'):
            src = open(os.path.join(csrc, f)).read()
            assert not re.search(r'#\s*if(def|ndef)?\s+\(?\s*(defined\s*\(\s*)?(NQ_|SC_)', src), f
            assert 'experimental.hpp' not in src, f
    hpp = open(os.path.join(csrc, 'common.hpp')).read()
    assert re.search(r'#ifdef NQ_EXPERIMENTAL\n#include "experimental.hpp"', hpp)
    # an instrumented stand-in: a shared object that defines one nisqa_debug_* symbol (what -DNQ_EXPERIMENTAL adds to the real one)
    stub = tmp_path / 'libdbg.so'
    (tmp_path / 'dbg.c').write_text('int nisqa_debug_phase_clock6(unsigned long long* o, int r) { (void)o; (void)r; return 0; }\n'
                                    'int nisqa_abi_version(void) { return 2; }\n')
    subprocess.check_call(['gcc', '-shared', '-fPIC', '-o', str(stub), str(tmp_path / 'dbg.c')])
    assert lib.exported_symbols(str(stub), 'nisqa_debug_') == ['nisqa_debug_phase_clock6']
    code = ('import os, sys; sys.path.insert(0, %r); from nisqa_amd import lib\n'
            'try:\n    lib.load(); print("LOADED")\n'
            'except RuntimeError as e:\n    print("REFUSED" if "instrumented" in str(e) else "OTHER " + str(e))\n'
            'except AttributeError as e:\n    print("PAST-THE-GATE")\n' % ROOT)
    env = dict(os.environ, NISQA_HIP_LIB=str(stub))
    env.pop('NISQA_ALLOW_DEBUG_LIB', None)
    assert subprocess.check_output([sys.executable, '-c', code], env=env).decode().strip().endswith('REFUSED')
    env['NISQA_ALLOW_DEBUG_LIB'] = '1'      # the gate opens (the stub then fails on its missing product symbols, which is fine)
    assert subprocess.check_output([sys.executable, '-c', code], env=env).decode().strip().endswith('PAST-THE-GATE')


def test_training_operators_are_declared_and_exported():
    from nisqa_amd import lib
    hdr = open(os.path.join(ROOT, 'include', 'nisqa_train.h')).read()
    declared = set(re.findall(r'^\s*(?:int|int64_t)\s+(nisqa_[a-z0-9_]+)\s*\(', hdr, re.M))
    assert declared == set(lib.TRAIN_SYMBOLS), declared ^ set(lib.TRAIN_SYMBOLS)
    L = lib.load()
    out = subprocess.check_output(['nm', '-D', '--defined-only', lib.LIB_PATH]).decode()
    assert declared <= set(re.findall(r' T (nisqa_\w+)', out))
    assert L.nisqa_gemm_f32_one(None, None, None, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1.0, None, 0, None) == lib.NISQA_ERR_ARG
    assert L.nisqa_adam_step(None, None, None, None, 10, 1e-3, 1, None) == lib.NISQA_ERR_ARG
    assert L.nisqa_elementwise(9, None, None, None, 1, 1, None, None) == lib.NISQA_ERR_ARG


def test_abi_argument_validation_without_gpu():
    from nisqa_amd import lib
    L = lib.load()
    cfg = lib.MelCfg(2048, 480, 960, 48, 1707, 4032, 1e-8, 80.0)      # wrong n_fft -> rejected before any launch
    assert L.nisqa_mel_db(None, None, None, 1, 10, cfg, None, None, None, None, None, None, None, None, None) == lib.NISQA_ERR_ARG
    assert L.nisqa_cnn_adapt(None, None, None, None, None, 1, 33, 4, None, None, None, None) == lib.NISQA_ERR_ARG
    assert L.nisqa_pool_att(None, None, None, 1, 32, 9, None, None, None, None) == lib.NISQA_ERR_ARG


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from nisqa_amd.engine import HipNisqa
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7))


# ---- nisqaModel / CLI plumbing -------------------------------------------------------------------------------
class FakeEngine(object):
    """Test double for the device engine: same interface, deterministic per-clip numbers on the CPU."""

    def __init__(self, heads):
        self.n_heads, self.device = heads, torch.device('cpu')

    def plan(self, lengths, sr, names=None):
        return BatchPlan(lengths, int(sr * 0.01), 4, 1300, names)

    def audio_plan(self, lengths, sr, names=None):                  # (ms_sr = None: the files' own rate)
        return self.plan(lengths, sr, names)

    def forward_audio(self, pcm, lengths, sr, plan):
        return self.forward_pcm(pcm, plan, sr)

    def resample(self, pcm, lengths, sr):
        return pcm

    def rate(self, sr):
        return int(sr)

    def forward_pcm(self, pcm, plan, sr):
        if pcm.dtype == torch.int16:
            pcm = pcm.to(torch.float32) / 32768.0
        rows = []
        for b in range(plan.n_clips):
            seg = pcm[plan.clip_off[b]:plan.clip_off[b + 1]]
            rows.append(torch.stack([seg.abs().mean() * (h + 1) + plan.n_wins[b] for h in range(self.n_heads)]))
        return torch.stack(rows).float()


@pytest.fixture()
def wav_dir(tmp_path):
    d = tmp_path / 'wavs'
    d.mkdir()
    for i in range(5):
        synth.write_wav(str(d / ('c%d.wav' % i)), synth.synth_pcm16(i, 0.3 + 0.1 * i), 48000)
    pd.DataFrame({'name': ['c%d.wav' % i for i in (3, 1, 4)], 'db': ['x', 'y', 'z']}).to_csv(d / 'list.csv', index=False)
    return d


def _ckpt(tmp_path, model='NISQA_DIM'):
    args = dict(synth.DIM_ARGS if model == 'NISQA_DIM' else synth.MOS_ARGS)
    args.update({'pretrained_model': False, 'tr_bs_val': 1, 'tr_num_workers': 0, 'tr_device': 'cuda'})
    p = tmp_path / ('%s.tar' % model)
    torch.save({'args': args, 'model_state_dict': synth.random_state_dict(7 if model == 'NISQA_DIM' else 8, model)}, p)
    return str(p)


def _args(mode, ckpt, **kw):
    a = {'mode': mode, 'pretrained_model': ckpt, 'deg': None, 'data_dir': None, 'output_dir': None, 'csv_file': None,
         'csv_deg': None, 'num_workers': 0, 'bs': 2, 'ms_channel': None, 'tr_bs_val': 2, 'tr_num_workers': 0}
    a.update(kw)
    return a


def test_nisqa_model_predict_dir_and_csv_plumbing(tmp_path, wav_dir, capsys):
    from nisqa_amd.NISQA_model import nisqaModel
    ck = _ckpt(tmp_path)
    out_dir = tmp_path / 'out'
    out_dir.mkdir()
    m = nisqaModel(_args('predict_dir', ck, data_dir=str(wav_dir), output_dir=str(out_dir)))
    assert m.args['dim'] is True and m.args['ms_seg_hop_length'] == 4 and m.args['name'] == 'rand_dim'
    assert len(m.ds_val) == 5 and list(m.ds_val.df.columns) == ['deg']
    m.model._engine = FakeEngine(5)
    import sys as _sys
    seen_interval = []
    real_forward = m.model._engine.forward_audio if hasattr(m.model._engine, 'forward_audio') else None
    before = _sys.getswitchinterval()
    if real_forward is not None:                                     # inside the loop the interpreter's switch interval is short ...
        def spy(*a, **k):
            seen_interval.append(_sys.getswitchinterval())
            return real_forward(*a, **k)
        m.model._engine.forward_audio = spy
    df = m.predict()
    assert _sys.getswitchinterval() == before                        # ... and restored behind it (NISQA_lib._predict)
    assert not seen_interval or max(seen_interval) <= 1.0001e-4
    assert list(df.columns) == ['deg', 'mos_pred', 'noi_pred', 'dis_pred', 'col_pred', 'loud_pred', 'model']
    csv = pd.read_csv(out_dir / 'NISQA_results.csv')
    assert list(csv.columns) == list(df.columns) and len(csv) == 5 and (csv['model'] == 'rand_dim').all()
    txt = capsys.readouterr().out
    assert 'Model architecture: NISQA_DIM' in txt and '---> Predicting ...' in txt and '# files: 5' in txt
    # predict_csv keeps the caller's columns and CSV order
    m2 = nisqaModel(_args('predict_csv', ck, data_dir=str(wav_dir), csv_file='list.csv', csv_deg='name'))
    m2.model._engine = FakeEngine(5)
    df2 = m2.predict()
    assert list(df2['name']) == ['c3.wav', 'c1.wav', 'c4.wav'] and list(df2['db']) == ['x', 'y', 'z']
    by_name = dict(zip(df['deg'], df['mos_pred']))
    assert [by_name[n] for n in df2['name']] == pytest.approx(list(df2['mos_pred']))
    assert 'model' not in df2.columns                        # only written when output_dir is set (NISQA_model.py:74-75)


def test_nisqa_model_predict_file_mos_only_and_errors(tmp_path, wav_dir):
    from nisqa_amd.NISQA_model import nisqaModel
    ck = _ckpt(tmp_path, 'NISQA')
    m = nisqaModel(_args('predict_file', ck, deg=str(wav_dir / 'c2.wav')))
    assert m.args['dim'] is False and len(m.ds_val) == 1
    m.model._engine = FakeEngine(1)
    df = m.predict()
    assert list(df.columns) == ['deg', 'mos_pred'] and df['mos_pred'].dtype == np.float64   # NISQA_lib.py:1438
    empty = tmp_path / 'empty'
    empty.mkdir()
    with pytest.raises(ValueError, match='No wav files found in data_dir'):
        nisqaModel(_args('predict_dir', ck, data_dir=str(empty)))
    with pytest.raises(NotImplementedError):
        nisqaModel(_args('nope', ck))
    short = tmp_path / 'short.wav'
    synth.write_wav(str(short), synth.synth_pcm16(1, 0.05), 48000)
    ms = nisqaModel(_args('predict_file', ck, deg=str(short)))
    ms.model._engine = FakeEngine(1)
    with pytest.raises(ValueError, match='Sample too short'):
        ms.predict()


def test_nisqa_model_evaluate_after_predict_csv(tmp_path, wav_dir, capsys):
    """run_evaluate.py's sequence: predict_csv with csv_con, then evaluate() (reference NISQA_model.py:48-52, 572-716)."""
    from nisqa_amd.NISQA_model import nisqaModel
    rng = np.random.default_rng(3)
    rows = []
    for db in ('DB2', 'DB1'):
        for con in (1, 2, 3):
            for k in range(2):
                rows.append({'name': 'c%d.wav' % ((con + k) % 5), 'db': db, 'con': con,
                             **{t: 1 + 4 * rng.random() for t in ('mos', 'noi', 'dis', 'col', 'loud')}})
    dfile = pd.DataFrame(rows)
    dfile.to_csv(wav_dir / 'files.csv', index=False)
    dcon = dfile.groupby(['db', 'con'], as_index=False)[['mos', 'noi', 'dis', 'col', 'loud']].mean()
    for t in ('mos', 'noi', 'dis', 'col', 'loud'):
        dcon[t + '_ci'] = 0.2
    dcon.to_csv(wav_dir / 'cons.csv', index=False)
    m = nisqaModel(_args('predict_csv', _ckpt(tmp_path), data_dir=str(wav_dir), csv_file='files.csv', csv_deg='name',
                         csv_con='cons.csv'))
    m.model._engine = FakeEngine(5)
    m.predict()
    capsys.readouterr()
    m.evaluate(mapping='first_order', do_print=True, do_plot=False)
    out = capsys.readouterr().out
    assert [l for l in out.splitlines() if l.startswith('-->')] == ['--> MOS:', '--> NOI:', '--> DIS:', '--> COL:', '--> LOUD:']
    assert 'DB1:' in out and 'rmse_star_map_con' in out and 'Average over MOS and dimensions: r_p=' in out
    assert list(m.db_results_val_mos['db']) == ['DB1', 'DB2'] and 'r_p_mean_con_loud' in m.r and 'rmse_all' in m.r
    assert 'y_hat_map' in m.ds_val.df
    # MOS-only model, no per-condition file
    m1 = nisqaModel(_args('predict_csv', _ckpt(tmp_path, 'NISQA'), data_dir=str(wav_dir), csv_file='files.csv',
                          csv_deg='name'))
    m1.model._engine = FakeEngine(1)
    m1.predict()
    capsys.readouterr()
    m1.evaluate()
    out = capsys.readouterr().out
    assert out.startswith('--> MOS:') and 'r_p_mean_file' in out and np.isnan(m1.r['r_p_mean_con'])
    with pytest.raises(NotImplementedError):
        m1.train()


def test_predict_without_gpu_raises_no_cpu_path(tmp_path, wav_dir):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from nisqa_amd.NISQA_model import nisqaModel
    m = nisqaModel(_args('predict_dir', _ckpt(tmp_path), data_dir=str(wav_dir)))
    with pytest.raises(RuntimeError, match='no CPU'):
        m.predict()


def test_cli_argument_errors():
    sys.path.insert(0, ROOT)
    import run_predict
    with pytest.raises(ValueError, match='--deg argument'):
        run_predict.build_args(['--mode', 'predict_file', '--pretrained_model', 'x.tar'])
    with pytest.raises(ValueError, match='--data_dir argument'):
        run_predict.build_args(['--mode', 'predict_dir', '--pretrained_model', 'x.tar'])
    with pytest.raises(ValueError, match='--csv_deg argument'):
        run_predict.build_args(['--mode', 'predict_csv', '--pretrained_model', 'x.tar', '--csv_file', 'a.csv'])
    with pytest.raises(NotImplementedError):
        run_predict.build_args(['--mode', 'train', '--pretrained_model', 'x.tar'])
    a = run_predict.build_args(['--mode', 'predict_csv', '--pretrained_model', 'x.tar', '--csv_file', 'a.csv',
                                '--csv_deg', 'f', '--bs', '7', '--num_workers', '3'])
    assert a['data_dir'] == '' and a['tr_bs_val'] == 7 and a['tr_num_workers'] == 3


def test_unsupported_architecture_is_rejected(tmp_path):
    from nisqa_amd import NISQA_lib as NL
    with pytest.raises(NotImplementedError, match='CNN-SA-AP'):
        NL.NISQA(cnn_model='adapt', td='lstm', pool='avg')                     # config/train_nisqa_cnn_lstm_avg.yaml
    with pytest.raises(NotImplementedError):
        NL.NISQA_DIM(cnn_model='standard', td='lstm', pool='last_step_bi')     # tts architecture is MOS-only
    m = NL.NISQA(cnn_model='standard', td='lstm', pool='last_step_bi', cnn_fc_out_h=20, td_lstm_h=128)
    assert 'time_dependency.model.lstm.weight_hh_l0_reverse' in m.state_dict()


def test_tts_packing_layouts():
    sd = synth.random_state_dict(9, 'NISQA_TTS')
    blob = W.pack_standard_cnn(sd)
    assert blob.shape == (W.CNNS_W_FLOATS,)
    fc = sd['cnn.model.fc_out.weight'].numpy()
    for (j, c, pix) in [(0, 0, 0), (19, 63, 11), (7, 13, 5)]:
        assert blob[W.CNNS_FC_W + (pix * 64 + c) * 20 + j] == fc[j, c * 12 + pix]
    lw = W.pack_lstm_laststep(sd)
    assert lw.shape == (W.LSTM_W_FLOATS,)
    whh_r = sd['time_dependency.model.lstm.weight_hh_l0_reverse'].numpy()
    assert lw[W.LSTM_DIR_FLOATS + W.LSTM_WHH + 300 * 128 + 17] == whh_r[300, 17]
    b = (sd['time_dependency.model.lstm.bias_ih_l0'] + sd['time_dependency.model.lstm.bias_hh_l0']).numpy()
    np.testing.assert_allclose(lw[W.LSTM_B:W.LSTM_B + 512], b, rtol=1e-6)
    assert lw[W.LSTM_POOL_W + 256] == sd['pool.model.linear.bias'].numpy()[0]


# ---- multi-process clip sharding (gloo, world_size 2) ------------------------------------------------------------
def test_shard_bounds_cover_everything():
    from nisqa_amd import dist
    for n in (1, 2, 7, 64, 100000):
        for w in (1, 2, 3, 8):
            b = [dist.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def test_balanced_bounds_equalise_work_on_length_sorted_lists():
    """Work-balanced contiguous shards: on a length-sorted list of mixed 3-30 s clips equal COUNTS give the last
    rank several times the first rank's segments; the prefix-sum boundaries keep every rank within 10 % of the mean."""
    from nisqa_amd import dist
    rng = np.random.default_rng(7)
    tok = np.sort(np.maximum(1, -(-(1 + (rng.uniform(3, 30, 100000) * 48000).astype(np.int64) // 480 - 14) // 4)))
    for w in (2, 3, 4, 8):
        b = dist.balanced_bounds(tok, w)
        assert b[0][0] == 0 and b[-1][1] == len(tok) and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        work = np.array([tok[lo:hi].sum() for lo, hi in b], dtype=np.float64)
        assert np.abs(work / work.mean() - 1).max() < 0.01, work
        by_count = np.array([tok[slice(*dist.shard_bounds(len(tok), r, w))].sum() for r in range(w)], dtype=np.float64)
        assert by_count.max() / by_count.min() > (1.5 if w == 2 else 2.5)                   # what it replaces
    # degenerate inputs: nothing to balance -> the count shards; fewer items than ranks -> empty shards, still a partition
    assert dist.balanced_bounds(np.zeros(7), 3) == [dist.shard_bounds(7, r, 3) for r in range(3)]
    assert dist.balanced_bounds([], 2) == [(0, 0), (0, 0)]
    b = dist.balanced_bounds([5, 1], 4)
    assert b[0][0] == 0 and b[-1][1] == 2 and all(b[i][1] == b[i + 1][0] and b[i][0] <= b[i][1] for i in range(3))
    b = dist.balanced_bounds([1, 1, 1, 100], 2)
    assert b == [(0, 3), (3, 4)]


_WORKER_BAL = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch, pandas as pd
import test_host as T
from nisqa_amd import NISQA_lib as NL, synth
rank = int(sys.argv[1]); world = int(sys.argv[2])
torch.distributed.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=rank, world_size=world)
files = %(files)r
ds = NL.SpeechQualityDataset(pd.DataFrame(files, columns=['deg']), data_dir=%(wavs)r, filename_column='deg',
                             mos_column='predict_only', dim=True, seg_length=15, seg_hop_length=4, ms_hop_length=0.01)
model = NL.NISQA_DIM(**{k: v for k, v in synth.DIM_ARGS.items() if k.startswith(('cnn_', 'td', 'pool', 'ms_seg_length', 'ms_n_mels'))})
seen = []
class Eng(T.FakeEngine):
    def forward_pcm(self, pcm, plan, sr):
        seen.append(int(np.sum(plan.n_wins)))
        return super().forward_pcm(pcm, plan, sr)
model._engine = Eng(5)
y, _ = NL.predict_dim(model, ds, 2, 'cpu', 0)
json.dump({'y': y.tolist(), 'tokens': int(sum(seen))}, open(os.path.join(%(out)r, 'r%%d.json' %% rank), 'w'))
torch.distributed.destroy_process_group()
'''


@pytest.mark.parametrize('world', [2, 3])
def test_predict_loop_shards_by_work_on_a_length_sorted_list(tmp_path, world):
    """gloo, world 2 and 3, a length-sorted list of 0.2 ... 2.4 s clips: every rank's segment count (what its engine was
    handed) is within 10 % of the mean, every rank ends with the full frame in INPUT order, equal to a one-process run."""
    import json
    import socket
    d = tmp_path / 'w'
    d.mkdir()
    durs = np.sort(np.concatenate((np.random.default_rng(3).uniform(0.2, 2.4, 44), [0.2, 2.4])))
    names = _mixed_wavs(d, durs.tolist())
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER_BAL % {'root': ROOT, 'port': port, 'wavs': str(d), 'out': str(tmp_path), 'files': names})
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world)], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [json.load(open(tmp_path / ('r%d.json' % r))) for r in range(world)]
    assert all(r['y'] == res[0]['y'] for r in res)
    tok = np.array([r['tokens'] for r in res], dtype=np.float64)
    assert np.abs(tok / tok.mean() - 1).max() < 0.10, tok
    from nisqa_amd import NISQA_lib as NL
    ds = NL.SpeechQualityDataset(pd.DataFrame(names, columns=['deg']), data_dir=str(d), filename_column='deg',
                                 mos_column='predict_only', dim=True, seg_length=15, seg_hop_length=4, ms_hop_length=0.01)
    model = NL.NISQA_DIM(**{k: v for k, v in synth.DIM_ARGS.items()
                            if k.startswith(('cnn_', 'td', 'pool', 'ms_seg_length', 'ms_n_mels'))})
    model._engine = FakeEngine(5)
    y, _ = NL.predict_dim(model, ds, 2, 'cpu', 0)
    np.testing.assert_allclose(np.array(res[0]['y']), y, rtol=0, atol=1e-6)
    assert (np.diff(y[:, 0]) != 0).any()


_WORKER_FAIL = r"""
import os, sys, json, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch, pandas as pd
import test_host as T
from nisqa_amd import NISQA_lib as NL, synth
rank = int(sys.argv[1]); world = int(sys.argv[2])
torch.distributed.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=rank, world_size=world)
files = %(files)r
ds = NL.SpeechQualityDataset(pd.DataFrame(files, columns=['deg']), data_dir=%(wavs)r, filename_column='deg',
                             mos_column='predict_only', dim=True, seg_length=15, seg_hop_length=4, ms_hop_length=0.01)
model = NL.NISQA_DIM(**{k: v for k, v in synth.DIM_ARGS.items() if k.startswith(('cnn_', 'td', 'pool', 'ms_seg_length', 'ms_n_mels'))})
model._engine = T.FakeEngine(5)
t0 = time.time()
res = {'raised': None}
try:
    NL.predict_dim(model, ds, 2, 'cpu', 0)
except Exception as e:
    res = {'raised': type(e).__name__, 'msg': str(e), 'seconds': time.time() - t0}
json.dump(res, open(os.path.join(%(out)r, 'r%%d.json' %% rank), 'w'))
torch.distributed.destroy_process_group()
"""


def _run_ranks(tmp_path, script_text, world, timeout=300):
    import json
    script = tmp_path / 'worker.py'
    script.write_text(script_text)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world)], env=env) for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=timeout) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [json.load(open(tmp_path / ('r%d.json' % r))) for r in range(world)]


def _free_port():
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    return port


@pytest.mark.parametrize('bad', ['unreadable', 'too_short', 'too_long'])
def test_a_bad_file_in_one_ranks_shard_fails_every_rank_with_the_reference_error(tmp_path, bad):
    """the rank whose shard holds an unreadable / too-short (0.1 s: 11 frames < 15) / too-long (60 s: 1497
    segments > 1300) clip raises the reference's ValueError (NL:2305-2306, 2259-2263, 2276-2277); the other rank used to
    block in the closing all_gather.  Now both ranks raise, within seconds, with the message one process gives."""
    d = tmp_path / 'w'
    d.mkdir()
    names = _mixed_wavs(d, [0.4] * 6)
    if bad == 'unreadable':
        (d / 'bad.wav').write_bytes(b'not a wav file at all')
        match = 'Could not load file'
    elif bad == 'too_short':
        synth.write_wav(str(d / 'bad.wav'), synth.synth_pcm16(1, 0.1), 48000)
        match = 'Sample too short'
    else:
        synth.write_wav(str(d / 'bad.wav'), np.zeros(60 * 48000, np.int16), 48000)
        match = 'Increase max window length'
    names = names + ['bad.wav']                           # last item: rank 1's shard under any split
    res = _run_ranks(tmp_path, _WORKER_FAIL % {'root': ROOT, 'port': _free_port(), 'wavs': str(d), 'out': str(tmp_path),
                                               'files': names}, 2, timeout=120)
    from nisqa_amd import NISQA_lib as NL
    ds = NL.SpeechQualityDataset(pd.DataFrame(names, columns=['deg']), data_dir=str(d), filename_column='deg',
                                 mos_column='predict_only', dim=True, seg_length=15, seg_hop_length=4, ms_hop_length=0.01)
    model = NL.NISQA_DIM(**{k: v for k, v in synth.DIM_ARGS.items()
                            if k.startswith(('cnn_', 'td', 'pool', 'ms_seg_length', 'ms_n_mels'))})
    model._engine = FakeEngine(5)
    with pytest.raises(ValueError, match=match) as one:
        NL.predict_dim(model, ds, 2, 'cpu', 0)
    for r in res:
        assert r['raised'] == 'ValueError' and r['msg'] == str(one.value) and 'bad.wav' in r['msg'], r
        assert r['seconds'] < 30


def test_raise_together_is_a_plain_raise_without_a_process_group():
    from nisqa_amd import dist
    dist.raise_together(None)
    with pytest.raises(ValueError, match='x'):
        dist.raise_together(ValueError('x'))


_WORKER_RT = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import torch
from nisqa_amd import dist
rank = int(sys.argv[1]); world = int(sys.argv[2])
torch.distributed.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=rank, world_size=world)
class Odd(Exception):
    pass
out = []
for case, err in (('none', None), ('value', ValueError('Could not load file x.wav') if rank == 1 else None),
                  ('two', (NotImplementedError('first') if rank == 1 else RuntimeError('second') if rank == 2 else None)),
                  ('custom', Odd('strange') if rank == 2 else None)):
    try:
        dist.raise_together(err)
        out.append([case, None, None])
    except Exception as e:
        out.append([case, type(e).__name__, str(e)])
json.dump(out, open(os.path.join(%(out)r, 'r%%d.json' %% rank), 'w'))
torch.distributed.destroy_process_group()
"""


def test_raise_together_carries_the_first_failing_ranks_exception_to_every_rank(tmp_path):
    """dist.raise_together at world 3 (gloo): nobody failed -> nobody raises; one rank failed -> everyone raises its type and message;
    two ranks failed -> the lower rank's exception wins everywhere (the failing higher rank included); an exception type outside the
    allow-list arrives as RuntimeError('<TypeName>: <message>') on the other ranks and as itself on its own."""
    res = _run_ranks(tmp_path, _WORKER_RT % {'root': ROOT, 'port': _free_port(), 'out': str(tmp_path)}, 3, timeout=120)
    for r, rows in enumerate(res):
        got = {c: (t, m) for c, t, m in rows}
        assert got['none'] == (None, None)
        assert got['value'] == ('ValueError', 'Could not load file x.wav')
        assert got['two'] == ('NotImplementedError', 'first')
        assert got['custom'] == (('Odd', 'strange') if r == 2 else ('RuntimeError', 'Odd: strange'))


def test_predict_loop_at_world_eight_matches_one_process(tmp_path):
    """The real loop (header probe, work-balanced contiguous shards, length-aware batches, fail-together exchange, closing
    all_gather) on eight gloo ranks with a counting engine: a length-sorted list of 0.2 ... 2.4 s clips, every rank ends with
    the full frame in input order, equal to a one-process run; segment counts per rank within 25 % of the mean (96 clips
    over 8 ranks: the granularity of a clip is 2-3 % of a rank's share)."""
    d = tmp_path / 'w'
    d.mkdir()
    durs = np.sort(np.concatenate((np.random.default_rng(5).uniform(0.2, 2.4, 94), [0.2, 2.4])))
    names = _mixed_wavs(d, durs.tolist())
    res = _run_ranks(tmp_path, _WORKER_BAL % {'root': ROOT, 'port': _free_port(), 'wavs': str(d), 'out': str(tmp_path),
                                              'files': names}, 8, timeout=600)
    assert all(r['y'] == res[0]['y'] for r in res)
    tok = np.array([r['tokens'] for r in res], dtype=np.float64)
    assert np.abs(tok / tok.mean() - 1).max() < 0.25, tok
    from nisqa_amd import NISQA_lib as NL
    ds = NL.SpeechQualityDataset(pd.DataFrame(names, columns=['deg']), data_dir=str(d), filename_column='deg',
                                 mos_column='predict_only', dim=True, seg_length=15, seg_hop_length=4, ms_hop_length=0.01)
    model = NL.NISQA_DIM(**{k: v for k, v in synth.DIM_ARGS.items()
                            if k.startswith(('cnn_', 'td', 'pool', 'ms_seg_length', 'ms_n_mels'))})
    seen = []

    class Eng(FakeEngine):
        def forward_pcm(self, pcm, plan, sr):
            seen.append(int(np.sum(plan.n_wins)))
            return super().forward_pcm(pcm, plan, sr)
    model._engine = Eng(5)
    y, _ = NL.predict_dim(model, ds, 2, 'cpu', 0)
    np.testing.assert_allclose(np.array(res[0]['y']), y, rtol=0, atol=1e-6)
    assert int(tok.sum()) == sum(seen)


def test_byte_cap_charges_what_staging_lays_out_for_mixed_sample_widths():
    """one stereo / 24-bit / float file widens its whole sample-rate group to float32 in Ingest._stage; the
    batch policy charges the byte cap with exactly that (LengthAware.staged_bytes) and sorts int16 clips before float ones
    inside a rate, so a batch mixes the two only at the seam."""
    from nisqa_amd import ingest
    rng = np.random.default_rng(11)
    n = 400
    frames = rng.integers(3 * 48000, 12 * 48000, n).astype(np.int64)
    srs = np.full(n, 48000, dtype=np.int64)
    widths = np.where(rng.random(n) < 0.1, 4, 2)                               # 10 % of the files need the host decoder
    tok = lambda f, r: np.maximum(1, -(-(1 + f // 480 - 14) // 4))
    cap = 32 << 20
    pol = ingest.LengthAware(range(n), 1, tok, min_tokens=1 << 30, byte_cap=cap)     # only the byte cap closes batches
    cuts = pol.cut(frames, srs, widths)
    assert sorted(k for c in cuts for k in c) == list(range(n))
    staged = [pol.staged_bytes(frames, srs, widths, c) for c in cuts]
    assert max(staged[:-1]) <= cap and staged[-1] <= cap + cap // 2              # the tail merge is soft by half, no more
    mixed = [c for c in cuts if len(set(widths[c].tolist())) == 2]
    assert len(mixed) <= 1                                                       # the seam
    # what the old accounting (frames x own width) would have let through: a batch of int16 clips plus one float clip
    c = [int(k) for k in np.flatnonzero(widths == 2)[:20]] + [int(np.flatnonzero(widths == 4)[0])]
    assert pol.staged_bytes(frames, srs, widths, c) == int(frames[c].sum()) * 4 > int((frames[c] * widths[c]).sum())


_WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import torch, pandas as pd
import test_host as T
from nisqa_amd import NISQA_lib as NL, synth
rank = int(sys.argv[1]); world = int(sys.argv[2])
torch.distributed.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=rank, world_size=world)
files = sorted(f for f in os.listdir(%(wavs)r) if f.endswith('.wav'))
ds = NL.SpeechQualityDataset(pd.DataFrame(files, columns=['deg']), data_dir=%(wavs)r, filename_column='deg',
                             mos_column='predict_only', dim=True)
model = NL.NISQA_DIM(**{k: v for k, v in synth.DIM_ARGS.items() if k.startswith(('cnn_', 'td', 'pool', 'ms_seg_length', 'ms_n_mels'))})
model._engine = T.FakeEngine(5)
y, _ = NL.predict_dim(model, ds, 2, 'cpu', 0)
json.dump({'y': y.tolist(), 'cols': list(ds.df.columns)}, open(os.path.join(%(out)r, 'r%%d.json' %% rank), 'w'))
torch.distributed.destroy_process_group()
'''


def test_predict_loop_sharded_over_two_gloo_ranks(tmp_path, wav_dir):
    import json
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % {'root': ROOT, 'port': port, 'wavs': str(wav_dir), 'out': str(tmp_path)})
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), '2'], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0 = json.load(open(tmp_path / 'r0.json'))
    r1 = json.load(open(tmp_path / 'r1.json'))
    assert r0['y'] == r1['y']                                 # every rank ends with the full, ordered result
    # single-process reference
    from nisqa_amd import NISQA_lib as NL
    files = sorted(f for f in os.listdir(wav_dir) if f.endswith('.wav'))
    ds = NL.SpeechQualityDataset(pd.DataFrame(files, columns=['deg']), data_dir=str(wav_dir), filename_column='deg',
                                 mos_column='predict_only', dim=True)
    model = NL.NISQA_DIM(**{k: v for k, v in synth.DIM_ARGS.items()
                            if k.startswith(('cnn_', 'td', 'pool', 'ms_seg_length', 'ms_n_mels'))})
    model._engine = FakeEngine(5)
    y, ynan = NL.predict_dim(model, ds, 2, 'cpu', 0)
    np.testing.assert_allclose(np.array(r0['y']), y, rtol=0, atol=1e-6)
    assert np.isnan(ynan).all() and ynan.shape == (5, 5)
    assert r0['cols'] == ['deg', 'mos_pred', 'noi_pred', 'dis_pred', 'col_pred', 'loud_pred']


def test_dataset_item_label_follows_the_index(tmp_path):
    """SpeechQualityDataset.__getitem__(k) returns the label of row k (NL:2217-2231), not of row 0."""
    from nisqa_amd import NISQA_lib as NL
    d = tmp_path / 'w'
    d.mkdir()
    for i in range(3):
        synth.write_wav(str(d / ('c%d.wav' % i)), synth.synth_pcm16(i, 0.3), 48000)
    df = pd.DataFrame({'deg': ['c0.wav', 'c1.wav', 'c2.wav'], 'mos': [1.5, 2.5, 3.5], 'noi': [1.0, 2.0, 3.0],
                       'dis': [4.0, 4.1, 4.2], 'col': [2.2, 2.3, 2.4], 'loud': [3.0, 3.1, np.nan]})

    class MelEngine(FakeEngine):
        def mel(self, pcm, plan, sr, clamp=True):
            return torch.zeros((int(plan.total_frames), 48)), None

    for dim, col in ((True, 'mos'), (False, 'mos')):
        ds = NL.SpeechQualityDataset(df, data_dir=str(d), filename_column='deg', mos_column=col, seg_length=15,
                                     max_length=40, seg_hop_length=4, ms_n_fft=4096, ms_hop_length=0.01,
                                     ms_win_length=0.02, ms_n_mels=48, ms_sr=None, ms_fmax=20000, dim=dim)
        ds.bind_engine(lambda: MelEngine(5 if dim else 1))
        for k in range(3):
            x, y, (idx, n_wins) = ds[k]
            want = df[['mos', 'noi', 'dis', 'col', 'loud']].iloc[k].to_numpy(np.float32) if dim else np.float32([df['mos'].iloc[k]])
            assert idx == k and y.dtype == np.float32 and y.shape == want.shape
            np.testing.assert_array_equal(y, want)
            assert tuple(x.shape) == (40, 1, 48, 15)
        np.testing.assert_array_equal(ds.labels(3)[:, 0], np.float32([1.5, 2.5, 3.5]))


def test_no_kernel_contains_the_packed_f32_op_sel_form_gfx950_misreads():
    """ISA lint.  On gfx950 (MI355X, ROCm 7.2) a v_pk_{add,mul,fma}_f32 whose LOW result reads the HIGH half of a VGPR
    src1 (op_sel:[x,1,...]) returns wrong values in lanes 48..63 while bf16 / f16 MFMA waves of ANOTHER kernel share
    the SIMD (tools/micro/corun6.hip; round 1 saw it as wrong mel frames next to the conv kernels of another stream).
    The same swap on src0 / src2 is exact, so kernels put swapped operands there; this test compiles every HIP source
    to gfx950 assembly and fails if the compiler or an asm helper produced the bad form anywhere."""
    import concurrent.futures
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.isfile(hipcc):
        pytest.skip('hipcc not on this machine')
    csrc = os.path.join(ROOT, 'nisqa_amd', 'csrc')
    srcs = sorted(f for f in os.listdir(csrc) if f.endswith('.hip'))
    bad_form = re.compile(r'v_pk_(add|mul|fma)_f32\b.*\bop_sel:\[[01],1')
    # per-file flags of the build ("train_td.o: CXXFLAGS += -fno-slp-vectorize"): the sources are linted as they are built
    extra = {m.group(1) + '.hip': m.group(2).split()
             for m in re.finditer(r'^(\w+)\.o: CXXFLAGS \+= (.*)$', open(os.path.join(csrc, 'Makefile')).read(), re.M)}

    def scan(f):
        r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-strict-aliasing', '-w'] + extra.get(f, []) +
                           ['-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', '-o', '-', f],
                           cwd=csrc, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.split('\n')
        return f, sum(1 for l in lines if 'v_pk_' in l and '_f32' in l), [l.strip() for l in lines if bad_form.search(l)]

    with concurrent.futures.ThreadPoolExecutor(8) as ex:
        res = list(ex.map(scan, srcs))
    assert any(n > 0 for _, n, _ in res), 'no packed f32 instruction found at all: the scan itself is broken'
    for f, n, bad in res:
        assert not bad, '%s: %d packed-f32 instruction(s) with op_sel on the low half of src1, e.g. %s' % (f, len(bad), bad[0])
    # the pattern does match the form the probe found
    assert bad_form.search('v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]')
    assert not bad_form.search('v_pk_add_f32 v[0:1], v[4:5], v[2:3] op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]')


def test_shipped_library_passes_the_isa_lint():
    """the lint also runs on the ARTIFACT (llvm-objdump of every gfx950 code object inside libnisqa_hip.so),
    at build time (csrc/Makefile) and here -- whatever flags or compiler version produced the library that ships."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import isa_lint
    from nisqa_amd import lib
    if not os.path.isfile(os.path.join(isa_lint.LLVM, 'llvm-objdump')):
        pytest.skip('llvm-objdump not on this machine')
    n_obj, n_pk, bad = isa_lint.scan_library(lib.LIB_PATH)
    assert n_obj >= 10 and n_pk > 1000 and not bad, (n_obj, n_pk, bad[:2])
    assert isa_lint.BAD.search('v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]')


def test_frame_to_string_equals_pandas_to_string():
    """predict() prints df.to_string(index=False) like the reference (NISQA_model.py:79); the fast formatter must give
    the same text, character for character, and must fall back to pandas for frames it does not cover."""
    import pandas as pd
    from nisqa_amd.NISQA_model import frame_to_string, _fast_frame_lines
    rng = np.random.default_rng(3)
    frames = []
    for n in (1, 2, 7, 300, 5000):
        df = pd.DataFrame({'deg': ['clip_%d.wav' % int(x) for x in rng.integers(0, 10 ** int(rng.integers(1, 7)), n)]})
        for c in ('mos_pred', 'noi_pred', 'dis_pred', 'col_pred', 'loud_pred'):
            df[c] = rng.uniform(1, 5, n).astype(np.float32).astype(float)
        df['model'] = 'NISQA_DIM'
        frames.append(df)
        g = df.copy()
        g['mos'] = np.where(rng.random(n) < 0.3, np.nan, np.round(rng.uniform(1, 5, n), 1))
        g['neg'] = rng.uniform(-3, 3, n)
        g['votes'] = rng.integers(-5, 2000, n)
        g['con'] = ['c%d' % i for i in range(n)]
        g['half'] = np.round(rng.uniform(0, 9, n)) / 2
        g['whole'] = np.round(rng.uniform(0, 9, n))
        frames.append(g)
        frames.append(g[['whole']])
        frames.append(pd.DataFrame({'x': np.full(n, np.nan)}))
    for df in frames:
        assert _fast_frame_lines(df) is not None
        assert frame_to_string(df) == df.to_string(index=False)
        assert '\n'.join(_fast_frame_lines(df)) == df.to_string(index=False)
    # frames the fast path must hand to pandas
    odd = [pd.DataFrame({'a': [1e-9, 2.0]}), pd.DataFrame({'a': [1e7, 2.0]}), pd.DataFrame({'a': [True, False]}),
           pd.DataFrame({'a': ['x', None]}), pd.DataFrame({'a': [np.inf, 1.0]}), pd.DataFrame({'a': []}),
           pd.DataFrame({'a': ['two\nlines', 'x']}), pd.DataFrame({'t': pd.to_datetime(['2020-01-01', '2021-05-05'])})]
    for df in odd:
        assert _fast_frame_lines(df) is None
        assert frame_to_string(df) == df.to_string(index=False)


def test_table_cells_formatted_inside_the_loop_give_the_reference_table(tmp_path, wav_dir, capsys, monkeypatch):
    """Round 5: nisqaModel.predict() formats the '%.6f' cells of the prediction columns batch by batch while the loop runs
    (NISQA_model.RowCells via NISQA_lib._predict(on_rows=...)) instead of behind the last batch; what is printed must be
    df.to_string(index=False) character for character, with the in-loop path on and off, and a RowCells that missed a row or
    disagrees with the frame is ignored."""
    from nisqa_amd import NISQA_model as NM
    outs = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('NISQA_FORMAT_IN_LOOP', flag)
        m = NM.nisqaModel(_args('predict_dir', _ckpt(tmp_path), data_dir=str(wav_dir)))
        m.model._engine = FakeEngine(5)
        seen = []
        orig = NM.RowCells.__call__
        monkeypatch.setattr(NM.RowCells, '__call__', lambda self, ids, rows: (seen.append(len(ids)), orig(self, ids, rows))[1])
        df = m.predict()
        monkeypatch.setattr(NM.RowCells, '__call__', orig)
        out = capsys.readouterr().out
        assert df.to_string(index=False) in out
        assert (sum(seen) == len(df)) == (flag == '1')
        assert set(m.timing) == {'predict_s', 'table_s'}
        outs[flag] = out[out.index('deg  mos_pred'):]             # (a one-time note about the reader threads may precede the table)
    assert outs['1'] == outs['0']
    df = pd.DataFrame({'deg': ['a.wav', 'b.wav', 'c.wav'], 'mos_pred': np.array([1.25, 3.5, 4.125], np.float32).astype(np.float64)})
    full = NM.RowCells(3, ['mos_pred'])
    full([2, 0, 1], np.array([[4.125], [1.25], [3.5]], np.float32))
    assert full.column('mos_pred', df['mos_pred'].to_numpy()) == ['1.250000', '3.500000', '4.125000']
    assert NM.frame_to_string(df, pre=full) == df.to_string(index=False)
    partial = NM.RowCells(3, ['mos_pred'])
    partial([0, 1], np.array([[1.25], [3.5]], np.float32))
    assert partial.column('mos_pred', df['mos_pred'].to_numpy()) is None
    wrong = NM.RowCells(3, ['mos_pred'])
    wrong([0, 1, 2], np.array([[9.0], [3.5], [4.125]], np.float32))
    assert wrong.column('mos_pred', df['mos_pred'].to_numpy()) is None and NM.frame_to_string(df, pre=wrong) == df.to_string(index=False)
    # cells justified inside the loop (final()): only for columns of non-negative values below 10 with no common trailing zeros;
    # anything else -- a value >= 10, a negative one, -0.0, NaN, 9.9999996 (formats as 10.000000), all-zero tails -- takes the general path
    rng = np.random.default_rng(2)
    base = rng.uniform(1, 5, (200, 2)).astype(np.float32)
    for label, edit, expect_final in (('regular', None, True), ('ten', 12.5, False), ('negative', -1.25, False), ('minus zero', -0.0, False),
                                      ('nan', np.nan, False), ('rounds to ten', 9.9999996, False), ('just below', 9.999999, True)):
        v = base.copy()
        if edit is not None:
            v[17, 1] = edit
        fr = pd.DataFrame({'deg': ['f%03d.wav' % i for i in range(200)], 'mos_pred': v[:, 0].astype(np.float64), 'loud_pred': v[:, 1].astype(np.float64)})
        rc = NM.RowCells(200, ['mos_pred', 'loud_pred'])
        order = rng.permutation(200)
        for s0 in range(0, 200, 64):
            ids = order[s0:s0 + 64]
            rc(ids.tolist(), v[ids])
        assert (rc.final('loud_pred', fr['loud_pred'].to_numpy()) is not None) == expect_final, label
        assert rc.final('mos_pred', fr['mos_pred'].to_numpy()) is not None
        assert NM.frame_to_string(fr, pre=rc) == fr.to_string(index=False), label
    zeros = pd.DataFrame({'deg': ['a', 'b'], 'mos_pred': [1.5, 2.25]})
    rz = NM.RowCells(2, ['mos_pred'])
    rz([0, 1], np.array([[1.5], [2.25]], np.float32))
    assert rz.final('mos_pred', zeros['mos_pred'].to_numpy()) is None and NM.frame_to_string(zeros, pre=rz) == zeros.to_string(index=False)


def test_ingest_cpu_budget_reader_cap_ring_reuse_and_path_cache(tmp_path, monkeypatch):
    """Host details of the predict loop: the reader count respects the cgroup CPU quota, page-locked rings
    are handed from one loop to the next, file names come out of the DataFrame once and travel with the staged groups."""
    import pandas as pd
    from nisqa_amd import ingest
    from nisqa_amd.NISQA_lib import SpeechQualityDataset
    n = ingest.cpu_budget()
    assert isinstance(n, int) and 1 <= n <= (os.cpu_count() or 1)
    paths = []
    for i in range(6):
        p = str(tmp_path / ('q%d.wav' % i))
        synth.write_wav(p, (np.arange(300 + i) % 97).astype(np.int16), 48000)
        paths.append(os.path.basename(p))
    df = pd.DataFrame({'deg': paths})
    ds = SpeechQualityDataset(df, data_dir=str(tmp_path), filename_column='deg', mos_column='predict_only', dim=True)
    assert ds.file_paths([0, 3, 5]) == [ds.file_path(i) for i in (0, 3, 5)]
    ds.df = df.iloc[::-1].reset_index(drop=True)                     # a new frame invalidates the cached column
    assert ds.file_paths([0]) == [ds.file_path(0)] == [os.path.join(str(tmp_path), paths[-1])]
    monkeypatch.setattr(ingest, 'cpu_budget', lambda: 6)
    ing = ingest.Ingest(ds, [[0, 1, 2], [3, 4, 5]], pin=False, num_workers=64)
    assert ing.workers == 3                                          # budget - 3 (producer, consumer, the interpreter)
    ring = ing.ring
    for staged in ing:
        assert [g.names for g in staged.groups] == [ds.file_paths(g.ids) for g in staged.groups]
        ing.ring.release_after(staged.slot, None)
    ing.close()
    ing.close()                                                      # idempotent
    assert ing.stats['batches'] == 2 and ing.stats['read'] > 0
    ing2 = ingest.Ingest(ds, [[0, 1]], pin=False, num_workers=1)     # the next loop takes over the finished loop's ring
    assert ing2.ring is ring
    for staged in ing2:
        ing2.ring.release_after(staged.slot, None)
    ing2.close()


# ---- length-aware batching (round 3) -----------------------------------------------
def _mixed_wavs(tmp_path, durs, srs=None):
    names = []
    for i, d in enumerate(durs):
        sr = 48000 if srs is None else srs[i]
        synth.write_wav(str(tmp_path / ('m%03d.wav' % i)), synth.synth_pcm16(50 + i, d, sr=sr) if sr != 48000 else synth.synth_pcm16(50 + i, d), sr)
        names.append('m%03d.wav' % i)
    return names


def test_length_aware_policy_cuts_by_work_and_keeps_every_item_once():
    from nisqa_amd import ingest
    rng = np.random.default_rng(4)
    frames = rng.integers(3 * 48000, 30 * 48000, 300).astype(np.int64)
    srs = np.full(300, 48000, dtype=np.int64)
    tok = lambda f, r: np.maximum(1, -(-(1 + f // 480 - 14) // 4))
    pol = ingest.LengthAware(range(300), 1, tok, min_tokens=16384, byte_cap=64 << 20)
    cuts = pol.cut(frames, srs, np.full(300, 2))
    assert sorted(k for c in cuts for k in c) == list(range(300))              # a partition
    flat = [k for c in cuts for k in c]
    assert (np.diff(frames[flat]) >= 0).all()                                  # sorted by length across batches
    for c in cuts[:-1]:
        t, b = int(tok(frames[c], srs[c]).sum()), int(frames[c].sum() * 2)
        assert b <= 64 << 20
        # closed because the work target was reached, or because one more clip would not have fitted
        nxt = flat[flat.index(c[-1]) + 1]
        assert t >= 16384 or b + frames[nxt] * 2 > 64 << 20
        assert t - int(tok(frames[c[-1:]], srs[c[-1:]])[0]) < 16384            # ... and not later than necessary
    # bs is a lower bound on the clip count; min_clips (the LSTM path) likewise
    cuts = ingest.LengthAware(range(300), 40, tok, min_tokens=0).cut(frames, srs, np.full(300, 2))
    assert [len(c) for c in cuts] == [40] * 7 + [20]
    cuts = ingest.LengthAware(range(300), 1, tok, min_tokens=0, min_clips=128).cut(frames, srs, np.full(300, 2))
    assert [len(c) for c in cuts] == [128, 172]            # a remainder under half a batch rides with the batch before it ...
    cuts = ingest.LengthAware(range(300), 1, tok, min_tokens=0, min_clips=128, byte_cap=200 << 20).cut(frames, srs, np.full(300, 2))
    assert [len(c) for c in cuts] == [128, 108, 64] and all(int(frames[c].sum() * 2) <= 200 << 20 for c in cuts)   # ... when it is small and fits
    # rates are kept apart once a batch is big enough, mixed otherwise
    srs2 = np.where(np.arange(300) % 2 == 0, 48000, 16000)
    cuts = ingest.LengthAware(range(300), 150, tok, min_tokens=0).cut(frames, srs2, np.full(300, 2))
    assert [sorted(set(srs2[c].tolist())) for c in cuts] == [[16000], [48000]]
    assert ingest.LengthAware([], 1, tok).cut(frames[:0], srs[:0], np.zeros(0, np.int64)) == []


@pytest.mark.parametrize('window', [4, 16384])
def test_predict_rows_come_back_in_input_order_under_length_sorting(tmp_path, monkeypatch, window):
    """The reference's default flags (--bs 1 --num_workers 0, run_predict.py:16-20) coalesce into work-sized batches of
    length-sorted clips; the frame still lists the files in directory / CSV order with each file's own row."""
    from nisqa_amd import ingest
    from nisqa_amd import NISQA_lib as NL
    from nisqa_amd.NISQA_model import nisqaModel
    durs = [0.9, 0.25, 0.6, 0.3, 1.4, 0.2, 0.75, 0.5, 1.1, 0.35, 0.45]
    names = _mixed_wavs(tmp_path, durs)
    order = [7, 2, 9, 0, 4, 1, 10, 3, 8, 5, 6]
    pd.DataFrame({'name': [names[i] for i in order]}).to_csv(tmp_path / 'l.csv', index=False)
    ck = _ckpt(tmp_path)
    seen = []
    real_stage = ingest.Ingest._stage

    def spy(self, idx, probed=None):
        seen.append(list(idx))
        return real_stage(self, idx, probed)
    monkeypatch.setattr(ingest.Ingest, '_stage', spy)
    monkeypatch.setattr(NL, 'MIN_TOKENS_SA', 60)                   # a 0.9 s clip has 20 segments
    real_init = ingest.LengthAware.__init__
    monkeypatch.setattr(ingest.LengthAware, '__init__',
                        lambda self, *a, **k: real_init(self, *a, **{**k, 'window': window}))
    m = nisqaModel(_args('predict_csv', ck, data_dir=str(tmp_path), csv_file='l.csv', csv_deg='name', bs=1, tr_bs_val=1))
    m.model._engine = FakeEngine(5)
    df = m.predict()
    assert list(df['name']) == [names[i] for i in order]
    eng = FakeEngine(5)
    for row, i in enumerate(order):                                  # each file's own numbers, whatever batch it rode in
        y, sr = NL.read_wav(str(tmp_path / names[i]))
        want = eng.forward_pcm(torch.from_numpy(y), eng.plan([len(y)], sr), sr)[0].numpy()
        assert np.allclose(df.iloc[row][['mos_pred', 'noi_pred', 'dis_pred', 'col_pred', 'loud_pred']].to_numpy(dtype=np.float64), want, rtol=1e-6)
    assert sorted(i for b in seen for i in b) == list(range(len(order)))
    assert len(seen) < len(order)                                   # bs = 1 did NOT mean one launch chain per file
    if window > len(order):
        lens = [durs[order[i]] for b in seen for i in b]
        assert lens == sorted(lens)                                  # similar lengths share a batch
    else:
        assert max(len(b) for b in seen) <= window
    # NISQA_EXACT_BS=1: the reference's batches (index order, exactly bs clips)
    seen.clear()
    monkeypatch.setenv('NISQA_EXACT_BS', '1')
    m = nisqaModel(_args('predict_csv', ck, data_dir=str(tmp_path), csv_file='l.csv', csv_deg='name', bs=4, tr_bs_val=4))
    m.model._engine = FakeEngine(5)
    df2 = m.predict()
    assert seen == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10]]
    assert np.allclose(df2['mos_pred'], df['mos_pred'], rtol=1e-6)


def test_predict_sees_an_in_place_edit_of_the_filename_column(tmp_path):
    """nothing about the file list is cached on the dataset between two predict() calls."""
    from nisqa_amd.NISQA_model import nisqaModel
    names = _mixed_wavs(tmp_path, [0.3, 0.5, 0.7])
    pd.DataFrame({'name': names}).to_csv(tmp_path / 'l.csv', index=False)
    m = nisqaModel(_args('predict_csv', _ckpt(tmp_path), data_dir=str(tmp_path), csv_file='l.csv', csv_deg='name'))
    m.model._engine = FakeEngine(5)
    a = m.predict()['mos_pred'].to_numpy().copy()
    m.ds_val.df['name'] = m.ds_val.df['name'].to_numpy()[::-1].copy()      # same frame object, same length
    b = m.predict()['mos_pred'].to_numpy()
    assert np.allclose(b, a[::-1]) and not np.allclose(b, a)


def test_probe_errors_surface_as_the_reference_value_error(tmp_path):
    from nisqa_amd.NISQA_model import nisqaModel
    names = _mixed_wavs(tmp_path, [0.3, 0.5])
    (tmp_path / 'bad.wav').write_bytes(b'not a wav file at all')
    pd.DataFrame({'name': names + ['bad.wav']}).to_csv(tmp_path / 'l.csv', index=False)
    m = nisqaModel(_args('predict_csv', _ckpt(tmp_path), data_dir=str(tmp_path), csv_file='l.csv', csv_deg='name'))
    m.model._engine = FakeEngine(5)
    with pytest.raises(ValueError, match='Could not load file .*bad.wav'):
        m.predict()


class _Evil(object):
    def __reduce__(self):
        return (print, ('code ran at load time',))


def test_checkpoint_with_pickled_objects_is_refused_unless_opted_in(tmp_path, monkeypatch, capsys):
    """the restricted unpickler is the only path by default; I/O errors are not masked."""
    from nisqa_amd.NISQA_model import _load_checkpoint
    good = _ckpt(tmp_path)
    assert set(_load_checkpoint(good)) == {'args', 'model_state_dict'}
    bad = str(tmp_path / 'evil.tar')
    torch.save({'args': {'x': _Evil()}, 'model_state_dict': {}}, bad)
    monkeypatch.delenv('NISQA_ALLOW_UNSAFE_CHECKPOINT', raising=False)
    with pytest.raises(RuntimeError, match='NISQA_ALLOW_UNSAFE_CHECKPOINT'):
        _load_checkpoint(bad)
    assert 'code ran at load time' not in capsys.readouterr().out
    with pytest.raises(FileNotFoundError):
        _load_checkpoint(str(tmp_path / 'missing.tar'))
    monkeypatch.setenv('NISQA_ALLOW_UNSAFE_CHECKPOINT', '1')
    ck = _load_checkpoint(bad)
    assert 'args' in ck and 'code ran at load time' in capsys.readouterr().out


def test_legacy_format_checkpoint_cannot_run_code_through_the_salvage_pass(tmp_path, monkeypatch):
    """torch's legacy (non-zip) reader reads the magic number / protocol / sys_info with
    pickle_module.load; the salvage module used to hand it the stock pickle.load, so a file whose FIRST pickle is a
    __reduce__ ran its callable although weights_only=True had refused it.  Every stream now goes through the allow-list."""
    import pickle
    from nisqa_amd.NISQA_model import _load_checkpoint
    trap = tmp_path / 'ran.txt'

    class Boom(object):
        def __reduce__(self):
            return (open, (str(trap), 'w'))
    evil = tmp_path / 'legacy.tar'
    with open(evil, 'wb') as f:
        pickle.dump(Boom(), f, protocol=2)                       # where the legacy reader expects the magic number
        pickle.dump(1001, f, protocol=2)
        pickle.dump({}, f, protocol=2)
    # a genuine legacy-format checkpoint with an object in it: refused by weights_only, salvaged through the allow-list
    legacy = tmp_path / 'legacy_ok.tar'
    torch.save({'args': {'a': 1}, 'model_state_dict': {'w': torch.arange(4.)}, 'trap': Boom()}, legacy,
               _use_new_zipfile_serialization=False)
    monkeypatch.delenv('NISQA_ALLOW_UNSAFE_CHECKPOINT', raising=False)
    with pytest.raises(Exception):
        _load_checkpoint(str(evil))
    assert not trap.exists()
    ck = _load_checkpoint(str(legacy))
    assert not trap.exists() and ck['args'] == {'a': 1} and torch.equal(ck['model_state_dict']['w'], torch.arange(4.))


def test_checkpoint_written_by_the_reference_trainer_is_salvaged_without_running_it(tmp_path, monkeypatch, capsys):
    """the reference's trainer saves db_results DataFrames and numpy scalars next to the tensors (reference
    NISQA_model.py:1096-1108); the restricted unpickler rejects such a file.  It is loaded through the salvage unpickler:
    only tensor-rebuild globals resolve, everything else becomes an inert stub (the file's code never runs), and only
    args + model_state_dict are kept.  A stub INSIDE args still refuses."""
    from nisqa_amd.NISQA_model import _load_checkpoint
    monkeypatch.delenv('NISQA_ALLOW_UNSAFE_CHECKPOINT', raising=False)
    sd = synth.random_state_dict(7)
    args = dict(synth.DIM_ARGS, now=__import__('datetime').datetime(2021, 3, 4))
    p = str(tmp_path / 'trained.tar')
    torch.save({'args': args, 'model_state_dict': sd, 'epoch': np.int64(3), 'r': {'r_p_mean_file': np.float64(0.9)},
                'db_results': {'db': pd.DataFrame({'x': [1.0, 2.0]})}, 'optimizer_state_dict': {'state': {0: torch.zeros(3)}},
                'trap': _Evil()}, p)
    ck = _load_checkpoint(p)
    out = capsys.readouterr().out
    assert 'code ran at load time' not in out and 'loaded args and model_state_dict only' in out
    assert set(ck) == {'args', 'model_state_dict'} and ck['args'] == args
    assert all(torch.equal(ck['model_state_dict'][k], torch.as_tensor(sd[k])) for k in sd) and len(ck['model_state_dict']) == len(sd)
    # the whole drop-in surface takes it
    from nisqa_amd import NISQA_lib as NL
    m = NL.NISQA_DIM(**{k: v for k, v in ck['args'].items() if k.startswith(('cnn_', 'td', 'pool', 'ms_seg_length', 'ms_n_mels'))})
    m.load_state_dict(ck['model_state_dict'], strict=True)
    bad = str(tmp_path / 'bad_args.tar')
    torch.save({'args': dict(args, lr=np.float64(1e-3)), 'model_state_dict': sd}, bad)
    with pytest.raises(RuntimeError, match='NISQA_ALLOW_UNSAFE_CHECKPOINT'):
        _load_checkpoint(bad)


@pytest.mark.parametrize('law', ['mulaw', 'alaw'])
def test_g711_wav_files_load_like_soundfile_would(tmp_path, law):
    """lb.load reads G.711 A-law / mu-law WAVs through libsndfile: code word -> 16-bit table value -> / 32768.  The table
    is checked against CPython's audioop (an independent G.711 implementation); the native probe accepts the file and the
    predict loop stages it as float32 through the host decoder."""
    import audioop
    import warnings
    from nisqa_amd import wavio, lib
    from nisqa_amd.NISQA_model import nisqaModel
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 256, 48000 // 2, dtype=np.uint8)
    stereo = np.stack([codes, codes[::-1]], 1)
    synth.write_wav(str(tmp_path / 'm.wav'), codes, 48000, g711=law)
    synth.write_wav(str(tmp_path / 's.wav'), stereo, 48000, g711=law)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        lin = audioop.ulaw2lin(codes.tobytes(), 2) if law == 'mulaw' else audioop.alaw2lin(codes.tobytes(), 2)
    want = np.frombuffer(lin, dtype='<i2').astype(np.float32) / np.float32(32768.0)
    y, sr = wavio.read_wav(str(tmp_path / 'm.wav'))
    assert sr == 48000 and y.dtype == np.float32 and np.array_equal(y, want)
    ys, _ = wavio.read_wav(str(tmp_path / 's.wav'))
    assert np.allclose(ys, 0.5 * (want + want[::-1]), atol=1e-7)
    y1, _ = wavio.read_wav(str(tmp_path / 's.wav'), ms_channel=1)
    assert np.array_equal(y1, want[::-1])
    L = lib.load_ingest()
    infos = (lib.WavInfo * 1)()
    paths = (ctypes.c_char_p * 1)(os.fsencode(str(tmp_path / 'm.wav')))
    assert L.nisqa_ingest_probe(paths, 1, infos, 1) == 0 and infos[0].tag == (7 if law == 'mulaw' else 6) and infos[0].n_frames == len(codes)
    pd.DataFrame({'name': ['m.wav', 's.wav']}).to_csv(tmp_path / 'l.csv', index=False)
    m = nisqaModel(_args('predict_csv', _ckpt(tmp_path), data_dir=str(tmp_path), csv_file='l.csv', csv_deg='name'))
    m.model._engine = FakeEngine(5)
    df = m.predict()
    eng = FakeEngine(5)
    ref = eng.forward_pcm(torch.from_numpy(want), eng.plan([len(want)], 48000), 48000)[0].numpy()
    assert np.allclose(df.iloc[0][['mos_pred', 'noi_pred', 'dis_pred', 'col_pred', 'loud_pred']].to_numpy(dtype=np.float64), ref, rtol=1e-6)


def test_length_aware_policy_invariants_on_random_inputs():
    """Property test (hypothesis): whatever the lengths, rates, hints and caps, the batches are a partition of the window,
    no batch but a merged tail exceeds the byte cap (and never by more than half), every batch but the last of a rate
    reaches the clip / work lower bounds unless the byte cap closed it, and within a rate lengths ascend across batches."""
    from hypothesis import given, settings, strategies as st
    from nisqa_amd import ingest

    @settings(max_examples=150, deadline=None)
    @given(st.lists(st.tuples(st.integers(700, 1500000), st.sampled_from([16000, 48000])), min_size=0, max_size=120),
           st.integers(1, 40), st.integers(0, 3000), st.integers(1, 24), st.integers(1 << 16, 1 << 23))
    def check(items, bs, min_tokens, min_clips, cap):
        frames = np.array([f for f, _ in items], dtype=np.int64)
        srs = np.array([r for _, r in items], dtype=np.int64)
        tok = lambda f, r: np.maximum(1, (1 + f // (r // 100) - 14 + 3) // 4)
        pol = ingest.LengthAware(range(len(items)), bs, tok, min_tokens=min_tokens, min_clips=min_clips, byte_cap=cap)
        cuts = pol.cut(frames, srs, np.full(len(items), 2, dtype=np.int64))
        assert sorted(k for c in cuts for k in c) == list(range(len(items)))
        need = max(bs, min_clips)
        for i, c in enumerate(cuts):
            assert c, 'empty batch'
            b = int(frames[c].sum() * 2)
            assert b <= cap + cap // 2 or len(c) == 1                      # one clip larger than the cap still gets its batch
            last_of_rate = i + 1 == len(cuts) or srs[cuts[i + 1][0]] != srs[c[-1]]
            if not last_of_rate and len(set(srs[c].tolist())) == 1:
                nxt = cuts[i + 1][0]
                full = len(c) >= need and int(tok(frames[c], srs[c]).sum()) >= min_tokens
                assert full or b + int(frames[nxt]) * 2 > cap
        for r in set(srs.tolist()):
            seq = [int(frames[k]) for c in cuts for k in c if srs[k] == r]
            assert seq == sorted(seq)
    check()


def test_training_step_tables_describe_the_ragged_products():
    """nisqa_amd.train.step_tables (host side of HipTrainer._prepare): one descriptor row per clip for each grouped GEMM, the
    softmax row tables, segment offsets -- checked against a direct construction of the ragged attention of three clips."""
    from nisqa_amd.train import step_tables
    L = np.array([3, 70, 1])
    parts, tiles = step_tables(L)
    tv = dict(parts)
    tok, sq = np.array([0, 3, 73, 74]), np.array([0, 9, 4909, 4910])
    assert tv['seg_off'].dtype == np.int32 and list(tv['seg_off']) == list(tok)
    qk = tv['desc_qk']
    assert qk.shape == (3, 10) and qk.dtype == np.int64
    assert list(qk[:, 0]) == list(tok[:-1] * 192) and list(qk[:, 2]) == list(sq[:-1])         # Q rows of the clip, its score block
    assert list(qk[:, 3]) == list(L) and list(qk[:, 4]) == list(L) and list(qk[:, 5]) == [64, 64, 64]
    assert list(qk[:, 9]) == [0, 1, 5] and tiles['qk'] == 6                                   # 1 + 2 x 2 + 1 tiles of 64 x 64
    pv = tv['desc_pv']
    assert list(pv[:, 3]) == list(L) and list(pv[:, 4]) == [64, 64, 64] and list(pv[:, 5]) == list(L) and tiles['pv'] == 4
    assert list(tv['desc_pool'][:, 3]) == [1, 1, 1] and tiles['pool'] == 3
    # softmax rows: row r of clip b starts at sq[b] + r * L[b] and has L[b] entries
    want_off = np.concatenate([sq[b] + np.arange(L[b]) * L[b] for b in range(3)])
    assert tv['att_off'].dtype == np.int64 and np.array_equal(tv['att_off'], want_off)
    assert np.array_equal(tv['att_len'], np.repeat(L, L)) and tv['att_len'].dtype == np.int32
    assert list(tv['pool_off']) == list(tok[:-1]) and list(tv['pool_len']) == list(L)
    # every kind the trainer asks for is there
    assert set(tiles) == {'qk', 'pv', 'dp', 'dv', 'dq', 'dk', 'pool', 'datt', 'outer'}


def test_fused_self_attention_block_plan_tables():
    """nisqa_tdtrain_plan (host side of csrc/train_td.hip, no GPU work): the padded token space, the descriptors of the one
    weight-gradient GEMM (d Y^T X per parameter matrix, tiles counted like nisqa_gemm_f32 counts them) and the column-sum
    jobs land on the parameters' own offsets; buffers do not overlap."""
    from nisqa_amd import lib
    L = lib.load()
    for lens, nl, nh in (([5, 40, 32, 1], 2, 5), ([247] * 3, 2, 1), ([33], 1, 1)):
        lens = np.array(lens)
        B, S = len(lens), int(lens.sum())
        NP = int(((lens + 31) // 32 * 32).sum())
        n_par = 4 + 12 * nl + 6 * nh
        poff = (np.arange(n_par, dtype=np.int32) * 50000)
        cap = 8 + (1 + 4 * nl + 2 * nh) * 10 + (2 + 6 * nl + nh) * 6
        out = np.zeros(cap, dtype=np.int64)
        assert L.nisqa_tdtrain_plan(B, S, NP, nl, nh, poff.ctypes.data, out.ctypes.data, cap) == 0
        assert L.nisqa_tdtrain_plan(B, S, NP, nl, nh, poff.ctypes.data, out.ctypes.data, cap - 1) == lib.NISQA_ERR_ARG
        assert L.nisqa_tdtrain_plan(B, S, NP + 1, nl, nh, poff.ctypes.data, out.ctypes.data, cap) == lib.NISQA_ERR_ARG
        ws, ng, tiles, nj = int(out[0]), int(out[2]), int(out[3]), int(out[4])
        assert ng == 1 + 4 * nl + 2 * nh and nj == 2 + 6 * nl + nh
        d = out[8:8 + ng * 10].reshape(ng, 10)
        j = out[8 + ng * 10:8 + ng * 10 + nj * 6].reshape(nj, 6)
        nt = ((d[:, 3] + 63) // 64) * ((d[:, 4] + 63) // 64)
        assert tiles == nt.sum() and (d[:, 9] == np.concatenate(([0], np.cumsum(nt)[:-1]))).all()
        assert (d[0, 3:9] == [64, 384, S, 64, 384, 384]).all() and d[0, 1] == out[5] and d[0, 2] == poff[0]
        assert set(d[:, 2].tolist()) == {int(poff[i]) for i in ([0] + [4 + 12 * l + q for l in range(nl) for q in (0, 2, 6, 8)]
                                                               + [4 + 12 * nl + 6 * h + q for h in range(nh) for q in (0, 2)])}
        assert (d[1:, 5] == NP).all() and (j[:, 3] == NP).all()
        # operand extents stay inside the workspace and the spans of different buffers do not overlap
        spans = sorted({(int(a), int(a + k * m)) for a, k, m in zip(d[:, 0], d[:, 5], d[:, 3])} |
                       {(int(b), int(b + k * n)) for b, k, n in zip(d[:, 1], d[:, 5], d[:, 4])})
        assert spans[-1][1] <= ws and all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
        bias_like = {int(poff[i]) for i in ([1, 2, 3] + [4 + 12 * l + q for l in range(nl) for q in (1, 3, 4, 5, 7, 9, 10, 11)]
                                            + [4 + 12 * nl + 6 * h + 1 for h in range(nh)])}
        assert {int(v) for v in j[:, 4:6].reshape(-1) if v >= 0} == bias_like


def _scipy_wav_files():
    import scipy
    d = os.path.join(os.path.dirname(scipy.__file__), 'io', 'tests', 'data')
    import glob
    return sorted(glob.glob(os.path.join(d, '*.wav')))


def test_wav_reader_against_files_written_by_other_tools():
    """Independent vectors for the ingest row: the WAV files scipy ships for its own reader tests (written by other tools:
    RIFX big-endian, RF64, WAVE_FORMAT_EXTENSIBLE, 5 / 12 / 20 / 24 / 32-bit PCM, 32 / 64-bit float, mu-law, truncated and
    corrupt headers).  Wherever scipy.io.wavfile reads a file of a width soundfile knows (<= 32 bits), wavio.read_wav and the
    native probe must read it too and agree bit for bit after soundfile's scaling; what scipy rejects as corrupt must raise
    the reference's error (NISQA_lib.py:2305-2306)."""
    import warnings
    import scipy.io.wavfile as sw
    from nisqa_amd import wavio
    files = _scipy_wav_files()
    if len(files) < 10:
        pytest.skip('scipy test data not installed')
    lib, L = _ingest_lib()
    seen = set()
    for f in files:
        name = os.path.basename(f)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                sr, ref = sw.read(f)
        except Exception:
            ref = None
        paths = (ctypes.c_char_p * 1)(os.fsencode(f))
        info = (lib.WavInfo * 1)()
        bad = L.nisqa_ingest_probe(paths, 1, info, 1)
        if ref is None and 'ulaw' not in name:                 # corrupt on purpose (scipy has no mu-law decoder; we do)
            with pytest.raises(ValueError, match='Could not load file'):
                wavio.read_wav(f)
            assert bad == 1
            seen.add('corrupt')
            continue
        if ref is not None and ref.dtype == np.int64:          # 36 .. 64-bit PCM: libsndfile has no such format either
            with pytest.raises(ValueError, match='Could not load file'):
                wavio.read_wav(f)
            assert bad == 1
            seen.add('wide')
            continue
        y, sr2 = wavio.read_wav(f)
        assert bad == 0 and info[0].sample_rate == sr2 and info[0].n_frames == len(y), name
        assert bool(info[0].tag & 0x10000) == ('-be' in name or 'Hz-be-' in name), name      # NISQA_WAV_TAG_BIG_ENDIAN
        if ref is None:
            seen.add('ulaw')
            continue
        assert sr2 == sr, name
        r = ref.astype(ref.dtype.newbyteorder('='))
        if r.dtype == np.uint8:
            rf = (r.astype(np.float32) - np.float32(128)) / np.float32(128)
        elif r.dtype == np.int16:
            rf = r.astype(np.float32) / np.float32(32768)
        elif r.dtype == np.int32:                               # 20 / 24-bit arrive left-justified in 32 bits
            rf = (r.astype(np.float64) / 2.0 ** 31).astype(np.float32)
        else:
            rf = r.astype(np.float32)
        if rf.ndim == 2:
            rf = np.mean(rf.T, axis=0, dtype=np.float32)        # librosa.to_mono
        if y.dtype == np.int16:
            y = y.astype(np.float32) / np.float32(32768)
        assert y.shape == rf.shape and np.array_equal(y, rf), name
        seen.add(str(ref.dtype) + ('/be' if info[0].tag & 0x10000 else ''))
    assert {'corrupt', 'wide', 'ulaw', 'uint8', 'int16', 'int32', 'float32', 'float64'} <= seen and any(k.endswith('/be') for k in seen), seen


def test_big_endian_pcm16_never_takes_the_verbatim_int16_path(tmp_path):
    """A RIFX mono PCM16 file has the layout of the fast path (2 bytes per sample, mono) but byte-swapped samples: staged next
    to little-endian files of the same rate, its group must go through the host decoder (float32), and the samples must be
    those of the little-endian twin."""
    import struct
    from nisqa_amd import ingest
    rng = np.random.default_rng(17)
    pcm = (rng.standard_normal(901) * 3000).astype(np.int16)
    le = str(tmp_path / 'le.wav')
    synth.write_wav(le, pcm, 16000)
    fmt = struct.pack('>HHIIHH', 1, 1, 16000, 32000, 2, 16)
    body = b'fmt ' + struct.pack('>I', 16) + fmt + b'data' + struct.pack('>I', 2 * len(pcm)) + pcm.astype('>i2').tobytes()
    be = str(tmp_path / 'be.wav')
    open(be, 'wb').write(b'RIFX' + struct.pack('>I', 4 + len(body)) + b'WAVE' + body)
    y_be, sr = wavio.read_wav(be)
    y_le, _ = wavio.read_wav(le)
    assert sr == 16000 and y_be.dtype == np.int16 and np.array_equal(y_be, y_le) and np.array_equal(y_le, pcm)
    h = wavio.probe(be)
    assert h.be and not h.fast
    h.close()
    ing = ingest.Ingest(_ListDataset([le, be, le]), [[0, 1, 2]], pin=False, num_workers=2)
    try:
        staged = next(iter(ing))
        (g,) = staged.groups
        assert not g.is_i16 and g.sr == 16000 and g.lengths == [901, 901, 901]
        host = ing.ring.buf[staged.slot][g.offset:g.offset + g.nbytes].view(torch.float32).numpy()
        want = pcm.astype(np.float32) / np.float32(32768.0)
        np.testing.assert_array_equal(host, np.concatenate([want, want, want]))
        ing.ring.release_after(staged.slot, None)
    finally:
        ing.close()


def test_batch_floor_grows_with_the_job():
    """batch_policy: the work floor of a batch is MIN_TOKENS_SA for jobs up to ~25 k items and grows to 4 x for jobs of 100 k and more
    (an H2D copy carries a fixed cost; a job still needs a few hundred batches to pipeline); --bs stays a lower bound, the LSTM
    architecture keeps its own clip floor, NISQA_MIN_TOKENS overrides."""
    from nisqa_amd import NISQA_lib as NL

    class Eng(object):
        arch = 0

    class Ds(object):
        seg_length, seg_hop_length, max_length = 15, 4, 1300
        ms_n_fft, ms_hop_length, ms_sr = 4096, 0.01, None
    for n, want in ((1000, 1), (25343, 1), (25344, 1), (50688, 2), (98304, 3), (101376, 4), (10 ** 6, 4)):
        pol = NL.batch_policy(Eng(), Ds(), range(n), 64)
        assert pol.min_tokens == NL.MIN_TOKENS_SA * want and pol.bs == 64 and pol.min_clips == 1, (n, pol.min_tokens)
    lstm = Eng()
    lstm.arch = 1
    pol = NL.batch_policy(lstm, Ds(), range(10 ** 6), 1)
    assert pol.min_tokens == 0 and pol.min_clips == NL.MIN_CLIPS_LSTM
    os.environ['NISQA_MIN_TOKENS'] = '777'
    try:
        assert NL.batch_policy(Eng(), Ds(), range(10 ** 6), 1).min_tokens == 777
    finally:
        del os.environ['NISQA_MIN_TOKENS']

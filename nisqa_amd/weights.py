"""Pack an unchanged NISQA ``model_state_dict`` into the device blobs the HIP kernels read.

Layouts mirror ``csrc/layout.hpp`` (tests/test_host.py checks the two agree):

* conv B-fragments  wf[tap][step][ntile][lane][4]:
      W[n = (lane&31)+32*ntile][c = 8*step + 4*(lane>>5) + kk][dy][dx] * bn_scale[n],  tap = dy*3+dx
* linear A-fragments af[step][mtile][lane][4]:
      W[row = (lane&31)+32*mtile][k = 8*step + 4*(lane>>5) + kk]

BatchNorm (eval, running statistics; reference NISQA_lib.py:690-705) is folded in float64:
scale = gamma / sqrt(var + 1e-5), w' = w * scale, t = (conv_bias - mean) * scale + beta.
"""
import numpy as np

BN_EPS = 1e-5

# ---- offsets (floats), identical to csrc/layout.hpp ------------------------------------------
CNN_W1 = 0
CNN_T1 = CNN_W1 + 16 * 9
CNN_WF2 = CNN_T1 + 16
CNN_T2 = CNN_WF2 + 9 * 2 * 1 * 256
CNN_WF3 = CNN_T2 + 32
CNN_T3 = CNN_WF3 + 9 * 4 * 2 * 256
CNN_WF4 = CNN_T3 + 64
CNN_T4 = CNN_WF4 + 9 * 8 * 2 * 256
CNN_WF5 = CNN_T4 + 64
CNN_T5 = CNN_WF5 + 9 * 8 * 2 * 256
CNN_WF6 = CNN_T5 + 64
CNN_T6 = CNN_WF6 + 9 * 8 * 2 * 256
CNN_W_FLOATS = CNN_T6 + 64

TD_PROJ_AF = 0
TD_PROJ_B = TD_PROJ_AF + 48 * 2 * 256
TD_LN0_G = TD_PROJ_B + 64
TD_LN0_B = TD_LN0_G + 64
TD_LAYER0 = TD_LN0_B + 64
TDL_QKV_AF = 0
TDL_QKV_B = TDL_QKV_AF + 8 * 6 * 256
TDL_OUT_AF = TDL_QKV_B + 192
TDL_OUT_B = TDL_OUT_AF + 8 * 2 * 256
TDL_LN1_G = TDL_OUT_B + 64
TDL_LN1_B = TDL_LN1_G + 64
TDL_FF1_AF = TDL_LN1_B + 64
TDL_FF1_B = TDL_FF1_AF + 8 * 2 * 256
TDL_FF2_AF = TDL_FF1_B + 64
TDL_FF2_B = TDL_FF2_AF + 8 * 2 * 256
TDL_LN2_G = TDL_FF2_B + 64
TDL_LN2_B = TDL_LN2_G + 64
TDL_FLOATS = TDL_LN2_B + 64

CNNS_FC_W = CNN_W_FLOATS
CNNS_FC_B = CNNS_FC_W + 768 * 20
CNNS_W_FLOATS = CNNS_FC_B + 32

LSTM_WIH = 0
LSTM_WHH = LSTM_WIH + 512 * 20
LSTM_B = LSTM_WHH + 512 * 128
LSTM_DIR_FLOATS = LSTM_B + 512
LSTM_POOL_W = 2 * LSTM_DIR_FLOATS
LSTM_W_FLOATS = LSTM_POOL_W + 256 + 4

CNNB_W1 = 0
CNNB_W2 = CNNB_W1 + 3 * 512
CNNB_W3 = CNNB_W2 + 9 * 1 * 2 * 512
CNNB_W4 = CNNB_W3 + 18 * 2 * 2 * 512
CNNB_W5 = CNNB_W4 + 36 * 2 * 2 * 512
CNNB_W6 = CNNB_W5 + 36 * 2 * 2 * 512
CNNB_U16S = CNNB_W6 + 36 * 2 * 2 * 512
# three-term fragments (csrc/cnn_bf16x6.hip): [3] terms per fragment
CNNX_W1 = 0
CNNX_W2 = CNNX_W1 + 3 * 512
CNNX_W3 = CNNX_W2 + 9 * 1 * 3 * 512
CNNX_W4 = CNNX_W3 + 18 * 2 * 3 * 512
CNNX_W5 = CNNX_W4 + 36 * 2 * 3 * 512
CNNX_W6 = CNNX_W5 + 36 * 2 * 3 * 512
CNNX_U16S = CNNX_W6 + 36 * 2 * 3 * 512
# two-term f16 fragments + per-layer constants (csrc/cnn_bf16.hip, formats F16X3 / F16X4)
CNNH_META = CNNB_U16S
CNNH_U16S = CNNB_U16S + 64

TDB_PROJ = 0
TDB_LAYER0 = TDB_PROJ + 24 * 2 * 2 * 512
TDBL_QKV = 0
TDBL_OUT = TDBL_QKV + 4 * 6 * 2 * 512
TDBL_FF1 = TDBL_OUT + 4 * 2 * 2 * 512
TDBL_FF2 = TDBL_FF1 + 4 * 2 * 2 * 512
TDBL_U16S = TDBL_FF2 + 4 * 2 * 2 * 512
PLB_U16S = 4 * 4 * 2 * 512
# three-term fragments (csrc/td_bf16x6.hip)
TDX_PROJ = 0
TDX_LAYER0 = TDX_PROJ + 24 * 2 * 3 * 512
TDXL_QKV = 0
TDXL_OUT = TDXL_QKV + 4 * 6 * 3 * 512
TDXL_FF1 = TDXL_OUT + 4 * 2 * 3 * 512
TDXL_FF2 = TDXL_FF1 + 4 * 2 * 3 * 512
TDXL_U16S = TDXL_FF2 + 4 * 2 * 3 * 512
PLX_U16S = 4 * 4 * 3 * 512

PL_W1_AF = 0
PL_B1 = PL_W1_AF + 8 * 4 * 256
PL_W2 = PL_B1 + 128
PL_W3 = PL_W2 + 128
PL_B2 = PL_W3 + 64
PL_FLOATS = PL_B2 + 4


def _np(sd, key):
    v = sd[key]
    if hasattr(v, 'detach'):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float64)


_LANE = np.arange(64)


def conv_b_fragments(wf):
    """wf [cout][cin][3][3] (already BN-scaled) -> flat [9][S][NT][64][4] float32."""
    cout, cin = wf.shape[:2]
    S, NT = cin // 8, cout // 32
    w9 = wf.reshape(cout, cin, 9)
    tap = np.arange(9)[:, None, None, None, None]
    s = np.arange(S)[None, :, None, None, None]
    nt = np.arange(NT)[None, None, :, None, None]
    lane = _LANE[None, None, None, :, None]
    kk = np.arange(4)[None, None, None, None, :]
    n = (lane & 31) + 32 * nt
    c = 8 * s + 4 * (lane >> 5) + kk
    return w9[n, c, tap].astype(np.float32).reshape(-1)


def linear_a_fragments(w):
    """w [rows][K] -> flat [K/8][rows/32][64][4] float32."""
    rows, K = w.shape
    S, MT = K // 8, rows // 32
    s = np.arange(S)[:, None, None, None]
    mt = np.arange(MT)[None, :, None, None]
    lane = _LANE[None, None, :, None]
    kk = np.arange(4)[None, None, None, :]
    return w[(lane & 31) + 32 * mt, 8 * s + 4 * (lane >> 5) + kk].astype(np.float32).reshape(-1)


def fold_bn(sd, pfx, i):
    w = _np(sd, pfx + 'conv%d.weight' % i)
    b = _np(sd, pfx + 'conv%d.bias' % i)
    scale = _np(sd, pfx + 'bn%d.weight' % i) / np.sqrt(_np(sd, pfx + 'bn%d.running_var' % i) + BN_EPS)
    t = (b - _np(sd, pfx + 'bn%d.running_mean' % i)) * scale + _np(sd, pfx + 'bn%d.bias' % i)
    return w * scale[:, None, None, None], t


def pack_adapt_cnn(sd, pfx='cnn.model.'):
    """AdaptCNN (cnn_model='adapt', 16/32/64 channels, 3x3 kernels, no fc) -> float32 [CNN_W_FLOATS]."""
    shapes = [(16, 1), (32, 16), (64, 32), (64, 64), (64, 64), (64, 64)]
    for i, (co, ci) in enumerate(shapes, 1):
        if tuple(sd[pfx + 'conv%d.weight' % i].shape) != (co, ci, 3, 3):
            raise NotImplementedError('HIP AdaptCNN kernel is built for the nisqa.tar geometry; conv%d is %s'
                                      % (i, tuple(sd[pfx + 'conv%d.weight' % i].shape)))
    if pfx + 'fc.weight' in sd:
        raise NotImplementedError('cnn_fc_out_h is not supported by the HIP AdaptCNN kernel')
    blob = np.zeros(CNN_W_FLOATS, np.float32)
    w, t = fold_bn(sd, pfx, 1)
    blob[CNN_W1:CNN_W1 + 144] = w.reshape(16, 9).astype(np.float32).reshape(-1)
    blob[CNN_T1:CNN_T1 + 16] = t
    for i, (wo, to) in zip(range(2, 7), [(CNN_WF2, CNN_T2), (CNN_WF3, CNN_T3), (CNN_WF4, CNN_T4),
                                          (CNN_WF5, CNN_T5), (CNN_WF6, CNN_T6)]):
        w, t = fold_bn(sd, pfx, i)
        fr = conv_b_fragments(w)
        blob[wo:wo + fr.size] = fr
        blob[to:to + t.size] = t
    return blob


def pack_self_att(sd, n_layers, pfx='time_dependency.model.'):
    """SelfAttention (d_model 64, 1 head, sa_h 64, input 384) -> float32 blob."""
    if tuple(sd[pfx + 'linear.weight'].shape) != (64, 384):
        raise NotImplementedError('HIP self-attention kernel needs Linear 384->64')
    blob = np.zeros(TD_LAYER0 + n_layers * TDL_FLOATS, np.float32)

    def put(off, arr):
        arr = np.asarray(arr, np.float32).reshape(-1)
        blob[off:off + arr.size] = arr

    put(TD_PROJ_AF, linear_a_fragments(_np(sd, pfx + 'linear.weight')))
    put(TD_PROJ_B, _np(sd, pfx + 'linear.bias'))
    put(TD_LN0_G, _np(sd, pfx + 'norm1.weight'))
    put(TD_LN0_B, _np(sd, pfx + 'norm1.bias'))
    for l in range(n_layers):
        p = pfx + 'layers.%d.' % l
        base = TD_LAYER0 + l * TDL_FLOATS
        if tuple(sd[p + 'self_attn.in_proj_weight'].shape) != (192, 64) or \
                tuple(sd[p + 'linear1.weight'].shape) != (64, 64):
            raise NotImplementedError('HIP self-attention kernel needs d_model=64, nhead=1, sa_h=64')
        put(base + TDL_QKV_AF, linear_a_fragments(_np(sd, p + 'self_attn.in_proj_weight')))
        put(base + TDL_QKV_B, _np(sd, p + 'self_attn.in_proj_bias'))
        put(base + TDL_OUT_AF, linear_a_fragments(_np(sd, p + 'self_attn.out_proj.weight')))
        put(base + TDL_OUT_B, _np(sd, p + 'self_attn.out_proj.bias'))
        put(base + TDL_LN1_G, _np(sd, p + 'norm1.weight'))
        put(base + TDL_LN1_B, _np(sd, p + 'norm1.bias'))
        put(base + TDL_FF1_AF, linear_a_fragments(_np(sd, p + 'linear1.weight')))
        put(base + TDL_FF1_B, _np(sd, p + 'linear1.bias'))
        put(base + TDL_FF2_AF, linear_a_fragments(_np(sd, p + 'linear2.weight')))
        put(base + TDL_FF2_B, _np(sd, p + 'linear2.bias'))
        put(base + TDL_LN2_G, _np(sd, p + 'norm2.weight'))
        put(base + TDL_LN2_B, _np(sd, p + 'norm2.bias'))
    return blob


def pack_pool_att(sd, head_prefixes):
    """PoolAttFF heads (64 -> 128 -> 1 scores, 64 -> 1 output) -> float32 [n_heads * PL_FLOATS]."""
    blob = np.zeros(len(head_prefixes) * PL_FLOATS, np.float32)
    for h, p in enumerate(head_prefixes):
        if tuple(sd[p + 'linear1.weight'].shape) != (128, 64):
            raise NotImplementedError('HIP pooling kernel needs pool_att_h=128 on d=64')
        base = h * PL_FLOATS
        fr = linear_a_fragments(_np(sd, p + 'linear1.weight'))
        blob[base + PL_W1_AF: base + PL_W1_AF + fr.size] = fr
        blob[base + PL_B1: base + PL_B1 + 128] = _np(sd, p + 'linear1.bias')
        blob[base + PL_W2: base + PL_W2 + 128] = _np(sd, p + 'linear2.weight').reshape(-1)
        blob[base + PL_W3: base + PL_W3 + 64] = _np(sd, p + 'linear3.weight').reshape(-1)
        blob[base + PL_B2] = _np(sd, p + 'linear2.bias').reshape(-1)[0]
        blob[base + PL_B2 + 1] = _np(sd, p + 'linear3.bias').reshape(-1)[0]
    return blob


# ---- split-bf16 fragments (csrc/cnn_bf16.hip) ---------------------------------------------------
def bf16_bits(x):
    """round-to-nearest-even float32 -> bf16 bit patterns (uint16), same formula as the device code"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_val(b):
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def bf16_split(x, terms=2):
    """x (float32) -> list of `terms` uint16 arrays with x ~= sum(bf16_val(part))"""
    r = np.asarray(x, dtype=np.float32)
    out = []
    for _ in range(terms):
        b = bf16_bits(r)
        out.append(b)
        r = (r - bf16_val(b)).astype(np.float32)
    return out


def conv_b_fragments_bf16(wf, terms=2, split=None):
    """wf [cout][cin][3][3] float32 (BN-scaled) -> uint16 [9*cin/16][NT][terms][64][8]; split: bf16_split (default) or f16_split"""
    cout, cin = wf.shape[:2]
    S16, NT = cin // 16, cout // 32
    w9 = np.asarray(wf, np.float32).reshape(cout, cin, 9)
    g = np.arange(9 * S16)[:, None, None, None]
    nt = np.arange(NT)[None, :, None, None]
    lane = _LANE[None, None, :, None]
    e = np.arange(8)[None, None, None, :]
    vals = w9[(lane & 31) + 32 * nt, 16 * (g % S16) + 8 * (lane >> 5) + e, g // S16]      # [G][NT][64][8]
    return np.stack((split or bf16_split)(vals, terms), 2).reshape(-1)                     # [G][NT][terms][64][8]


def conv_b_fragments_bf16_nsplit(wf, terms=2, split=None):
    """conv5/conv6 fragments for the N-split 16x16x32 form: uint16 [4 waves][18 steps][terms][64][8],
    value = W[n = 16*w + (lane&15)][c = 32*(g&1) + 8*(lane>>4) + e][tap = g>>1]"""
    w9 = np.asarray(wf, np.float32).reshape(64, 64, 9)
    w = np.arange(4)[:, None, None, None]
    g = np.arange(18)[None, :, None, None]
    lane = _LANE[None, None, :, None]
    e = np.arange(8)[None, None, None, :]
    vals = w9[16 * w + (lane & 15), 32 * (g & 1) + 8 * (lane >> 4) + e, g >> 1]             # [4][18][64][8]
    return np.stack((split or bf16_split)(vals, terms), 2).reshape(-1)


def pack_adapt_cnn_bf16(sd, pfx='cnn.model.', conv1_pairs=False, terms=2):
    """bf16 hi/lo weight fragments for cnn_front_bf16_kernel -> uint16 [CNNB_U16S] (biases: pack_adapt_cnn); terms = 3:
    hi/mid/lo fragments for cnn_front_bf16x6_kernel -> uint16 [CNNX_U16S] (three bf16 terms hold an fp32 weight exactly).

    conv1 B operand [k][n], lane (n = lane & 31, h = lane >> 5) holds k-slots 8h .. 8h+7.  Plain layout (StandardCNN
    kernel): k = tap ky*3 + kx, n = channel.  conv1_pairs (AdaptCNN kernel): a row is a pair of mel-adjacent output
    pixels, n = c + 16*dm, k = 4*kx + dmm over the 4 mels the pair touches: w[c][ky = dmm - dm][kx]."""
    if terms not in (2, 3):
        raise ValueError('terms must be 2 or 3')
    offs = [CNNB_W2, CNNB_W3, CNNB_W4, CNNB_W5, CNNB_W6] if terms == 2 else [CNNX_W2, CNNX_W3, CNNX_W4, CNNX_W5, CNNX_W6]
    blob = np.zeros(CNNB_U16S if terms == 2 else CNNX_U16S, np.uint16)
    w, _ = fold_bn(sd, pfx, 1)
    w1 = w.reshape(16, 9).astype(np.float32)
    full = np.zeros((64, 8), np.float32)
    for lane in range(64):
        j, h = lane & 31, lane >> 5
        for e in range(8):
            k = 8 * h + e
            if conv1_pairs:
                c, dm, kx, dmm = j & 15, j >> 4, k >> 2, k & 3
                if k < 12 and 0 <= dmm - dm <= 2:
                    full[lane, e] = w1[c, (dmm - dm) * 3 + kx]
            elif j < 16 and k < 9:
                full[lane, e] = w1[j, k]
    for t, part in enumerate(bf16_split(full, 3)):
        blob[CNNB_W1 + t * 512: CNNB_W1 + (t + 1) * 512] = part.reshape(-1)
    for i, off in zip(range(2, 7), offs):
        w, _ = fold_bn(sd, pfx, i)
        fr = conv_b_fragments_bf16_nsplit(w.astype(np.float32), terms) if i >= 5 else conv_b_fragments_bf16(w.astype(np.float32), terms)
        blob[off:off + fr.size] = fr
    return blob


# ---- two-term f16 fragments (csrc/cnn_bf16.hip, formats F16X3 / F16X4) ------------------------------------------------------
def f16_split(x, terms=2):
    """x (float32, |x| < 65504) -> list of `terms` uint16 arrays of f16 bit patterns with x ~= sum(parts): each term is the
    round-to-nearest-even f16 of the running remainder (numpy's conversion = v_cvt_pk_f16_f32's).  Two terms hold 11 + 11
    significand bits and the second term's sign: the fp32 value itself unless its remainder is an odd multiple of ulp32(x) beyond
    2048 ulp32 -- then the pair is one fp32 ulp off (about a quarter of random values)."""
    r = np.asarray(x, dtype=np.float32)
    out = []
    for _ in range(terms):
        h = r.astype(np.float16)
        out.append(h.view(np.uint16))
        r = (r - h.astype(np.float32)).astype(np.float32)
    return out


def f16_val(b):
    return np.asarray(b, dtype=np.uint16).view(np.float16).astype(np.float32)


def pack_adapt_cnn_f16(sd, pfx='cnn.model.'):
    """f16 hi/lo weight fragments for cnn_front_f16_kernel -> uint16 [CNNH_U16S]: the layout of pack_adapt_cnn_bf16(conv1_pairs=True,
    terms=2) with f16 bit patterns of W_l * 2^kw_l (kw_l: the layer's largest |W| lands in [2^14, 2^15), so even weights 2^-24 of it
    keep an absolute precision of 2^-39 of the largest), followed by the per-layer constants the kernel derives its activation
    scales from (layout.hpp CNNH_META): int32 kw[6], float32 G[6] = max over output channels of sum |W| (BatchNorm folded, rounded
    up), float32 T[6] = max |shift|.  |layer output| <= (largest |layer input|) * G + T."""
    blob = np.zeros(CNNH_U16S, np.uint16)
    meta_i = np.zeros(32, np.int32)
    meta_f = meta_i.view(np.float32)
    folded = [fold_bn(sd, pfx, i) for i in range(1, 7)]
    for l, (w, t) in enumerate(folded):
        w32 = np.asarray(w, np.float32)
        wmax = float(np.abs(w32).max())
        kw = 0 if wmax == 0 or not np.isfinite(wmax) else 15 - int(np.frexp(wmax)[1])        # wmax * 2^kw in [2^14, 2^15)
        kw = int(min(max(kw, -60), 60))
        meta_i[l] = kw
        meta_f[8 + l] = np.float32(np.abs(w32.astype(np.float64)).reshape(w32.shape[0], -1).sum(1).max() * (1 + 1e-6))
        meta_f[16 + l] = np.float32(np.abs(np.asarray(t, np.float32)).max() * (1 + 1e-6))
    w1 = np.ldexp(np.asarray(folded[0][0], np.float32).reshape(16, 9), int(meta_i[0])).astype(np.float32)
    full = np.zeros((64, 8), np.float32)
    for lane in range(64):
        j, h = lane & 31, lane >> 5
        for e in range(8):
            k = 8 * h + e
            c, dm, kx, dmm = j & 15, j >> 4, k >> 2, k & 3
            if k < 12 and 0 <= dmm - dm <= 2:
                full[lane, e] = w1[c, (dmm - dm) * 3 + kx]
    for t, part in enumerate(f16_split(full, 2)):
        blob[CNNB_W1 + t * 512: CNNB_W1 + (t + 1) * 512] = part.reshape(-1)
    for i, off in zip(range(2, 7), [CNNB_W2, CNNB_W3, CNNB_W4, CNNB_W5, CNNB_W6]):
        ws = np.ldexp(np.asarray(folded[i - 1][0], np.float32), int(meta_i[i - 1])).astype(np.float32)
        fr = conv_b_fragments_bf16_nsplit(ws, 2, split=f16_split) if i >= 5 else conv_b_fragments_bf16(ws, 2, split=f16_split)
        blob[off:off + fr.size] = fr
    blob[CNNH_META:CNNH_META + 64] = meta_i.view(np.uint16)
    return blob


# ---- nisqa_tts.tar architecture ----------------------------------------------------------------------
def pack_standard_cnn(sd, pfx='cnn.model.'):
    """StandardCNN (16/32/64 channels, 3x3 kernels, fc_out 768 -> 20) -> float32 [CNNS_W_FLOATS]."""
    shapes = [(16, 1), (32, 16), (64, 32), (64, 64), (64, 64), (64, 64)]
    for i, (co, ci) in enumerate(shapes, 1):
        if tuple(sd[pfx + 'conv%d.weight' % i].shape) != (co, ci, 3, 3):
            raise NotImplementedError('HIP StandardCNN kernel is built for the nisqa_tts.tar geometry')
    if pfx + 'fc_out.weight' not in sd or tuple(sd[pfx + 'fc_out.weight'].shape) != (20, 768):
        raise NotImplementedError('HIP StandardCNN kernel needs cnn_fc_out_h=20')
    blob = np.zeros(CNNS_W_FLOATS, np.float32)
    w, t = fold_bn(sd, pfx, 1)
    blob[CNN_W1:CNN_W1 + 144] = w.reshape(16, 9).astype(np.float32).reshape(-1)
    blob[CNN_T1:CNN_T1 + 16] = t
    for i, (wo, to) in zip(range(2, 7), [(CNN_WF2, CNN_T2), (CNN_WF3, CNN_T3), (CNN_WF4, CNN_T4),
                                          (CNN_WF5, CNN_T5), (CNN_WF6, CNN_T6)]):
        w, t = fold_bn(sd, pfx, i)
        fr = conv_b_fragments(w)
        blob[wo:wo + fr.size] = fr
        blob[to:to + t.size] = t
    fc = _np(sd, pfx + 'fc_out.weight').reshape(20, 64, 12)            # [j][c][pixel]  (flatten c*12 + y*2 + x, NL:830)
    blob[CNNS_FC_W:CNNS_FC_W + 768 * 20] = np.transpose(fc, (2, 1, 0)).reshape(-1)    # [pixel*64 + c][j]
    blob[CNNS_FC_B:CNNS_FC_B + 20] = _np(sd, pfx + 'fc_out.bias')
    return blob


def pack_lstm_laststep(sd, lpfx='time_dependency.model.lstm.', ppfx='pool.model.'):
    """nn.LSTM(20, 128, bidirectional) + PoolLastStepBi linear 256 -> 1 -> float32 [LSTM_W_FLOATS]."""
    if tuple(sd[lpfx + 'weight_ih_l0'].shape) != (512, 20) or tuple(sd[lpfx + 'weight_hh_l0'].shape) != (512, 128) \
            or lpfx + 'weight_ih_l0_reverse' not in sd or lpfx + 'weight_ih_l1' in sd:
        raise NotImplementedError('HIP LSTM kernel needs one bidirectional layer, input 20, hidden 128')
    blob = np.zeros(LSTM_W_FLOATS, np.float32)
    for d, sfx in enumerate(('', '_reverse')):
        base = d * LSTM_DIR_FLOATS
        blob[base + LSTM_WIH: base + LSTM_WIH + 512 * 20] = _np(sd, lpfx + 'weight_ih_l0' + sfx).reshape(-1)
        blob[base + LSTM_WHH: base + LSTM_WHH + 512 * 128] = _np(sd, lpfx + 'weight_hh_l0' + sfx).reshape(-1)
        blob[base + LSTM_B: base + LSTM_B + 512] = (_np(sd, lpfx + 'bias_ih_l0' + sfx) + _np(sd, lpfx + 'bias_hh_l0' + sfx))
    blob[LSTM_POOL_W: LSTM_POOL_W + 256] = _np(sd, ppfx + 'linear.weight').reshape(-1)
    blob[LSTM_POOL_W + 256] = _np(sd, ppfx + 'linear.bias').reshape(-1)[0]
    return blob


# ---- split-bf16 fragments of the linear layers (csrc/td_bf16.hip) ---------------------------------------------
def linear_a_fragments_bf16(w, chain, terms=2):
    """w [rows][K] -> uint16 [K/16][rows/32][terms][64][8].  k-slot e of lane half h = column 16s + 8h + e (natural: the B
    operand is read from memory) or 16s + (e&3) + 8(e>>2) + 4h (chain: the B operand is the previous D fragment)."""
    w = np.asarray(w, np.float32)
    rows, K = w.shape
    s = np.arange(K // 16)[:, None, None, None]
    mt = np.arange(rows // 32)[None, :, None, None]
    lane = _LANE[None, None, :, None]
    e = np.arange(8)[None, None, None, :]
    col = 16 * s + ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) if chain else 8 * (lane >> 5) + e)
    vals = w[(lane & 31) + 32 * mt, col]
    return np.stack(bf16_split(vals, terms), 2).reshape(-1)


def linear_a_fragments_bf16_t16(w, chain, terms=3):
    """w [rows][K] -> uint16 [rows/64][K/32][4][terms][64][8]: 16 x 32 fragments of v_mfma_f32_16x16x32_bf16 for the 16-token tiles
    of csrc/td16_bf16x6.hip, one 64-row GEMM block after the other.  Lane l = (i = l & 15, g = l >> 4) of fragment (s, mt) holds row
    64 blk + 16 mt + i and, in k-slot e, column 32 s + 8 g + e (natural: the other operand is read from memory) or
    16 (2 s + (e >> 2)) + 4 g + (e & 3) (chain: the other operand is the previous GEMM's D tiles, feature 16 mt + 4 g + r in
    register r of tile mt)."""
    w = np.asarray(w, np.float32)
    rows, K = w.shape
    blk = np.arange(rows // 64)[:, None, None, None, None]
    s = np.arange(K // 32)[None, :, None, None, None]
    mt = np.arange(4)[None, None, :, None, None]
    lane = _LANE[None, None, None, :, None]
    e = np.arange(8)[None, None, None, None, :]
    g = lane >> 4
    col = 16 * (2 * s + (e >> 2)) + 4 * g + (e & 3) if chain else 32 * s + 8 * g + e
    vals = w[64 * blk + 16 * mt + (lane & 15), col]
    return np.stack(bf16_split(vals, terms), 3).reshape(-1)


def pack_self_att_bf16(sd, n_layers, pfx='time_dependency.model.', terms=2):
    """terms = 2: hi / lo fragments of td_bf16.hip (32 x 16 fragments, TDB_* offsets); terms = 3: hi / mid / lo fragments of
    td16_bf16x6.hip (16 x 32 fragments for its 16-token tiles; the TDX_* offsets: the blocks hold the same number of elements)"""
    if terms == 2:
        PROJ, LAYER0, QKV, OUT, FF1, FF2, LSZ = TDB_PROJ, TDB_LAYER0, TDBL_QKV, TDBL_OUT, TDBL_FF1, TDBL_FF2, TDBL_U16S
        frag = lambda w, chain: linear_a_fragments_bf16(w, chain=chain, terms=2)
    else:
        PROJ, LAYER0, QKV, OUT, FF1, FF2, LSZ = TDX_PROJ, TDX_LAYER0, TDXL_QKV, TDXL_OUT, TDXL_FF1, TDXL_FF2, TDXL_U16S
        frag = lambda w, chain: linear_a_fragments_bf16_t16(w, chain=chain, terms=3)
    blob = np.zeros(LAYER0 + n_layers * LSZ, np.uint16)

    def put(off, fr):
        blob[off:off + fr.size] = fr

    put(PROJ, frag(_np(sd, pfx + 'linear.weight'), False))
    for l in range(n_layers):
        p = pfx + 'layers.%d.' % l
        base = LAYER0 + l * LSZ
        put(base + QKV, frag(_np(sd, p + 'self_attn.in_proj_weight'), True))
        put(base + OUT, frag(_np(sd, p + 'self_attn.out_proj.weight'), True))
        put(base + FF1, frag(_np(sd, p + 'linear1.weight'), True))
        put(base + FF2, frag(_np(sd, p + 'linear2.weight'), True))
    return blob


def pack_pool_att_bf16(sd, head_prefixes, terms=2):
    """terms = 3: the blocks of pool_score_bf16x6_kernel followed by pack_pool_att_t16 (the pooling tail of td16_layer_kernel)"""
    sz = PLB_U16S if terms == 2 else PLX_U16S
    blob = np.zeros(len(head_prefixes) * sz, np.uint16)
    for h, p in enumerate(head_prefixes):
        fr = linear_a_fragments_bf16(_np(sd, p + 'linear1.weight'), chain=True, terms=terms)
        blob[h * sz: h * sz + fr.size] = fr
    return blob if terms == 2 else np.concatenate([blob, pack_pool_att_t16(sd, head_prefixes)])


def pack_pool_att_t16(sd, head_prefixes):
    """PoolAttFF heads for the pooling tail of csrc/td16_bf16x6.hip: a head's linear1 [128][64] as two 64-row blocks of
    linear_a_fragments_bf16_t16 (24 fragments of 1 KB each), all blocks first, then one 1 KB block of float32 per 64-row block:
    b1[64] | w2[64] | w3[64] | b2, b3, 0 ... (the second block of a head repeats w3 / b2 / b3) -> uint16 [2 H * 12288 + 2 H * 512]"""
    H = len(head_prefixes)
    frags = np.zeros((2 * H, 24 * 512), np.uint16)
    par = np.zeros((2 * H, 256), np.float32)
    for h, p in enumerate(head_prefixes):
        w1 = _np(sd, p + 'linear1.weight')
        if tuple(w1.shape) != (128, 64):
            raise NotImplementedError('HIP pooling kernel needs pool_att_h=128 on d=64')
        fr = linear_a_fragments_bf16_t16(w1, chain=True, terms=3).reshape(2, -1)
        for j in range(2):
            frags[2 * h + j] = fr[j]
            par[2 * h + j, 0:64] = _np(sd, p + 'linear1.bias')[64 * j:64 * j + 64]
            par[2 * h + j, 64:128] = _np(sd, p + 'linear2.weight').reshape(-1)[64 * j:64 * j + 64]
            par[2 * h + j, 128:192] = _np(sd, p + 'linear3.weight').reshape(-1)
            par[2 * h + j, 192] = _np(sd, p + 'linear2.bias').reshape(-1)[0]
            par[2 * h + j, 193] = _np(sd, p + 'linear3.bias').reshape(-1)[0]
    return np.concatenate([frags.reshape(-1), par.reshape(-1).view(np.uint16)])

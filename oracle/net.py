"""Oracle (test infrastructure, NOT product code): CPU fp32 restatement of the
reference network half of the predict path, driven directly by an unchanged
``model_state_dict``.

Pinned: ``tests/test_oracle.py`` checks every function here against fixtures
written by the reference's own torch modules (``tests/golden/make_golden.py``).

Each function cites the reference lines it restates (NL = nisqa/NISQA_lib.py).
All arithmetic is float32 torch on CPU, eval-mode semantics (no dropout,
BatchNorm on running statistics).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
LN_EPS = 1e-5


def _t(x):
    return x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))


# -- segmenting ---------------------------------------------------------------

def segment_specs(spec, seg_length=15, seg_hop=4, max_length=None):
    """NL:2239-2282.  spec [n_mels, T] -> x [n_wins(,max_length), 1, n_mels, seg_length], n_wins."""
    if seg_length % 2 == 0:
        raise ValueError('seg_length must be odd! (seg_lenth={})'.format(seg_length))
    spec = _t(spec).float()
    n_mels, T = spec.shape
    n_full = T - (seg_length - 1)
    if n_full < 1:
        raise ValueError('Sample too short. Only {} windows available but seg_length={}.'.format(T, seg_length))
    starts = torch.arange(0, n_full, seg_hop)
    n_wins = int(math.ceil(n_full / seg_hop)) if seg_hop > 1 else n_full
    assert len(starts) == n_wins
    x = torch.stack([spec[:, s:s + seg_length] for s in starts.tolist()], 0).unsqueeze(1)
    if max_length is not None:
        if max_length < n_wins:
            raise ValueError('n_wins {} > max_length {}. Increase max window length ms_max_segments!'.format(n_wins, max_length))
        pad = torch.zeros((max_length, 1, n_mels, seg_length))
        pad[:n_wins] = x
        x = pad
    return x, n_wins


def n_wins_of(T, seg_length=15, seg_hop=4):
    """Segment count segment_specs produces for a T-frame spectrogram (NL:2256, NL:2271-2273)."""
    return int(math.ceil((T - (seg_length - 1)) / seg_hop))


# -- AdaptCNN -------------------------------------------------------------------

def _conv_bn_relu(sd, pfx, i, x, padding):
    """conv_i -> bn_i (running stats) -> relu  (NL:690-705)."""
    x = F.conv2d(x, sd[pfx + 'conv%d.weight' % i], sd[pfx + 'conv%d.bias' % i], padding=padding)
    x = F.batch_norm(x, sd[pfx + 'bn%d.running_mean' % i], sd[pfx + 'bn%d.running_var' % i],
                     sd[pfx + 'bn%d.weight' % i], sd[pfx + 'bn%d.bias' % i], False, 0.0, BN_EPS)
    return F.relu(x)


def adapt_cnn(sd, x, pool_1=(24, 7), pool_2=(12, 5), pool_3=(6, 3), pfx='cnn.model.',
              return_stages=False):
    """AdaptCNN.forward, eval mode, fc_out_h=None (NL:688-710).  x [N,1,48,15] -> [N,384]."""
    st = {}
    x = _conv_bn_relu(sd, pfx, 1, x, (1, 1))
    x = F.adaptive_max_pool2d(x, tuple(pool_1)); st['p1'] = x
    x = _conv_bn_relu(sd, pfx, 2, x, (1, 1))
    x = F.adaptive_max_pool2d(x, tuple(pool_2)); st['p2'] = x
    x = _conv_bn_relu(sd, pfx, 3, x, (1, 1)); st['c3'] = x
    x = _conv_bn_relu(sd, pfx, 4, x, (1, 1))
    x = F.adaptive_max_pool2d(x, tuple(pool_3)); st['p3'] = x
    x = _conv_bn_relu(sd, pfx, 5, x, (1, 1)); st['c5'] = x
    x = _conv_bn_relu(sd, pfx, 6, x, (1, 0))
    x = x.reshape(x.shape[0], -1)
    if pfx + 'fc.weight' in sd:
        x = F.linear(x, sd[pfx + 'fc.weight'], sd[pfx + 'fc.bias'])
    return (x, st) if return_stages else x


def standard_cnn(sd, x, pfx='cnn.model.'):
    """StandardCNN.forward, eval mode (NL:811-836).  x [N,1,48,15] -> [N,768] or fc_out."""
    x = _conv_bn_relu(sd, pfx, 1, x, 1)
    x = F.max_pool2d(x, 2, stride=2, padding=(0, 1))
    x = _conv_bn_relu(sd, pfx, 2, x, 1)
    x = F.max_pool2d(x, 2, stride=2)
    x = _conv_bn_relu(sd, pfx, 3, x, 1)
    x = _conv_bn_relu(sd, pfx, 4, x, 1)
    x = F.max_pool2d(x, 2, stride=2)
    x = _conv_bn_relu(sd, pfx, 5, x, 1)
    x = _conv_bn_relu(sd, pfx, 6, x, 1)
    x = x.reshape(x.shape[0], -1)
    if pfx + 'fc_out.weight' in sd:
        x = F.linear(x, sd[pfx + 'fc_out.weight'], sd[pfx + 'fc_out.bias'])
    return x


# -- self-attention (one clip, no padding: key mask is then a no-op) --------------

def self_attention(sd, feat, num_layers=2, pfx='time_dependency.model.'):
    """SelfAttention.forward + SelfAttentionLayer.forward for ONE clip (NL:988-996, NL:1025-1040).

    feat [L, 384] holds exactly the clip's n_wins valid rows, so the key-padding
    mask (NL:1027-1029) masks nothing; the reference's per-clip result does not
    depend on the padded rows (checked <=5e-7).
    nhead=1, d=64, post-norm, ReLU FFN, identity positional encoding.
    """
    x = F.linear(feat, sd[pfx + 'linear.weight'], sd[pfx + 'linear.bias'])
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd[pfx + 'norm1.weight'], sd[pfx + 'norm1.bias'], LN_EPS)
    for l in range(num_layers):
        p = pfx + 'layers.%d.' % l
        qkv = F.linear(x, sd[p + 'self_attn.in_proj_weight'], sd[p + 'self_attn.in_proj_bias'])
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        att = torch.softmax((q * (1.0 / math.sqrt(d))) @ k.t(), dim=-1) @ v
        att = F.linear(att, sd[p + 'self_attn.out_proj.weight'], sd[p + 'self_attn.out_proj.bias'])
        x = F.layer_norm(x + att, (d,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], LN_EPS)
        h = F.linear(F.relu(F.linear(x, sd[p + 'linear1.weight'], sd[p + 'linear1.bias'])),
                     sd[p + 'linear2.weight'], sd[p + 'linear2.bias'])
        x = F.layer_norm(x + h, (d,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], LN_EPS)
    return x


def pool_att_ff(sd, x, pfx):
    """PoolAttFF.forward for ONE clip of valid rows (NL:1171-1183).  x [L,64] -> [1]."""
    att = F.linear(F.relu(F.linear(x, sd[pfx + 'linear1.weight'], sd[pfx + 'linear1.bias'])),
                   sd[pfx + 'linear2.weight'], sd[pfx + 'linear2.bias'])     # [L,1]
    att = torch.softmax(att.t(), dim=1)                                        # [1,L]
    pooled = att @ x                                                           # [1,64]
    return F.linear(pooled, sd[pfx + 'linear3.weight'], sd[pfx + 'linear3.bias']).reshape(-1)


# -- LSTM / last-step pooling (nisqa_tts.tar path) ---------------------------------

def bilstm(sd, x, pfx='time_dependency.model.lstm.'):
    """nn.LSTM(batch_first, bidirectional, 1 layer) on one clip (NL:925-943).  x [L,I] -> [L,2H]."""
    def run(seq, sfx):
        w_ih, w_hh = sd[pfx + 'weight_ih_l0' + sfx], sd[pfx + 'weight_hh_l0' + sfx]
        b = sd[pfx + 'bias_ih_l0' + sfx] + sd[pfx + 'bias_hh_l0' + sfx]
        H = w_hh.shape[1]
        h = torch.zeros(H); c = torch.zeros(H); out = []
        for t in range(seq.shape[0]):
            g = w_ih @ seq[t] + w_hh @ h + b
            i, f, gg, o = torch.sigmoid(g[:H]), torch.sigmoid(g[H:2 * H]), torch.tanh(g[2 * H:3 * H]), torch.sigmoid(g[3 * H:])
            c = f * c + i * gg
            h = o * torch.tanh(c)
            out.append(h)
        return torch.stack(out, 0)
    fwd = run(x, '')
    bwd = run(x.flip(0), '_reverse').flip(0)
    return torch.cat([fwd, bwd], 1)


def pool_last_step_bi(sd, x, pfx='pool.model.'):
    """PoolLastStepBi for one clip of valid rows (NL:1107-1115)."""
    H = x.shape[1] // 2
    z = torch.cat([x[-1, :H], x[0, H:]], 0)
    return F.linear(z, sd[pfx + 'linear.weight'], sd[pfx + 'linear.bias']).reshape(-1)


# -- whole network on one clip -----------------------------------------------------

def predict_from_melspec(sd, args, spec, return_stages=False, dtype=torch.float32):
    """model.forward on one clip's [n_mels,T] dB spectrogram (NL:137-142 / NL:260-268).

    Returns float32 numpy [5] (NISQA_DIM: mos, noi, dis, col, loud; NL:1461-1465) or [1] (NISQA).
    dtype=torch.float64: the same operators in double precision (stages come back as float64) -- the yardstick the tests use
    to compare the ROUNDING error of the GPU precision modes; the reference itself computes in float32.
    """
    sd = {k: _t(v).to(dtype) for k, v in sd.items() if k.split('.')[-1] != 'num_batches_tracked'}
    with torch.no_grad():
        x, n_wins = segment_specs(spec, args['ms_seg_length'], args['ms_seg_hop_length'], None)
        x = x.to(dtype)
        if n_wins > args['ms_max_segments']:
            raise ValueError('n_wins {} > max_length {}. Increase max window length ms_max_segments!'.format(
                n_wins, args['ms_max_segments']))
        if args['cnn_model'] == 'adapt':
            feat = adapt_cnn(sd, x, args['cnn_pool_1'], args['cnn_pool_2'], args['cnn_pool_3'])
        elif args['cnn_model'] == 'standard':
            feat = standard_cnn(sd, x)
        else:
            raise NotImplementedError(args['cnn_model'])
        if args['td'] == 'self_att':
            td = self_attention(sd, feat, args['td_sa_num_layers'])
        elif args['td'] == 'lstm':
            td = bilstm(sd, feat)
        else:
            raise NotImplementedError(args['td'])
        if args['model'] == 'NISQA_DIM':
            out = torch.cat([pool_att_ff(sd, td, 'pool_layers.%d.model.' % h) for h in range(5)])
        elif args['pool'] == 'att':
            out = pool_att_ff(sd, td, 'pool.model.')
        elif args['pool'] == 'last_step_bi':
            out = pool_last_step_bi(sd, td)
        else:
            raise NotImplementedError(args['pool'])
    out = out.numpy().astype(np.float32)
    if return_stages:
        return out, {'feat': feat.numpy(), 'td': td.numpy(), 'n_wins': n_wins}
    return out

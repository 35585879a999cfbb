"""Oracle (test infrastructure, NOT product code): CPU fp32 restatement of ONE training step of the CNN-SA-AP
model -- ``model.train(); y_hat = model(x, n_wins); loss = biasLoss.get_loss(...); loss.backward(); opt.step()``
(reference nisqa/NISQA_model.py:131-152 / 330-352) -- as a torch-autograd function of an unchanged
``model_state_dict``.

Pinned: ``tests/test_oracle_train.py`` checks loss, every gradient, the BatchNorm running statistics and the
parameters after an Adam step against fixtures written by the reference's own modules in train mode
(``tests/golden/make_golden_train.py``; dropout probabilities set to 0 there because the reference's dropout masks
come from torch's global RNG inside its modules and cannot be injected).

Train-mode semantics restated (NL = nisqa/NISQA_lib.py):
  * Framewise packs the valid segments of the batch (NL:487-502): BatchNorm2d normalises with the statistics of
    ALL valid segments of the batch (biased variance) and updates running_mean / running_var with momentum 0.1
    (unbiased variance), num_batches_tracked += 1;
  * Dropout2d(cnn_dropout) after pool2, relu3, pool3, relu5 (NL:696-705): one Bernoulli per (segment, channel);
  * SelfAttentionLayer (NL:1025-1040): dropout on the attention probabilities (nn.MultiheadAttention), dropout1 on
    the attention output, dropout inside the FFN, dropout2 on the FFN output;
  * PoolAttFF (NL:1171-1183) with pool_att_dropout;
  * loss: mean squared error over the non-NaN targets, summed over the heads (NL:1946-1950, NISQA_model.py:341-347),
    optionally through the per-sample cubic bias mapping (NL:1880-1892);
  * torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) on every parameter (NISQA_model.py:96).
Dropout masks are explicit inputs here (dict name -> tensor already scaled by 1/(1-p)); absent = no dropout.
"""
import math

import torch
import torch.nn.functional as F

from . import net as onet

BN_MOMENTUM = 0.1


def param_keys(sd):
    """Trainable tensors of a state_dict (everything except BatchNorm buffers)."""
    return [k for k in sd if k.split('.')[-1] not in ('running_mean', 'running_var', 'num_batches_tracked')]


def _conv_bn_relu_train(sd, pfx, i, x, padding, stats):
    w, b = sd[pfx + 'conv%d.weight' % i], sd[pfx + 'conv%d.bias' % i]
    z = F.conv2d(x, w, b, padding=padding)
    stats['bn%d' % i] = (z.detach().mean((0, 2, 3)), z.detach().var((0, 2, 3), unbiased=False), z.numel() // z.shape[1])
    z = F.batch_norm(z, None, None, sd[pfx + 'bn%d.weight' % i], sd[pfx + 'bn%d.bias' % i], True, 0.0, onet.BN_EPS)
    return F.relu(z)


def adapt_cnn_train(sd, x, pools, masks, stats, pfx='cnn.model.'):
    """AdaptCNN.forward in train mode (NL:688-710).  x [S,1,48,15] -> [S,384]; masks 'cnn_d1'..'cnn_d4' [S,C,1,1]."""
    m = lambda k, t: t * masks[k] if masks and k in masks else t
    x = _conv_bn_relu_train(sd, pfx, 1, x, (1, 1), stats)
    x = F.adaptive_max_pool2d(x, tuple(pools[0]))
    x = _conv_bn_relu_train(sd, pfx, 2, x, (1, 1), stats)
    x = m('cnn_d1', F.adaptive_max_pool2d(x, tuple(pools[1])))
    x = m('cnn_d2', _conv_bn_relu_train(sd, pfx, 3, x, (1, 1), stats))
    x = _conv_bn_relu_train(sd, pfx, 4, x, (1, 1), stats)
    x = m('cnn_d3', F.adaptive_max_pool2d(x, tuple(pools[2])))
    x = m('cnn_d4', _conv_bn_relu_train(sd, pfx, 5, x, (1, 1), stats))
    x = _conv_bn_relu_train(sd, pfx, 6, x, (1, 0), stats)
    return x.reshape(x.shape[0], -1)


def self_attention_train(sd, feat, num_layers, masks, clip, pfx='time_dependency.model.'):
    """SelfAttention + layers for one clip of valid rows, with the train-mode dropouts (NL:988-996, 1025-1040).
    masks: 'td%d_p' [L,L] (attention probabilities), 'td%d_1', 'td%d_f', 'td%d_2' [L,d], keyed per clip."""
    g = lambda k, t: t * masks[(clip, k)] if masks and (clip, k) in masks else t
    x = F.linear(feat, sd[pfx + 'linear.weight'], sd[pfx + 'linear.bias'])
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd[pfx + 'norm1.weight'], sd[pfx + 'norm1.bias'], onet.LN_EPS)
    for l in range(num_layers):
        p = pfx + 'layers.%d.' % l
        qkv = F.linear(x, sd[p + 'self_attn.in_proj_weight'], sd[p + 'self_attn.in_proj_bias'])
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        prob = g('td%d_p' % l, torch.softmax((q * (1.0 / math.sqrt(d))) @ k.t(), dim=-1))
        att = F.linear(prob @ v, sd[p + 'self_attn.out_proj.weight'], sd[p + 'self_attn.out_proj.bias'])
        x = F.layer_norm(x + g('td%d_1' % l, att), (d,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], onet.LN_EPS)
        h = g('td%d_f' % l, F.relu(F.linear(x, sd[p + 'linear1.weight'], sd[p + 'linear1.bias'])))
        h = F.linear(h, sd[p + 'linear2.weight'], sd[p + 'linear2.bias'])
        x = F.layer_norm(x + g('td%d_2' % l, h), (d,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], onet.LN_EPS)
    return x


def forward_train(sd, args, segs, n_wins, masks=None):
    """segs [S,1,48,15]: the valid segments of the batch, clip after clip; n_wins [B].
    -> (y_hat [B, heads], BatchNorm batch statistics {bn_i: (mean, biased var, count)})."""
    stats = {}
    feat = adapt_cnn_train(sd, segs, (args['cnn_pool_1'], args['cnn_pool_2'], args['cnn_pool_3']), masks, stats)
    heads = ['pool_layers.%d.model.' % h for h in range(5)] if args['model'] == 'NISQA_DIM' else ['pool.model.']
    out, o = [], 0
    for b, n in enumerate(int(v) for v in n_wins):
        td = self_attention_train(sd, feat[o:o + n], args['td_sa_num_layers'], masks, b)
        out.append(torch.cat([onet.pool_att_ff(sd, td, pfx) for pfx in heads]))
        o += n
    return torch.stack(out), stats


def nan_mse_loss(y_hat, y, bias=None):
    """sum over heads of mean((y - y_hat)^2 over the non-NaN targets) (NL:1946-1950); bias [B,4] maps y_hat first."""
    loss = 0.0
    for h in range(y_hat.shape[1]):
        yh = y_hat[:, h]
        if bias is not None:
            yh = bias[:, 0] + bias[:, 1] * yh + bias[:, 2] * yh ** 2 + bias[:, 3] * yh ** 3
        ok = ~torch.isnan(y[:, h])
        loss = loss + torch.mean((y[ok, h] - yh[ok]) ** 2)
    return loss


def train_step(sd, args, segs, n_wins, y, lr=1e-3, adam=None, masks=None, bias=None):
    """One optimiser step.  sd: state_dict (tensors or arrays, float32); y [B, heads] (NaN = unlabelled).
    -> dict(loss, y_hat, grads{key}, sd (updated copy incl. BatchNorm buffers), adam (state for the next step))."""
    sd = {k: onet._t(v).clone() for k, v in sd.items()}
    keys = param_keys(sd)
    for k in keys:
        sd[k] = sd[k].float().requires_grad_(True)
    y_hat, stats = forward_train(sd, args, onet._t(segs).float(), n_wins, masks)
    loss = nan_mse_loss(y_hat, onet._t(y).float(), None if bias is None else onet._t(bias).float())
    grads = dict(zip(keys, torch.autograd.grad(loss, [sd[k] for k in keys])))
    adam = adam or {'step': 0, 'm': {k: torch.zeros_like(sd[k]) for k in keys}, 'v': {k: torch.zeros_like(sd[k]) for k in keys}}
    t = adam['step'] + 1
    new = {k: v.detach().clone() for k, v in sd.items()}
    m2, v2 = {}, {}
    for k in keys:                                                   # torch.optim.Adam, default hyper-parameters
        g = grads[k]
        m2[k] = 0.9 * adam['m'][k] + 0.1 * g
        v2[k] = 0.999 * adam['v'][k] + 0.001 * g * g
        denom = (v2[k].sqrt() / math.sqrt(1 - 0.999 ** t)) + 1e-8
        new[k] = new[k] - (lr / (1 - 0.9 ** t)) * m2[k] / denom
    for i in range(1, 7):                                            # BatchNorm2d buffers (momentum 0.1)
        mean, var, cnt = stats['bn%d' % i]
        p = 'cnn.model.bn%d.' % i
        new[p + 'running_mean'] = (1 - BN_MOMENTUM) * new[p + 'running_mean'].float() + BN_MOMENTUM * mean
        new[p + 'running_var'] = (1 - BN_MOMENTUM) * new[p + 'running_var'].float() + BN_MOMENTUM * var * (cnt / (cnt - 1))
        if p + 'num_batches_tracked' in new:
            new[p + 'num_batches_tracked'] = new[p + 'num_batches_tracked'] + 1
    return {'loss': float(loss.detach()), 'y_hat': y_hat.detach().numpy(), 'grads': {k: g.numpy() for k, g in grads.items()},
            'sd': new, 'adam': {'step': t, 'm': m2, 'v': v2}}

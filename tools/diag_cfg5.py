"""Diagnostic (round 4): per-tensor gradient error of the training step at configs[4] size against the float64 gradients of
the reference fixture, for the kernel variants the environment switches select."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers, make_golden_train as mk
from nisqa_amd import synth
from nisqa_amd.train import HipTrainer

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg5_mos'
prec = sys.argv[2] if len(sys.argv) > 2 else 'f32'
g = helpers.golden('train_%s.npz' % name)
if name == 'cfg5_mos':
    args, sd, heads = dict(synth.MOS_ARGS), synth.random_state_dict(int(g['seed_sd']), 'NISQA'), 1
else:
    a, s = helpers.load_checkpoint(helpers.find_weights('nisqa.tar'))
    args, sd, heads = dict(a), {k: v.numpy() for k, v in s.items()}, 5
args.update({'cnn_dropout': 0.0, 'td_sa_dropout': 0.0, 'pool_att_dropout': 0.0})
specs, y = mk.batch_cfg5(int(g['seed_batch']), int(g['n_clips']), heads)
tr = HipTrainer(args, sd, 'cuda:0', lr=1e-3, precision=prec)
loss = tr.step_spec(specs, y)
torch.cuda.synchronize()
print(name, prec, {k: v for k, v in os.environ.items() if k.startswith('NISQA_HIP_TRAIN')}, 'loss', float(loss), float(g['loss1_f64']))
for k, gr in tr.grads().items():
    a64, a32 = g['grad64/' + k], g['grad/' + k]
    sc = max(1e-3, float(np.abs(a64).max()))
    e = np.abs(gr.numpy() - a64)
    print('  %-55s max|g| %9.3e  hip-ref64 %.2e  ref32-ref64 %.2e  at %s' % (k, sc, e.max() / sc, np.abs(a32 - a64).max() / sc,
                                                                         np.unravel_index(e.argmax(), e.shape)))

if os.environ.get('NISQA_HIP_TRAIN_DEBUG') == '1':
    # the same step in float64 on the CPU with the intermediates kept: d loss / d z_i (pre-BatchNorm conv outputs) and
    # d loss / d (layer output), compared with the HIP step's dz_i / da_i
    import torch.nn.functional as F
    from oracle import net as onet, train as otrain
    sd64 = {k: torch.as_tensor(np.asarray(v)).double() if np.asarray(v).dtype.kind == 'f' else torch.as_tensor(np.asarray(v)) for k, v in sd.items()}
    for k in otrain.param_keys(sd64):
        sd64[k].requires_grad_(True)
    segs = torch.cat([onet.segment_specs(s_, 15, 4, None)[0] for s_ in specs]).double()
    n_wins = [int(v) for v in g['n_wins']]
    pools = (args['cnn_pool_1'], args['cnn_pool_2'], args['cnn_pool_3'])
    keep = {}
    x = segs
    pfx = 'cnn.model.'
    for i in range(1, 7):
        z = F.conv2d(x, sd64[pfx + 'conv%d.weight' % i], sd64[pfx + 'conv%d.bias' % i], padding=(1, 0) if i == 6 else (1, 1))
        z.retain_grad(); keep['z%d' % i] = z
        a = F.relu(F.batch_norm(z, None, None, sd64[pfx + 'bn%d.weight' % i], sd64[pfx + 'bn%d.bias' % i], True, 0.0, onet.BN_EPS))
        if i in (1, 2, 4):
            a = F.adaptive_max_pool2d(a, tuple(pools[{1: 0, 2: 1, 4: 2}[i]]))
        a.retain_grad(); keep['a%d' % i] = a
        x = a
    feat = x.reshape(x.shape[0], -1)
    heads_p = ['pool_layers.%d.model.' % h for h in range(5)] if args['model'] == 'NISQA_DIM' else ['pool.model.']
    out, o = [], 0
    for b, n in enumerate(n_wins):
        td = otrain.self_attention_train(sd64, feat[o:o + n], args['td_sa_num_layers'], None, b)
        out.append(torch.cat([onet.pool_att_ff(sd64, td, p_) for p_ in heads_p]))
        o += n
    loss64 = otrain.nan_mse_loss(torch.stack(out), torch.as_tensor(y).double())
    loss64.backward()
    print('float64 oracle loss', float(loss64))
    D = tr._debug
    for i in range(6, 1, -1):
        co = keep['z%d' % i].shape[1]
        want_dz = keep['z%d' % i].grad.permute(0, 2, 3, 1).reshape(-1, co).numpy()
        got_dz = D['dz%d' % i].cpu().numpy().reshape(-1, co)
        want_z = keep['z%d' % i].detach().permute(0, 2, 3, 1).reshape(-1, co).numpy()
        got_z = D['z%d' % i].cpu().numpy().reshape(-1, co)
        want_da = keep['a%d' % i].grad.permute(0, 2, 3, 1).reshape(-1, co).numpy()
        got_da = D['da%d' % i].cpu().numpy().reshape(-1, co)
        e = np.abs(got_dz - want_dz)
        r, c_ = np.unravel_index(e.argmax(), e.shape)
        nz_w, nz_g = (want_dz != 0), (got_dz != 0)
        print('layer %d: z max|d| %.2e (max|z| %.2e)  da max|d| %.2e (max %.2e)  dz max|d| %.2e (max %.2e) at row %d ch %d: hip %.6e ref %.6e; '
              'z there hip %.7f ref %.7f' % (i, np.abs(got_z - want_z).max(), np.abs(want_z).max(), np.abs(got_da - want_da).max(), np.abs(want_da).max(),
                                            e.max(), np.abs(want_dz).max(), r, c_, got_dz[r, c_], want_dz[r, c_], got_z[r, c_], want_z[r, c_]))
        big = np.argwhere(e > 0.05 * np.abs(want_dz).max())
        print('   entries with |d dz| > 5 %% of max: %d; per-channel count of those: %s' % (len(big), np.bincount(big[:, 1], minlength=co).tolist() if len(big) else []))

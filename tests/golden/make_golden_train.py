"""Golden vectors for oracle/train.py, produced by the REFERENCE's own modules in train mode
(/root/reference/nisqa/NISQA_lib.py NISQA / NISQA_DIM .forward, biasLoss._nan_mse, torch.optim.Adam as at
NISQA_model.py:96,131-152,341-352).

Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py
The reference is imported through oracle/ref_shim.py (empty librosa stand-in; only torch code runs).  Dropout
probabilities are set to 0: the reference draws its masks from torch's global RNG inside its modules, they cannot
be injected, and the oracle takes masks as explicit inputs instead (tested separately against plain autograd).
Stored per case: inputs' seeds, loss and y_hat of two consecutive steps, every gradient of the first, every state_dict
entry after the first and the BatchNorm buffers after the second.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_shim, net as onet                       # noqa: E402
from nisqa_amd import synth                                    # noqa: E402


def batch(seed, n_clips, heads):
    """Seeded spectrogram-like inputs: clips of ragged length as [48, T] dB-ish arrays, labels with one NaN."""
    rng = np.random.default_rng(seed)
    specs = [(-40 + 18 * rng.standard_normal((48, int(T)))).astype(np.float32) for T in rng.integers(15, 90, n_clips)]
    y = (1 + 4 * rng.random((n_clips, heads))).astype(np.float32)
    if n_clips > 2:
        y[1, heads - 1] = np.nan
    return specs, y


def run(name, args, seed_sd, seed_batch, n_clips, lr, checkpoint=None):
    """checkpoint: start from the reference's published weights (fine-tuning) instead of a seeded random state_dict; the
    fixture then holds the first step only (loss, y_hat, gradients) -- the weights themselves stay in the checkpoint."""
    NL = ref_shim.import_reference_lib()
    args = dict(args or {})
    if checkpoint is not None:
        ck = torch.load(checkpoint, map_location='cpu')
        args = dict(ck['args'])
        sd0 = {k: v.numpy() for k, v in ck['model_state_dict'].items()}
    args.update({'cnn_dropout': 0.0, 'td_sa_dropout': 0.0, 'pool_att_dropout': 0.0})
    if checkpoint is None:
        sd0 = synth.random_state_dict(seed_sd, args['model'])
    margs = {k: args[k] for k in ref_shim.MODEL_ARG_KEYS}
    model = {'NISQA': NL.NISQA, 'NISQA_DIM': NL.NISQA_DIM}[args['model']](**margs)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd0.items()}, strict=True)
    model.train()
    heads = 5 if args['model'] == 'NISQA_DIM' else 1
    specs, y = batch(seed_batch, n_clips, heads)
    L = args['ms_max_segments']
    xs, nw = zip(*[onet.segment_specs(s, args['ms_seg_length'], args['ms_seg_hop_length'], L) for s in specs])
    x, n_wins = torch.stack(xs), torch.tensor(nw)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    import pandas as pd
    loss_fn = NL.biasLoss(pd.Series(['db'] * n_clips), anchor_db=None, mapping=None, min_r=None, do_print=False)
    out = {'n_wins': np.array(nw), 'seed_sd': seed_sd, 'seed_batch': seed_batch, 'n_clips': n_clips, 'lr': lr}
    yt = torch.as_tensor(y)
    for step in (1, 2):
        y_hat = model(x, n_wins)
        loss = sum(loss_fn.get_loss(yt[:, h].view(-1, 1), y_hat[:, h].view(-1, 1), np.arange(n_clips)) for h in range(heads))
        loss.backward()
        out['loss%d' % step] = float(loss)
        out['y_hat%d' % step] = y_hat.detach().numpy()
        if step == 1:
            for k, p in model.named_parameters():
                out['grad/' + k] = p.grad.detach().numpy().copy()
        if checkpoint is not None:
            break
        opt.step()
        opt.zero_grad()
        for k, v in model.state_dict().items():
            if step == 1 or k.split('.')[-1].startswith(('running', 'num_batches')):    # step 2: buffers only (size)
                out['sd%d/%s' % (step, k)] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'train_%s.npz' % name), **out)
    print(name, 'loss', out['loss1'], out.get('loss2'), 'segments', int(sum(nw)))


if __name__ == '__main__':
    torch.manual_seed(0)
    run('mos', synth.MOS_ARGS, 8, 31, 4, 1e-3)
    run('dim', synth.DIM_ARGS, 7, 32, 3, 1e-3)
    if os.path.isfile('/root/reference/weights/nisqa.tar'):
        run('dim_real', None, -1, 33, 6, 1e-3, checkpoint='/root/reference/weights/nisqa.tar')

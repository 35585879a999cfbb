"""Shared test helpers: checkpoint lookup, seeded random state dicts, fixture loading."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# args of weights/nisqa.tar that the hot path reads (copied VALUES, read from the checkpoint;
# used to build random-weight models on machines where the checkpoint file is absent).
DIM_ARGS = {
    'model': 'NISQA_DIM', 'name': 'rand_dim',
    'ms_sr': None, 'ms_fmax': 20000, 'ms_n_fft': 4096, 'ms_hop_length': 0.01, 'ms_win_length': 0.02,
    'ms_n_mels': 48, 'ms_seg_length': 15, 'ms_seg_hop_length': 4, 'ms_max_segments': 1300,
    'cnn_model': 'adapt', 'cnn_c_out_1': 16, 'cnn_c_out_2': 32, 'cnn_c_out_3': 64,
    'cnn_kernel_size': (3, 3), 'cnn_dropout': 0.2, 'cnn_fc_out_h': None,
    'cnn_pool_1': [24, 7], 'cnn_pool_2': [12, 5], 'cnn_pool_3': [6, 3],
    'td': 'self_att', 'td_sa_d_model': 64, 'td_sa_nhead': 1, 'td_sa_pos_enc': False,
    'td_sa_num_layers': 2, 'td_sa_h': 64, 'td_sa_dropout': 0.1,
    'td_lstm_h': None, 'td_lstm_num_layers': None, 'td_lstm_dropout': None, 'td_lstm_bidirectional': None,
    'td_2': 'skip', 'td_2_sa_d_model': None, 'td_2_sa_nhead': None, 'td_2_sa_pos_enc': None,
    'td_2_sa_num_layers': None, 'td_2_sa_h': None, 'td_2_sa_dropout': None, 'td_2_lstm_h': None,
    'td_2_lstm_num_layers': None, 'td_2_lstm_dropout': None, 'td_2_lstm_bidirectional': None,
    'pool': 'att', 'pool_att_h': 128, 'pool_att_dropout': 0, 'tr_parallel': False,
    'dim': True, 'double_ended': False,
}
MOS_ARGS = dict(DIM_ARGS, model='NISQA', name='rand_mos', dim=False)


def find_weights(name='nisqa.tar'):
    """Real checkpoint if it can be found on this machine, else None.

    /root/reference exists only in the build container; ``__graft_entry__.build()``
    stages a copy under oracle/_ref/weights/ (git-ignored, travels with gpurun).
    """
    cands = [os.environ.get('NISQA_WEIGHTS_DIR', ''), os.path.join(ROOT, 'oracle', '_ref', 'weights'),
             '/root/reference/weights']
    for d in cands:
        p = os.path.join(d, name) if d else ''
        if p and os.path.isfile(p):
            return p
    return None


def random_state_dict(seed, model='NISQA_DIM'):
    """Seeded random-init weights with the exact key set / shapes of nisqa.tar (or nisqa_mos_only.tar).

    BatchNorm gets non-trivial running stats and a few NEGATIVE gammas (the BN fold must not
    assume positive scale); magnitudes keep activations O(1) through six conv layers.
    """
    g = torch.Generator().manual_seed(int(seed))

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    sd = {}
    chans = [1, 16, 32, 64, 64, 64, 64]
    for i in range(1, 7):
        cin, cout = chans[i - 1], chans[i]
        sd['cnn.model.conv%d.weight' % i] = rn(cout, cin, 3, 3, std=(2.0 / (cin * 9)) ** 0.5)
        sd['cnn.model.conv%d.bias' % i] = rn(cout, std=0.1)
        gamma = 1.0 + rn(cout, std=0.2)
        gamma[::7] = -gamma[::7]
        sd['cnn.model.bn%d.weight' % i] = gamma
        sd['cnn.model.bn%d.bias' % i] = rn(cout, std=0.2)
        sd['cnn.model.bn%d.running_mean' % i] = rn(cout, std=0.3)
        sd['cnn.model.bn%d.running_var' % i] = 0.5 + torch.rand(cout, generator=g)
        sd['cnn.model.bn%d.num_batches_tracked' % i] = torch.tensor(100, dtype=torch.int64)
    # conv1 sees dB values of magnitude ~40: keep its output O(1)
    sd['cnn.model.conv1.weight'] = sd['cnn.model.conv1.weight'] * 0.05
    td = 'time_dependency.model.'
    sd[td + 'norm1.weight'] = 1.0 + rn(64, std=0.1)
    sd[td + 'norm1.bias'] = rn(64, std=0.1)
    sd[td + 'linear.weight'] = rn(64, 384, std=384 ** -0.5)
    sd[td + 'linear.bias'] = rn(64, std=0.1)
    for l in range(2):
        p = td + 'layers.%d.' % l
        sd[p + 'self_attn.in_proj_weight'] = rn(192, 64, std=0.25)
        sd[p + 'self_attn.in_proj_bias'] = rn(192, std=0.1)
        sd[p + 'self_attn.out_proj.weight'] = rn(64, 64, std=0.125)
        sd[p + 'self_attn.out_proj.bias'] = rn(64, std=0.1)
        sd[p + 'linear1.weight'] = rn(64, 64, std=0.125)
        sd[p + 'linear1.bias'] = rn(64, std=0.1)
        sd[p + 'linear2.weight'] = rn(64, 64, std=0.125)
        sd[p + 'linear2.bias'] = rn(64, std=0.1)
        for n in ('norm1', 'norm2'):
            sd[p + n + '.weight'] = 1.0 + rn(64, std=0.1)
            sd[p + n + '.bias'] = rn(64, std=0.1)
    heads = ['pool_layers.%d.model.' % h for h in range(5)] if model == 'NISQA_DIM' else ['pool.model.']
    for p in heads:
        sd[p + 'linear1.weight'] = rn(128, 64, std=0.125)
        sd[p + 'linear1.bias'] = rn(128, std=0.1)
        sd[p + 'linear2.weight'] = rn(1, 128, std=0.3)
        sd[p + 'linear2.bias'] = rn(1, std=0.1)
        sd[p + 'linear3.weight'] = rn(1, 64, std=0.125)
        sd[p + 'linear3.bias'] = 3.0 + rn(1, std=0.1)
    return sd


def load_checkpoint(path):
    ck = torch.load(path, map_location='cpu')
    return ck['args'], ck['model_state_dict']


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

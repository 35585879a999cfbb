"""Shared test helpers: checkpoint lookup, seeded random state dicts, fixture loading."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

from nisqa_amd.synth import DIM_ARGS, MOS_ARGS, TTS_ARGS, random_state_dict  # noqa: E402,F401


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


def load_checkpoint(path):
    ck = torch.load(path, map_location='cpu')
    return ck['args'], ck['model_state_dict']


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

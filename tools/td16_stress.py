"""Determinism stress of the self-attention / pooling chain: the same batch N times, outputs compared bit for bit with the first run
(a race in the LDS staging shows up as a run that differs).  python tools/td16_stress.py [runs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth
from nisqa_amd.engine import HipNisqa, BatchPlan

dev = torch.device('cuda:0')
eng = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), dev)
eng32 = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), dev, precision='f32')
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(3)
for name, nw in [('64 x 247', [247] * 64), ('mixed', list(rng.integers(1, 400, 96))), ('short', list(rng.integers(1, 40, 300))), ('long', [1300, 900, 33, 64, 65])]:
    plan = BatchPlan.from_n_wins(np.asarray(nw, np.int64))
    idx = torch.from_numpy(plan.token_index()).to(dev)
    feat = torch.zeros((plan.total_tok, 384), device=dev)
    feat[idx] = torch.randn((len(idx), 384), device=dev)
    ref_x = eng.td(feat, plan).clone()
    ref_o = eng.td_pool(feat, plan).clone()
    bad_x = bad_o = 0
    for i in range(runs):
        x = eng.td(feat, plan)
        o = eng.td_pool(feat, plan)
        bad_x += int(not torch.equal(x[idx], ref_x[idx]))
        bad_o += int(not torch.equal(o, ref_o))
    o2 = eng.pool(ref_x, plan)
    x32 = eng32.td(feat, plan)
    o32 = eng32.pool(x32, plan)
    print('   vs the exact-fp32 kernels: td max|d| %.3g, two-call pooling %.3g, fused %.3g' % (float((x32[idx] - ref_x[idx]).abs().max()), float((o32 - o2).abs().max()), float((o32 - ref_o).abs().max())))
    print('%-10s tokens %6d: td differs in %d / %d runs, td_pool in %d / %d; fused vs two-call pooling max|d| %.3g' % (
        name, plan.total_tok, bad_x, runs, bad_o, runs, float((o2 - ref_o).abs().max())))

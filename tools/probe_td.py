"""How the self-attention / pooling stage scales with the clip length (key tiles per clip) at a fixed number of 32-token tiles:
what part of a td_layer launch is the attention loop, what part the per-tile chain that does not depend on the clip."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nisqa_amd import synth
from nisqa_amd.engine import HipNisqa, BatchPlan
dev = torch.device('cuda:0')
eng = HipNisqa(dict(synth.DIM_ARGS), synth.random_state_dict(7, 'NISQA_DIM'), dev)
for n_wins, clips in ((247, 64), (120, 128), (56, 256), (24, 512)):       # ~512 tiles each
    plan = BatchPlan.from_n_wins([n_wins] * clips)
    feat = torch.randn((plan.total_tok, 384), device=dev)
    for _ in range(5):
        x = eng.td(feat, plan); o = eng.pool(x, plan)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_td = t_pool = 0.0
    for _ in range(20):
        e[0].record(); x = eng.td(feat, plan); e[1].record(); o = eng.pool(x, plan); e[2].record()
        torch.cuda.synchronize()
        t_td += e[0].elapsed_time(e[1]); t_pool += e[1].elapsed_time(e[2])
    print('n_wins %4d x %3d clips (%d tiles, %d key tiles per clip): td %.1f us, pool %.1f us' % (n_wins, clips, plan.total_tok // 32, -(-n_wins // 32), t_td * 50, t_pool * 50))

"""Golden-vector generator -- runs ONLY in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What the fixtures are pinned against (also recorded inside each .npz as `provenance`):

* mel [48,T]  : oracle.mel, a RESTATEMENT of librosa 0.8.1 -> "parity unpinned" at this stage
                (librosa is not installed here and not vendored in the reference).
* network     : the REFERENCE'S OWN torch modules (NISQA_lib.py NISQA_DIM / NISQA, segment_specs),
                imported through oracle.ref_shim (empty `librosa` stand-in module; deviation stated
                there), run on CPU fp32 in padded batches exactly as predict_dim does (NL:1441-1467).

Inputs are regenerated from seeds at test time (nisqa_amd.synth); a CRC of each PCM clip is stored.
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from nisqa_amd import synth                       # noqa: E402
from oracle import mel as omel, ref_shim          # noqa: E402
import helpers                                    # noqa: E402

CLIPS = [('seed', 0, 1.0), ('seed', 1, 3.0), ('seed', 2, 10.0), ('seed', 3, 2.37),
         ('edge', 'zeros', 0), ('edge', 'sine', 0), ('edge', 'min', 0), ('edge', 'max', 0)]
STAGE_CLIPS = [0, 3, 6]           # indices into CLIPS whose intermediate tensors are stored


def clip_pcm(c):
    return synth.synth_pcm16(c[1], c[2]) if c[0] == 'seed' else synth.edge_clip(c[1])


def run_reference(args, sd, specs, stage_idx=None):
    """Reference forward on a padded batch (segment_specs + model), returns outputs and stages."""
    model, NL = ref_shim.build_reference_model(args, sd)
    xs, nw = [], []
    for s in specs:
        x, n = NL.segment_specs('golden', s, args['ms_seg_length'], args['ms_seg_hop_length'],
                                args['ms_max_segments'])
        xs.append(x); nw.append(int(n))
    outs = np.zeros((len(specs), 5 if args['model'] == 'NISQA_DIM' else 1), np.float32)
    stages = {}
    with torch.no_grad():
        # batch 1: everything but the 52 s clip, mixed lengths -> exercises pack/pad + key masks
        small = [i for i in range(len(specs)) if nw[i] < 1000]
        big = [i for i in range(len(specs)) if nw[i] >= 1000]
        for grp in (small, big):
            if not grp:
                continue
            xb = torch.stack([xs[i] for i in grp], 0)
            nb = torch.tensor([nw[i] for i in grp])
            outs[grp] = model(xb, nb).numpy()
        for i in (STAGE_CLIPS if stage_idx is None else stage_idx):
            seg = xs[i][:nw[i]]
            feat = model.cnn.model(seg)
            td, _ = model.time_dependency(feat.unsqueeze(0), torch.tensor([nw[i]]))
            stages['feat_%d' % i] = feat.numpy()
            stages['td_%d' % i] = td[0].numpy()
    return outs, np.array(nw), stages


def main():
    assert ref_shim.reference_available(), 'needs /root/reference'
    pcm = [clip_pcm(c) for c in CLIPS]
    specs = [omel.melspec_db_from_audio(p.astype(np.float32) / np.float32(32768.0), 48000) for p in pcm]
    common = {
        'clip_kind': np.array([c[0] for c in CLIPS]),
        'clip_id': np.array([str(c[1]) for c in CLIPS]),
        'clip_seconds': np.array([float(c[2]) for c in CLIPS]),
        'pcm_crc32': np.array([zlib.crc32(p.tobytes()) for p in pcm], dtype=np.uint64),
        'n_frames': np.array([s.shape[1] for s in specs]),
        'stage_clips': np.array(STAGE_CLIPS),
    }
    mel_fix = dict(common)
    mel_fix['provenance'] = np.array('mel: oracle.mel RESTATEMENT of librosa 0.8.1 (parity unpinned)')
    for i in STAGE_CLIPS + [1]:
        mel_fix['mel_%d' % i] = specs[i]
    np.savez_compressed(os.path.join(HERE, 'mel_oracle.npz'), **mel_fix)

    sets = []
    real = helpers.find_weights('nisqa.tar')
    real_mos = helpers.find_weights('nisqa_mos_only.tar')
    a, sd = helpers.load_checkpoint(real); sets.append(('dim_real', a, sd))
    a, sd = helpers.load_checkpoint(real_mos); sets.append(('mos_real', a, sd))
    sets.append(('dim_rand', dict(helpers.DIM_ARGS), helpers.random_state_dict(7, 'NISQA_DIM')))
    sets.append(('mos_rand', dict(helpers.MOS_ARGS), helpers.random_state_dict(8, 'NISQA')))
    # nisqa_tts.tar architecture (StandardCNN + BiLSTM + last-step pooling): fmax 8000, segment hop 1
    tts_ids = [0, 3, 4, 5, 6, 1]
    tts_specs_all = {i: omel.melspec_db_from_audio(pcm[i].astype(np.float32) / np.float32(32768.0), 48000, fmax=8000.0)
                     for i in tts_ids}
    a, sd = helpers.load_checkpoint(helpers.find_weights('nisqa_tts.tar'))
    tts_sets = [('tts_real', a, sd), ('tts_rand', dict(helpers.TTS_ARGS), helpers.random_state_dict(9, 'NISQA_TTS'))]
    for name, args, sd in tts_sets:
        outs, nw, stages = run_reference(args, sd, [tts_specs_all[i] for i in tts_ids], stage_idx=[0, 1, 4])
        fix = dict(common)
        fix['provenance'] = np.array('network: reference torch modules (StandardCNN, LSTM, PoolLastStepBi) via '
                                     'oracle.ref_shim, CPU fp32, padded batch; input mel: oracle.mel restatement, '
                                     'fmax 8000; weights: ' + name)
        fix['clip_index'] = np.array(tts_ids)
        fix['out'] = outs
        fix['n_wins'] = nw
        fix.update(stages)
        np.savez_compressed(os.path.join(HERE, 'net_%s.npz' % name), **fix)
        print(name, '\n', outs)
    for name, args, sd in sets:
        outs, nw, stages = run_reference(args, sd, specs)
        fix = dict(common)
        fix['provenance'] = np.array(
            'network: reference torch modules via oracle.ref_shim (librosa stand-in), CPU fp32, '
            'padded batch; input mel: oracle.mel restatement; weights: ' + name)
        fix['out'] = outs
        fix['n_wins'] = nw
        fix.update(stages)
        np.savez_compressed(os.path.join(HERE, 'net_%s.npz' % name), **fix)
        print(name, '\n', outs)


if __name__ == '__main__':
    main()

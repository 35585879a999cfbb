"""The mel-stage pin -- runs ONLY on a machine that has the reference's environment (env.yml: librosa==0.8.1,
libsndfile / soundfile); neither is installable in the build container, so the file this writes is NOT in the repo yet
and the mel stage stays "parity unpinned" (tests/test_oracle.py reports XFAIL until it exists).

    conda env create -f /path/to/NISQA/env.yml && conda activate nisqa
    python tests/golden/make_golden_librosa.py          # writes tests/golden/mel_librosa.npz; commit it

For each of the eight parity clips (same seeds as make_golden.py; PCM16 written to a WAV file and read back through
lb.load exactly as NISQA_lib.py:2299-2304 does) it stores what NISQA_lib.get_librosa_melspec (NL:2284-2331) computes:
lb.feature.melspectrogram(S=None, n_fft=4096, hop=0.01 sr, win=0.02 sr, window='hann', center=True,
pad_mode='reflect', power=1.0, n_mels=48, fmin=0, fmax=20000 | 8000, htk=False, norm='slaney') -> lb.core.amplitude_to_db
(ref=1.0, amin=1e-4, top_db=80).  The 52 s clip is stored at fmax 20000 only.
"""
import os
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CLIPS = [('seed', 0, 1.0), ('seed', 1, 3.0), ('seed', 2, 10.0), ('seed', 3, 2.37),
         ('edge', 'zeros', 0), ('edge', 'sine', 0), ('edge', 'min', 0), ('edge', 'max', 0)]


def main():
    import librosa as lb
    if lb.__version__ != '0.8.1':
        raise SystemExit('need librosa==0.8.1 (reference env.yml:16), found ' + lb.__version__)
    from nisqa_amd import synth
    fix = {'librosa_version': np.array(lb.__version__), 'numpy_version': np.array(np.__version__)}
    crc = []
    with tempfile.TemporaryDirectory() as d:
        for i, c in enumerate(CLIPS):
            pcm = synth.synth_pcm16(c[1], c[2]) if c[0] == 'seed' else synth.edge_clip(c[1])
            crc.append(zlib.crc32(pcm.tobytes()))
            path = os.path.join(d, 'c%d.wav' % i)
            synth.write_wav(path, pcm, 48000)
            y, sr = lb.load(path, sr=None)                                            # NL:2304
            assert sr == 48000 and len(y) == len(pcm)
            for tag, fmax in (('mel', 20000.0), ('mel8k', 8000.0)):
                if tag == 'mel8k' and c[1] == 'max':
                    continue
                S = lb.feature.melspectrogram(y=y, sr=sr, S=None, n_fft=4096, hop_length=int(sr * 0.01),
                                              win_length=int(sr * 0.02), window='hann', center=True,
                                              pad_mode='reflect', power=1.0, n_mels=48, fmin=0.0, fmax=fmax,
                                              htk=False, norm='slaney')               # NL:2311-2328
                fix['%s_%d' % (tag, i)] = lb.core.amplitude_to_db(S, ref=1.0, amin=1e-4, top_db=80.0).astype(np.float32)
        # round 5 -- lb.load(path, sr=ms_sr): librosa's default resampler (kaiser_best -> resampy) on a 16 kHz and a 44.1 kHz clip
        # (nisqa_resample / oracle.mel.resample_kaiser_best restate it by recollection), and a FLAC file written by soundfile
        # itself with its bytes (the decoder of libnisqa_ingest.so has only seen streams of tests/flac_enc.py so far)
        import resampy
        fix['resampy_version'] = np.array(resampy.__version__)
        for tag, sr_in, seed in (('rs16', 16000, 40), ('rs44', 44100, 41)):
            pcm = synth.synth_pcm16(seed, 1.0, sr=sr_in)
            path = os.path.join(d, tag + '.wav')
            synth.write_wav(path, pcm, sr_in)
            y, sr = lb.load(path, sr=48000)                                          # NL:2304 with ms_sr = 48000
            assert sr == 48000
            fix[tag + '_pcm_crc32'] = np.array(zlib.crc32(pcm.tobytes()), dtype=np.uint64)
            fix[tag + '_48k'] = y.astype(np.float32)
        try:
            import soundfile as sf
            for tag, data in (('flac_mono16', synth.synth_pcm16(42, 1.0)),
                              ('flac_stereo16', np.stack([synth.synth_pcm16(43, 0.7), synth.synth_pcm16(44, 0.7)], 1))):
                path = os.path.join(d, tag + '.flac')
                sf.write(path, data, 48000, subtype='PCM_16')
                y, sr = lb.load(path, sr=None)
                fix[tag + '_bytes'] = np.frombuffer(open(path, 'rb').read(), dtype=np.uint8)
                fix[tag + '_y'] = y.astype(np.float32)
        except Exception as e:                                                       # noqa: BLE001
            print('no FLAC pin written:', e)
    fix['pcm_crc32'] = np.array(crc, dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, 'mel_librosa.npz'), **fix)
    print('wrote', os.path.join(HERE, 'mel_librosa.npz'))


if __name__ == '__main__':
    main()

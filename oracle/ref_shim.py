"""Oracle support (test infrastructure): import the reference's torch modules in
a container that has no librosa.

DEVIATION, stated wherever it is used: ``nisqa/NISQA_lib.py:10`` does
``import librosa as lb`` at module scope and librosa is not installed here, so
an EMPTY stand-in module is registered as ``sys.modules['librosa']`` before
the import.  Only torch code of the reference then runs (model classes NL:29-1417,
``segment_specs`` NL:2239-2282); ``lb.load`` / ``lb.feature.melspectrogram`` /
``lb.core.amplitude_to_db`` are never reached.  /root/reference exists only in
the build container: this module is used by ``tests/golden/make_golden.py``
and by CPU tests that skip when the reference is absent, never on the GPU box.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('NISQA_REFERENCE_ROOT', '/root/reference')


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'nisqa', 'NISQA_lib.py'))


def import_reference_lib():
    """Return the reference's ``nisqa.NISQA_lib`` module (torch parts usable)."""
    if not reference_available():
        raise ImportError('reference tree not present at ' + REFERENCE_ROOT)
    sys.dont_write_bytecode = True            # keep /root/reference clean
    standin = 'librosa' not in sys.modules
    if standin:
        sys.modules['librosa'] = types.ModuleType('librosa')   # the stand-in
    import matplotlib
    matplotlib.use('Agg')
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    try:
        return importlib.import_module('nisqa.NISQA_lib')
    finally:
        if standin:                            # the reference keeps its own `lb`; nobody else should see the stand-in
            del sys.modules['librosa']


MODEL_ARG_KEYS = [
    'ms_seg_length', 'ms_n_mels', 'cnn_model', 'cnn_c_out_1', 'cnn_c_out_2', 'cnn_c_out_3',
    'cnn_kernel_size', 'cnn_dropout', 'cnn_pool_1', 'cnn_pool_2', 'cnn_pool_3', 'cnn_fc_out_h',
    'td', 'td_sa_d_model', 'td_sa_nhead', 'td_sa_pos_enc', 'td_sa_num_layers', 'td_sa_h',
    'td_sa_dropout', 'td_lstm_h', 'td_lstm_num_layers', 'td_lstm_dropout', 'td_lstm_bidirectional',
    'td_2', 'td_2_sa_d_model', 'td_2_sa_nhead', 'td_2_sa_pos_enc', 'td_2_sa_num_layers', 'td_2_sa_h',
    'td_2_sa_dropout', 'td_2_lstm_h', 'td_2_lstm_num_layers', 'td_2_lstm_dropout',
    'td_2_lstm_bidirectional', 'pool', 'pool_att_h', 'pool_att_dropout']


def build_reference_model(args, state_dict):
    """The reference's own module tree with ``state_dict`` loaded strict (NISQA_model.py:1012-1023)."""
    NL = import_reference_lib()
    margs = {k: args[k] for k in MODEL_ARG_KEYS}
    model = {'NISQA': NL.NISQA, 'NISQA_DIM': NL.NISQA_DIM}[args['model']](**margs)
    model.load_state_dict(state_dict, strict=True)
    model.eval()
    return model, NL

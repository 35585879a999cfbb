"""ctypes binding of libnisqa_hip.so (the C ABI of include/nisqa_hip.h).

The product path has no fallback: if the shared library is missing this module raises at load
time (build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C nisqa_amd/csrc``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NISQA_HIP_LIB') or os.path.join(_HERE, 'libnisqa_hip.so')     # override: A/B of two builds

NISQA_OK, NISQA_ERR_ARG, NISQA_ERR_LAUNCH, NISQA_ERR_WORKSPACE = 0, 1, 2, 3
ABI_VERSION = 2

c_p = ctypes.c_void_p
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64


class MelCfg(ctypes.Structure):
    """nisqa_mel_cfg"""
    _fields_ = [('n_fft', c_i32), ('hop', c_i32), ('win', c_i32), ('n_mels', c_i32), ('n_bins', c_i32),
                ('w_floats', c_i32), ('amin_sq', ctypes.c_float), ('top_db', ctypes.c_float)]


class ModelDev(ctypes.Structure):
    """nisqa_model_dev"""
    _fields_ = [('window', c_p), ('twiddle', c_p), ('band_start', c_p), ('band_len', c_p), ('band_woff', c_p),
                ('band_w', c_p), ('cnn_w', c_p), ('td_w', c_p), ('pool_w', c_p),
                ('n_layers', c_i32), ('n_heads', c_i32), ('seg_hop', c_i32), ('stage_events', c_p),
                ('cnn_wb', c_p), ('cnn_mode', c_i32), ('td_wb', c_p), ('pool_wb', c_p), ('arch', c_i32)]


# name -> (restype, argtypes); every symbol include/nisqa_hip.h declares
SYMBOLS = {
    'nisqa_abi_version': (ctypes.c_int, []),
    'nisqa_mel_db': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, ctypes.POINTER(MelCfg), c_p, c_p, c_p, c_p, c_p, c_p,
                                    c_p, c_p, c_p]),
    'nisqa_mel_db_pcm16': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, ctypes.POINTER(MelCfg), c_p, c_p, c_p, c_p, c_p, c_p,
                                          c_p, c_p, c_p]),
    'nisqa_mel_finalize': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_p, ctypes.c_float, c_p, c_i32, c_p]),
    'nisqa_cnn_adapt': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_cnn_front': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'nisqa_cnn_adapt_bf16': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_cnn_adapt_segments_bf16': (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_cnn_adapt_bf16x6': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_cnn_adapt_segments_bf16x6': (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_cnn_adapt_f16': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_p]),
    'nisqa_cnn_adapt_segments_f16': (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_p]),
    'nisqa_cnn_adapt_segments': (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_cnn_back': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p]),
    'nisqa_cnn_standard': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_cnn_standard_bf16': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_cnn_standard_bf16x6': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_cnn_standard_f16': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_p]),
    'nisqa_lstm_laststep': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_td_selfatt': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_td_selfatt_bf16': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_pool_att_bf16': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_td_selfatt_bf16x6': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_pool_att_bf16x6': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_td_pool_bf16x6': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_pool_score_bf16': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_pool_score_bf16x6': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_pool_final': (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'nisqa_pool_att': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_workspace_bytes': (ctypes.c_size_t, [c_i32, c_i32, c_i32]),
    'nisqa_predict_batch': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, ctypes.POINTER(MelCfg),
                                           ctypes.POINTER(ModelDev), c_p, ctypes.c_size_t, c_p, c_p]),
    'nisqa_predict_batch_pcm16': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, ctypes.POINTER(MelCfg),
                                                 ctypes.POINTER(ModelDev), c_p, ctypes.c_size_t, c_p, c_p]),
    'nisqa_pcm16_to_f32': (ctypes.c_int, [c_p, c_p, c_i64, c_p]),
    'nisqa_resample_workspace_bytes': (ctypes.c_size_t, [c_i32, c_i64]),
    'nisqa_resample': (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_p, c_i32, c_i64, ctypes.c_double, c_p, c_i32, c_i32, c_p, ctypes.c_size_t, c_p, c_p]),
    'nisqa_selftest_mfma': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_p]),
    'nisqa_probe_mfma_sustained': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_p]),
}

class TdTrainArgs(ctypes.Structure):
    """nisqa_tdtrain_args (include/nisqa_train.h)"""
    _fields_ = [('n_clips', c_i32), ('n_tokens', c_i32), ('n_tokens_padded', c_i32), ('n_layers', c_i32), ('n_heads', c_i32),
                ('n_wgrad_groups', c_i32), ('n_wgrad_tiles', c_i32), ('n_colsum_jobs', c_i32),
                ('seg_off', c_p), ('ptok_off', c_p), ('tile_clip', c_p), ('sq_off', c_p), ('params', c_p), ('grads', c_p),
                ('poff', c_p), ('ws', c_p), ('frags', c_p), ('labels', c_p), ('bias_map', c_p), ('inv_count', c_p),
                ('mask_p', c_p * 4), ('mask_1', c_p * 4), ('mask_f', c_p * 4), ('mask_2', c_p * 4),
                ('wgrad_desc', c_p), ('colsum_jobs', c_p)]


# include/nisqa_train.h: operators of the training step (same shared library)
c_f = ctypes.c_float
TRAIN_SYMBOLS = {
    'nisqa_tdtrain_plan': (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_i64]),
    'nisqa_tdtrain_step': (ctypes.c_int, [ctypes.POINTER(TdTrainArgs), c_p]),
    'nisqa_gemm_f32': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_f, c_p]),
    'nisqa_gemm_f32_one': (ctypes.c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32,
                                          c_f, c_p, c_i32, c_p]),
    'nisqa_conv1_fwd': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_conv1_wgrad': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'nisqa_conv1_moments': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p]),
    'nisqa_conv1_bn_act_pool_fwd': (ctypes.c_int, [c_p] * 4 + [c_i32] * 3 + [c_p] * 13),
    'nisqa_conv1_bn_act_pool_bwd': (ctypes.c_int, [c_p] * 4 + [c_i32] * 3 + [c_p] * 14),
    'nisqa_im2col_mel': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p]),
    'nisqa_im2col3x3': (ctypes.c_int, [c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p]),
    'nisqa_conv3x3_gemm': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p]),
    'nisqa_conv3x3_fwd_stats': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'nisqa_conv3x3_gemm_bf16': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_p]),
    'nisqa_segconv_supported': (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    'nisqa_segconv_frag_bytes': (ctypes.c_int64, [c_i32, c_i32, c_i32]),
    'nisqa_segconv_pack': (ctypes.c_int, [c_i32, c_p, c_i32, c_i32, c_p, c_p]),
    'nisqa_segconv_pack_many': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_segconv_bf16': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'nisqa_segconv_frag_bytes_f32': (ctypes.c_int64, [c_i32, c_i32, c_i32]),
    'nisqa_segconv_pack_f32_many': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_segconv_f32': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'nisqa_segconv_frag_bytes_x6': (ctypes.c_int64, [c_i32, c_i32, c_i32]),
    'nisqa_segconv_frag_bytes_f16': (ctypes.c_int64, [c_i32, c_i32, c_i32]),
    'nisqa_segconv_pack_f16_many': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_segconv_f16': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'nisqa_segconv_pack_x6_many': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_p, c_p, c_p]),
    'nisqa_segconv_bf16x6': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]),
    'nisqa_segconv_wgrad_bf16': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]),
    'nisqa_col2im3x3': (ctypes.c_int, [c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p]),
    'nisqa_col_dot': (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_p, c_p]),
    'nisqa_bn_act_pool_fwd': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                             c_p, c_p, c_p, c_p]),
    'nisqa_bn_act_pool_bwd1': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                              c_p, c_p, c_p]),
    'nisqa_bn_act_pool_bwd': (ctypes.c_int, [c_p] * 7 + [c_i32] * 6 + [c_p] * 5),
    'nisqa_bn_pool_bwd_sums': (ctypes.c_int, [c_p] * 7 + [c_i32] * 6 + [c_p] * 2),
    'nisqa_segconv_wgrad_bn_bf16': (ctypes.c_int, [c_p] * 13 + [c_i32] * 8 + [c_p]),
    'nisqa_segconv_wgrad_f32': (ctypes.c_int, [c_p] * 13 + [c_i32] * 8 + [c_p]),
    'nisqa_segconv_wgrad_bf16x6': (ctypes.c_int, [c_p] * 13 + [c_i32] * 8 + [c_p]),
    'nisqa_bn_bwd2': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i32, c_p, c_p, c_p, c_p]),
    'nisqa_layernorm_fwd': (ctypes.c_int, [c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p]),
    'nisqa_layernorm_bwd': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i64, c_p, c_p]),
    'nisqa_softmax_rows_fwd': (ctypes.c_int, [c_p, c_p, c_p, c_i64, c_f, c_p, c_p]),
    'nisqa_softmax_rows_bwd': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i64, c_f, c_p, c_p]),
    'nisqa_elementwise': (ctypes.c_int, [c_i32, c_p, c_p, c_p, c_i64, c_i32, c_p, c_p]),
    'nisqa_mse_loss': (ctypes.c_int, [c_p, c_p, c_p, c_i32, c_i32, c_p, c_p, c_p]),
    'nisqa_dropout_mask': (ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint64, c_f, c_i64, c_p, c_p]),
    'nisqa_cast_scatter': (ctypes.c_int, [c_p, c_p, c_i32, c_p, c_p]),
    'nisqa_adam_step': (ctypes.c_int, [c_p, c_p, c_p, c_p, c_i64, c_f, c_i32, c_p]),
}

_lib = None


def exported_symbols(path, prefix=''):
    """Names in the dynamic symbol table (.dynsym) of an ELF64 shared object that are DEFINED there and start with `prefix`."""
    import struct
    with open(path, 'rb') as f:
        d = f.read()
    if d[:4] != b'\x7fELF' or d[4] != 2 or d[5] != 1:
        raise RuntimeError('nisqa_amd: %s is not a little-endian ELF64 file' % path)
    shoff, = struct.unpack_from('<Q', d, 0x28)
    shentsize, shnum = struct.unpack_from('<HH', d, 0x3A)
    sec = [struct.unpack_from('<IIQQQQIIQQ', d, shoff + i * shentsize) for i in range(shnum)]
    out = []
    for (_, typ, _, _, off, size, link, _, _, entsize) in sec:
        if typ != 11 or not entsize:                 # SHT_DYNSYM
            continue
        stroff = sec[link][4]
        for k in range(size // entsize):
            st_name, _, _, st_shndx = struct.unpack_from('<IBBH', d, off + k * entsize)
            if st_shndx == 0:                        # undefined: an import
                continue
            end = d.index(b'\0', stroff + st_name)
            name = d[stroff + st_name:end].decode('ascii', 'replace')
            if name.startswith(prefix):
                out.append(name)
    return sorted(out)


def refuse_debug_library(path):
    """A library that exports nisqa_debug_* readers was built with -DNQ_EXPERIMENTAL (csrc/experimental.hpp: in-kernel phase clocks and
    whatever else an experiment compiled in): it times differently from the product and is never what a caller should score audio
    with.  Refused unless NISQA_ALLOW_DEBUG_LIB=1 (the tools/ that read the clocks set it)."""
    dbg = exported_symbols(path, 'nisqa_debug_')
    if dbg and os.environ.get('NISQA_ALLOW_DEBUG_LIB') != '1':
        raise RuntimeError('nisqa_amd: %s exports %s -- an instrumented (-DNQ_EXPERIMENTAL) build, not the product library; '
                           'set NISQA_ALLOW_DEBUG_LIB=1 to load it for a measurement' % (path, ', '.join(dbg)))
    return dbg


def load():
    """Load (once) and return the ctypes library with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be
    # mapped BEFORE libnisqa_hip.so so that our DT_NEEDED libamdhip64.so.7 binds to that same runtime; loaded
    # the other way round the process ends up with two HIP runtimes and torch's streams/pointers are foreign.
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            'nisqa_amd: HIP library not built: %s is missing (run __graft_entry__.build() or '
            '`make -C nisqa_amd/csrc`); there is no CPU fallback.' % LIB_PATH)
    refuse_debug_library(LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in list(SYMBOLS.items()) + list(TRAIN_SYMBOLS.items()):
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.nisqa_abi_version() != ABI_VERSION:
        raise RuntimeError('nisqa_amd: libnisqa_hip.so ABI %d != expected %d' % (lib.nisqa_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


# ---- libnisqa_ingest.so (include/nisqa_ingest.h): native WAV ingest, plain C++ -- loadable without a GPU ----------
INGEST_PATH = os.path.join(_HERE, 'libnisqa_ingest.so')
INGEST_ABI_VERSION = 2
WAV_OK, WAV_ERR_OPEN, WAV_ERR_FORMAT, WAV_ERR_READ = 0, 1, 2, 3
WAV_TAG_PCM, WAV_TAG_FLAC = 1, 0xF1AC


class WavInfo(ctypes.Structure):
    """nisqa_wav_info"""
    _fields_ = [('status', c_i32), ('tag', c_i32), ('channels', c_i32), ('bits', c_i32), ('block_align', c_i32),
                ('sample_rate', c_i32), ('data_offset', c_i64), ('n_frames', c_i64)]


INGEST_SYMBOLS = {
    'nisqa_ingest_abi_version': (ctypes.c_int, []),
    'nisqa_ingest_probe': (ctypes.c_int, [ctypes.POINTER(ctypes.c_char_p), c_i32, ctypes.POINTER(WavInfo), c_i32]),
    'nisqa_ingest_read': (ctypes.c_int, [ctypes.POINTER(ctypes.c_char_p), c_i32, ctypes.POINTER(WavInfo), c_p,
                                         ctypes.POINTER(c_i64), c_i32]),
    'nisqa_ingest_decode_flac': (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(WavInfo), c_p]),
}

_ingest = None


def load_ingest():
    """Load (once) and return libnisqa_ingest.so with typed entry points; raises if it has not been built."""
    global _ingest
    if _ingest is not None:
        return _ingest
    if not os.path.isfile(INGEST_PATH):
        raise RuntimeError('nisqa_amd: native ingest library not built: %s is missing (run __graft_entry__.build() '
                           'or `make -C nisqa_amd/csrc`)' % INGEST_PATH)
    lib = ctypes.CDLL(INGEST_PATH)
    for name, (res, args) in INGEST_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.nisqa_ingest_abi_version() != INGEST_ABI_VERSION:
        raise RuntimeError('nisqa_amd: libnisqa_ingest.so ABI %d != expected %d'
                           % (lib.nisqa_ingest_abi_version(), INGEST_ABI_VERSION))
    _ingest = lib
    return lib


class NisqaHipError(RuntimeError):
    pass


def check(rc, what):
    if rc != NISQA_OK:
        names = {1: 'NISQA_ERR_ARG', 2: 'NISQA_ERR_LAUNCH', 3: 'NISQA_ERR_WORKSPACE'}
        raise NisqaHipError('%s failed: %s' % (what, names.get(rc, rc)))

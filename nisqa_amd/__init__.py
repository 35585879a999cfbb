"""nisqa_amd -- MI355X (gfx950) native engine for the NISQA predict hot path.

Drop-in surface: ``nisqa_amd.NISQA_model.nisqaModel(args).predict()`` and ``run_predict.py``
(same flags, same result frame/CSV as gabrielmittag/NISQA); the mel front end, framewise CNN,
self-attention and attention pooling run as hand-written HIP kernels behind the C ABI declared in
``include/nisqa_hip.h`` (``libnisqa_hip.so``).  There is no CPU or PyTorch-eager fallback: the
engine raises if the HIP library or a GPU is missing.
"""
__version__ = '0.1.0'

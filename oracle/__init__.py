"""CPU oracle for the NISQA predict hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``nisqa_amd``) never
imports this package and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):

* network half (segmenting, AdaptCNN, self-attention, attention pooling):
  PINNED -- ``tests/golden/make_golden.py`` runs the reference's own torch
  modules (imported from /root/reference through a librosa stand-in, because
  librosa is not installed here) and the fixtures it wrote are checked
  against ``oracle.net`` in ``tests/test_oracle.py``.
* mel half (WAV decode, STFT, mel filterbank, dB): **parity unpinned** --
  the arithmetic lives in librosa 0.8.1 (env.yml:16), which is neither
  vendored under /root/reference nor installable here; ``oracle.mel`` is a
  restatement of that library's published algorithm.
"""

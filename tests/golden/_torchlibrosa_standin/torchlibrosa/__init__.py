"""TEST-ONLY stand-in for the third-party package ``torchlibrosa==0.0.4``.

The reference (`/root/reference/pytorch/models.py:10-11`) imports
``torchlibrosa.stft.{Spectrogram,LogmelFilterBank}`` and
``torchlibrosa.augmentation.SpecAugmentation``; that package (and ``librosa``,
which it calls for the Hann window and the mel matrix) is neither vendored in the
reference nor installed in this image.  This stand-in restates the published
0.0.4 semantics so the *genuine*
reference model code can be imported by ``tests/golden/make_golden.py`` to
produce golden vectors.  It is never imported by the product or by GPU tests.

Parity status: F1/F2/F4 are "parity unpinned" by any reference test; they are
cross-checked against ``torch.stft`` and ``transformers.audio_utils.mel_filter_bank``
in ``tests/test_oracle_frontend.py``.
"""
__version__ = "0.0.4-standin"

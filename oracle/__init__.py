"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

Plain-PyTorch (CPU, fp32) / numpy restatement of the reference hot path
(log-mel front-end -> bn0 -> SpecAugment -> mixup -> 4x ConvBlock -> head ->
clip_bce -> Adam-amsgrad), each function citing the reference file:line it
follows.  It exists to CHECK the HIP path; it is never the thing shipped or
measured.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg
of `bench.py` may import this package.  The product package
(`sound_event_detection_dcase2017_task4_amd`) must never import it and fails
loudly when its HIP library is missing.

Pinning status
--------------
* Trunk / heads / loss / mixup / Adam (SURVEY.md §8a rows F3, F5, C1-C5, H1-H3,
  L1, O1): PINNED against golden vectors produced in the build container by the
  reference's own `pytorch/models.py`, `losses.py`, `pytorch_utils.py` imported
  unmodified (`tests/golden/make_golden.py`; fixtures in `tests/golden/*.npz`;
  checked by `tests/test_oracle_*.py`).
* Transformer heads (`MultiHead`, `models.py:587-665`): PINNED the same way; the two `nn.Dropout`s of `MultiHead` are
  replaced in the generator by a module applying SEEDED keep masks (`oracle.model.dropout_masks`), which the oracle and
  the product take as explicit inputs, so training-mode parity is checkable (the reference draws them from torch's RNG).
* Front-end (rows F1, F2, F4 = third-party `torchlibrosa==0.0.4` + `librosa`,
  absent from /root/reference and from this image): **parity unpinned** by any
  reference test.  Restated from the published 0.0.4 algorithm, anchored on the
  reference's call sites (`pytorch/models.py:251-262`, `:284-292`) and
  cross-checked against two independent implementations (`torch.stft`,
  `transformers.audio_utils.mel_filter_bank`) in `tests/test_oracle_frontend.py`.
* `Mixup.get_lambda` (`utils/utilities.py:220-242`; module not importable here
  because it imports librosa/sed_eval): restated; pinned by numpy's frozen
  legacy `RandomState(1234).beta(1, 1)` stream.
"""

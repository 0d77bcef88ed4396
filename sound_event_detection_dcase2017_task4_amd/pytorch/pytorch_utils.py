"""Tensor utilities on the hot path; same names and behaviour as reference pytorch/pytorch_utils.py."""
import logging
from collections import defaultdict

import numpy as np
import torch

from .. import ops


def move_data_to_device(x, device):
    """numpy array -> tensor on `device`: float* becomes float32, int* becomes int64, anything else (e.g. the array
    of audio names) is handed back untouched (reference pytorch_utils.py:6-15)."""
    kind = np.dtype(x.dtype).kind if not torch.is_tensor(x) else None
    if kind == 'f':
        return torch.as_tensor(np.asarray(x, dtype=np.float32)).to(device)
    if kind in 'iu':
        return torch.as_tensor(np.asarray(x, dtype=np.int64)).to(device)
    return x


def append_to_dict(dict, key, value):
    """dict[key] is a list that grows (reference pytorch_utils.py:18-22)."""
    dict.setdefault(key, []).append(value)


def _to_host(v):
    # loader tensors may live in recycled (pinned / device) buffers: always copy out
    return v.cpu().numpy().copy() if torch.is_tensor(v) else v


def forward(model, data_loader, return_input=False, return_target=False):
    """Eval-mode inference over a loader (reference pytorch_utils.py:25-77).  Returns numpy arrays concatenated over the
    batches: 'audio_name', 'clipwise_output', 'framewise_output' (when the model has one), plus 'waveform' /
    'target' / 'strong_target' on request.  Batches may carry numpy float waveforms (reference loaders) or int16 /
    float tensors (PinnedBatchLoader)."""
    device = next(model.parameters()).device
    model.eval()
    collected = defaultdict(list)
    wanted_from_batch = (['waveform'] if return_input else []) + (['target', 'strong_target'] if return_target else [])
    for batch in data_loader:
        wave = batch['waveform']
        wave_dev = wave.to(device) if torch.is_tensor(wave) else move_data_to_device(wave, device)
        with torch.no_grad():
            out = model(wave_dev)
        host = {key: out[key].detach().cpu().numpy() for key in ('clipwise_output', 'framewise_output') if key in out}
        try:
            ops.check_device_errors(nonfinite=True)      # the copies above synchronised: the flag is current
        except ops.NonFiniteOperand as err:
            if err.skipped_steps:
                # not an inference problem: optimiser steps BEFORE this loop were refused and nobody had polled yet (the train
                # CLI drains with optimizer.poll(0) before it evaluates) -- the training loop must hear about it
                raise
            # a non-finite operand in a split-f16 convolution: redo this batch on the fp32 MFMA kernels, which propagate
            # NaN / inf exactly like the reference's torch ops
            logging.warning('non-finite activations in batch %d of the inference loop: re-running it on the fp32 kernels',
                            len(collected['audio_name']))
            prev, ops.USE_SF16 = ops.USE_SF16, False
            try:
                with torch.no_grad():
                    out = model(wave_dev)
                host = {key: out[key].detach().cpu().numpy() for key in ('clipwise_output', 'framewise_output') if key in out}
            finally:
                ops.USE_SF16 = prev
        collected['audio_name'].append(batch['audio_name'])
        for key, val in host.items():
            collected[key].append(val)
        for key in wanted_from_batch:
            if key in batch:
                collected[key].append(_to_host(batch[key]))
    return {key: np.concatenate(parts, axis=0) for key, parts in collected.items()}


def do_mixup(x, mixup_lambda):
    """out[i] = lam[2i] * x[2i] + lam[2i+1] * x[2i+1], (N, ...) -> (N/2, ...) (reference pytorch_utils.py:80-93)."""
    return ops.mixup_rows(x, mixup_lambda)

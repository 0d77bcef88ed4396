"""Tensor utilities on the hot path (reference pytorch/pytorch_utils.py)."""
import numpy as np
import torch

from .. import ops


def move_data_to_device(x, device):
    """pytorch_utils.py:6-15: float arrays -> float32 tensor, int arrays -> int64 tensor, others unchanged."""
    if 'float' in str(x.dtype):
        x = torch.Tensor(x)
    elif 'int' in str(x.dtype):
        x = torch.LongTensor(x)
    else:
        return x
    return x.to(device)


def append_to_dict(dict, key, value):
    if key in dict.keys():
        dict[key].append(value)
    else:
        dict[key] = [value]


def forward(model, data_loader, return_input=False, return_target=False):
    """Batched eval-mode inference (pytorch_utils.py:25-77) -> dict of numpy arrays."""
    device = next(model.parameters()).device
    output_dict = {}
    def host(v):                                         # loader tensors may live in recycled buffers: copy out
        return v.cpu().numpy().copy() if torch.is_tensor(v) else v

    for n, batch_data_dict in enumerate(data_loader):
        w = batch_data_dict['waveform']                  # numpy float (reference loaders) or int16 / float tensor
        batch_waveform = w.to(device) if torch.is_tensor(w) else move_data_to_device(w, device)
        with torch.no_grad():
            model.eval()
            batch_output = model(batch_waveform)
        append_to_dict(output_dict, 'audio_name', batch_data_dict['audio_name'])
        append_to_dict(output_dict, 'clipwise_output', batch_output['clipwise_output'].data.cpu().numpy())
        if 'framewise_output' in batch_output.keys():
            append_to_dict(output_dict, 'framewise_output', batch_output['framewise_output'].data.cpu().numpy())
        if return_input:
            append_to_dict(output_dict, 'waveform', host(batch_data_dict['waveform']))
        if return_target:
            if 'target' in batch_data_dict.keys():
                append_to_dict(output_dict, 'target', host(batch_data_dict['target']))
            if 'strong_target' in batch_data_dict.keys():
                append_to_dict(output_dict, 'strong_target', host(batch_data_dict['strong_target']))
    for key in output_dict.keys():
        output_dict[key] = np.concatenate(output_dict[key], axis=0)
    return output_dict


def do_mixup(x, mixup_lambda):
    """pytorch_utils.py:80-93: out[i] = lam[2i]*x[2i] + lam[2i+1]*x[2i+1]   (N, ...) -> (N/2, ...)."""
    return ops.mixup_rows(x, mixup_lambda)

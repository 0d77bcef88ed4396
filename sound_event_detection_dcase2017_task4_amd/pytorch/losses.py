"""Loss of the hot path (reference pytorch/losses.py:5-17), computed by the HIP kernel sed_clip_bce."""
from .. import ops


def clip_bce(output_dict, target_dict):
    """Binary cross entropy (mean, log terms clamped at -100) of clipwise_output vs target."""
    return ops.ClipBceFn.apply(output_dict['clipwise_output'], target_dict['target'])


def get_loss_func(loss_type):
    if loss_type == 'clip_bce':
        return clip_bce

"""Host side of SpecAugmentation (torchlibrosa 0.0.4 `SpecAugmentation(64, 2, 8, 2)` as constructed at reference
pytorch/models.py:261-262 and called at :291-292): draws the stripe positions from the GLOBAL torch CPU RNG in
exactly the package's order, so a seeded run drops the same stripes as the reference.  The zeroing itself happens
inside the fused bn0 kernel (csrc/bn.hip).

Package order: for every sample n (batch order), `stripes_num` x { distance = randint(0, drop_width),
bgn = randint(0, total - distance) } on the time axis; then a second pass over the batch for the mel axis.  Each
draw consumes one 32-bit Mersenne-Twister output (`r % range`), so the whole stream can be replayed in a handful of
batched `torch.randint` calls (one per distinct modulus) instead of 8 Python-level calls per sample.
"""
import numpy as np
import torch


def draw_specaug_stripes(batch, frames, mel_bins=64, time_drop_width=64, freq_drop_width=8):
    """-> int32 (batch, 8): [t_bgn0, t_len0, t_bgn1, t_len1, f_bgn0, f_len0, f_bgn1, f_len1].
    Advances the global CPU generator by exactly 8*batch draws, like the package."""
    n = batch * 8
    state = torch.get_rng_state()

    def stream(rng):
        torch.set_rng_state(state)
        return torch.randint(0, int(rng), (n,)).numpy()

    pos = np.arange(batch)[:, None] * 4 + np.arange(2)[None, :] * 2       # index of each `distance` draw
    pos_f = 4 * batch + pos
    d_t = stream(time_drop_width)[pos]
    d_f = stream(freq_drop_width)[pos_f]
    b_t = np.zeros_like(d_t)
    b_f = np.zeros_like(d_f)
    for d in np.unique(d_t):
        s = stream(frames - int(d))
        m = d_t == d
        b_t[m] = s[pos[m] + 1]
    for d in np.unique(d_f):
        s = stream(mel_bins - int(d))
        m = d_f == d
        b_f[m] = s[pos_f[m] + 1]
    stream(2)                                                                # leave the generator n draws ahead
    out = np.empty((batch, 8), dtype=np.int32)
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = b_t[:, 0], d_t[:, 0], b_t[:, 1], d_t[:, 1]
    out[:, 4], out[:, 5], out[:, 6], out[:, 7] = b_f[:, 0], d_f[:, 0], b_f[:, 1], d_f[:, 1]
    return out

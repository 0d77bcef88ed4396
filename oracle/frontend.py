"""ORACLE (test infrastructure only) — log-mel front-end, CPU fp32.

Restates `torchlibrosa==0.0.4` `Spectrogram` + `LogmelFilterBank` +
`SpecAugmentation` exactly as the reference constructs and calls them
(`/root/reference/pytorch/models.py:251-262` ctor, `:284-292` forward).
torchlibrosa/librosa are third-party and absent -> "parity unpinned", see
`oracle/__init__.py`.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# utils/config.py:1-13
SAMPLE_RATE = 32000
WINDOW_SIZE = 1024
HOP_SIZE = 320
MEL_BINS = 64
FMIN = 50
FMAX = 14000
N_BINS = WINDOW_SIZE // 2 + 1
AMIN = 1e-10


def hann_window(n=WINDOW_SIZE):
    """Periodic Hann (`fftbins=True`), float64.  models.py:246 `window='hann'`."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def _slaney_hz_to_mel(hz):
    hz = np.asarray(hz, dtype=np.float64)
    lin = hz * 3.0 / 200.0
    log_region = 15.0 + np.log(np.maximum(hz, 1e-300) / 1000.0) * (27.0 / np.log(6.4))
    return np.where(hz >= 1000.0, log_region, lin)


def _slaney_mel_to_hz(mel):
    mel = np.asarray(mel, dtype=np.float64)
    lin = mel * 200.0 / 3.0
    log_region = 1000.0 * np.exp((mel - 15.0) * (np.log(6.4) / 27.0))
    return np.where(mel >= 15.0, log_region, lin)


def mel_matrix(sr=SAMPLE_RATE, n_fft=WINDOW_SIZE, n_mels=MEL_BINS, fmin=FMIN, fmax=FMAX):
    """`librosa.filters.mel(...).T` -> (513, 64) float32: Slaney mel scale, triangular,
    area-normalised (`2/(f[m+2]-f[m])`).  Call site: models.py:256-258."""
    n_bins = n_fft // 2 + 1
    bin_hz = np.linspace(0.0, sr / 2.0, n_bins)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax), n_mels + 2))
    W = np.zeros((n_mels, n_bins), dtype=np.float32)
    for m in range(n_mels):
        left, centre, right = edges[m], edges[m + 1], edges[m + 2]
        rising = (bin_hz - left) / (centre - left)
        falling = (right - bin_hz) / (right - centre)
        W[m] = np.clip(np.minimum(rising, falling), 0.0, None)      # stored in float32, like librosa
    W *= (2.0 / (edges[2:] - edges[:-2]))[:, None]                  # in-place float32 *= float64
    return np.ascontiguousarray(W.T)


def dft_weights(n_fft=WINDOW_SIZE):
    """Frozen `conv_real/conv_imag` weights (513,1,1024) f32 = Re/Im(exp(-2*pi*i*n*k/N) * hann[n])
    (torchlibrosa STFT ctor; state_dict keys `spectrogram_extractor.stft.conv_{real,imag}.weight`)."""
    n = np.arange(n_fft)
    k = np.arange(n_fft // 2 + 1)
    omega = np.exp(-2j * np.pi / n_fft)
    Wc = np.power(omega, np.outer(k, n)) * hann_window(n_fft)[None, :]
    return (np.real(Wc).astype(np.float32)[:, None, :], np.imag(Wc).astype(np.float32)[:, None, :])


_CACHE = {}


def _consts():
    if not _CACHE:
        wr, wi = dft_weights()
        _CACHE["wr"] = torch.from_numpy(wr)
        _CACHE["wi"] = torch.from_numpy(wi)
        _CACHE["mel"] = torch.from_numpy(mel_matrix())
    return _CACHE


def power_spectrogram(x):
    """F1.  x (B2, L) f32 -> (B2, 1, T, 513): reflect-pad 512, two strided conv1d, re^2+im^2.
    models.py:284."""
    c = _consts()
    z = F.pad(x[:, None, :], (WINDOW_SIZE // 2, WINDOW_SIZE // 2), mode="reflect")
    re = F.conv1d(z, c["wr"], stride=HOP_SIZE)
    im = F.conv1d(z, c["wi"], stride=HOP_SIZE)
    return (re * re + im * im).transpose(1, 2)[:, None, :, :]


def logmel(x):
    """F1+F2.  (B2, L) -> (B2, 1, T, 64): matmul with melW, 10*log10(clamp(., 1e-10)) - 0.
    models.py:284-285."""
    c = _consts()
    mel = torch.matmul(power_spectrogram(x), c["mel"])
    return 10.0 * torch.log10(torch.clamp(mel, min=AMIN))


def draw_specaug_stripes(batch, frames, mel_bins=MEL_BINS, time_width=64, time_num=2, freq_width=8, freq_num=2):
    """F4 draw order (models.py:261-262, :291-292): for every sample, `time_num` x {distance =
    randint(0, width), bgn = randint(0, total - distance)} on the time axis; THEN a second pass over
    the batch for the mel axis.  Uses the global torch CPU RNG exactly like the package does.
    Returns int32 (batch, 2*(time_num+freq_num)) rows = [tb0, td0, tb1, td1, fb0, fd0, fb1, fd1]."""
    out = np.zeros((batch, 2 * (time_num + freq_num)), dtype=np.int32)
    for n in range(batch):
        for s in range(time_num):
            d = int(torch.randint(low=0, high=time_width, size=(1,))[0])
            b = int(torch.randint(low=0, high=frames - d, size=(1,))[0])
            out[n, 2 * s], out[n, 2 * s + 1] = b, d
    for n in range(batch):
        for s in range(freq_num):
            d = int(torch.randint(low=0, high=freq_width, size=(1,))[0])
            b = int(torch.randint(low=0, high=mel_bins - d, size=(1,))[0])
            out[n, 2 * time_num + 2 * s], out[n, 2 * time_num + 2 * s + 1] = b, d
    return out


def apply_specaug(x, stripes, time_num=2, freq_num=2):
    """Zero the stripes in place-equivalent fashion.  x (B2,1,T,M)."""
    mask = torch.ones_like(x)
    for n in range(x.shape[0]):
        for s in range(time_num):
            b, d = int(stripes[n, 2 * s]), int(stripes[n, 2 * s + 1])
            mask[n, :, b:b + d, :] = 0
        for s in range(freq_num):
            b, d = int(stripes[n, 2 * time_num + 2 * s]), int(stripes[n, 2 * time_num + 2 * s + 1])
            mask[n, :, :, b:b + d] = 0
    return x * mask


def mixup_lambdas(batch_size, random_state):
    """F6.  utils/utilities.py:220-242: per pair lam = RandomState.beta(1,1,1)[0]; emit [lam, 1-lam]."""
    out = []
    for _ in range(0, batch_size, 2):
        lam = random_state.beta(1.0, 1.0, 1)[0]
        out.append(lam)
        out.append(1.0 - lam)
    return np.array(out)

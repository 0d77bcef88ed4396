"""Stand-in for torchlibrosa 0.0.4 ``stft`` module (test-only, see package docstring)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _hann_periodic(n):
    # librosa.filters.get_window('hann', n, fftbins=True) == scipy periodic Hann
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz,
                    min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def librosa_mel(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney') -> (n_mels, 1+n_fft//2) f32."""
    n_bins = 1 + n_fft // 2
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


class DFTBase(nn.Module):
    def dft_matrix(self, n):
        (x, y) = np.meshgrid(np.arange(n), np.arange(n))
        omega = np.exp(-2 * np.pi * 1j / n)
        return np.power(omega, x * y)


class STFT(DFTBase):
    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window='hann',
                 center=True, pad_mode='reflect', freeze_parameters=True):
        super().__init__()
        assert pad_mode in ['constant', 'reflect'] and window == 'hann'
        self.n_fft, self.center, self.pad_mode = n_fft, center, pad_mode
        win_length = n_fft if win_length is None else win_length
        hop_length = int(win_length // 4) if hop_length is None else hop_length
        assert win_length == n_fft
        fft_window = _hann_periodic(win_length)
        self.W = self.dft_matrix(n_fft)
        out_channels = n_fft // 2 + 1
        self.conv_real = nn.Conv1d(1, out_channels, n_fft, stride=hop_length, padding=0, bias=False)
        self.conv_imag = nn.Conv1d(1, out_channels, n_fft, stride=hop_length, padding=0, bias=False)
        self.conv_real.weight.data = torch.Tensor(
            np.real(self.W[:, 0:out_channels] * fft_window[:, None]).T)[:, None, :]
        self.conv_imag.weight.data = torch.Tensor(
            np.imag(self.W[:, 0:out_channels] * fft_window[:, None]).T)[:, None, :]
        if freeze_parameters:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, input):
        x = input[:, None, :]
        if self.center:
            x = F.pad(x, pad=(self.n_fft // 2, self.n_fft // 2), mode=self.pad_mode)
        real = self.conv_real(x)
        imag = self.conv_imag(x)
        real = real[:, None, :, :].transpose(2, 3)
        imag = imag[:, None, :, :].transpose(2, 3)
        return real, imag


class Spectrogram(nn.Module):
    def __init__(self, n_fft=2048, hop_length=None, win_length=None, window='hann',
                 center=True, pad_mode='reflect', power=2.0, freeze_parameters=True):
        super().__init__()
        self.power = power
        self.stft = STFT(n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window,
                         center=center, pad_mode=pad_mode, freeze_parameters=True)

    def forward(self, input):
        (real, imag) = self.stft.forward(input)
        spectrogram = real ** 2 + imag ** 2
        if self.power == 2.0:
            pass
        else:
            spectrogram = spectrogram ** (self.power / 2.0)
        return spectrogram


class LogmelFilterBank(nn.Module):
    def __init__(self, sr=32000, n_fft=2048, n_mels=64, fmin=50, fmax=14000, is_log=True,
                 ref=1.0, amin=1e-10, top_db=80.0, freeze_parameters=True):
        super().__init__()
        self.is_log, self.ref, self.amin, self.top_db = is_log, ref, amin, top_db
        self.melW = nn.Parameter(torch.Tensor(librosa_mel(sr, n_fft, n_mels, fmin, fmax).T))
        if freeze_parameters:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, input):
        mel_spectrogram = torch.matmul(input, self.melW)
        return self.power_to_db(mel_spectrogram) if self.is_log else mel_spectrogram

    def power_to_db(self, input):
        log_spec = 10.0 * torch.log10(torch.clamp(input, min=self.amin, max=np.inf))
        log_spec -= 10.0 * np.log10(np.maximum(self.amin, self.ref))
        if self.top_db is not None:
            if self.top_db < 0:
                raise ValueError('top_db must be non-negative')
            log_spec = torch.clamp(log_spec, min=log_spec.max().item() - self.top_db, max=np.inf)
        return log_spec

"""CPU: pin the oracle's front-end (F1/F2/F4/F6) against the committed golden vectors and
against two independent implementations (torch.stft, transformers mel_filter_bank)."""
import os

import numpy as np
import pytest
import torch

from oracle import frontend as ofe


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "frontend.npz"))


def waves(seed, n, length):
    return (np.random.RandomState(seed).randn(n, length) * 0.1).astype(np.float32)


def test_mel_matrix_matches_golden_and_survey_constants(fx):
    W = ofe.mel_matrix()
    assert W.shape == (513, 64) and W.dtype == np.float32
    np.testing.assert_allclose(W, fx["melW"], rtol=0, atol=2e-9)
    # SURVEY.md Appendix A.2
    assert abs(W.max() - 0.018149516) < 1e-8 and abs(W.sum() - 2.048126) < 1e-5
    assert int((W != 0).sum()) == 866
    assert abs(W[3, 0] - 0.01501181) < 1e-8 and abs(W[424, 63] - 0.00134121) < 1e-8


def test_mel_matrix_vs_transformers():
    tr = pytest.importorskip("transformers.audio_utils")
    ref = tr.mel_filter_bank(513, 64, 50, 14000, 32000, norm="slaney", mel_scale="slaney")
    np.testing.assert_allclose(ofe.mel_matrix(), ref.astype(np.float32), rtol=0, atol=5e-9)


def test_dft_weights_match_golden(fx):
    wr, wi = ofe.dft_weights()
    rows = [0, 1, 7, 256, 512]
    np.testing.assert_allclose(wr[rows, 0], fx["conv_real_rows"], atol=1e-7)
    np.testing.assert_allclose(wi[rows, 0], fx["conv_imag_rows"], atol=1e-7)


def test_logmel_golden_1s_and_10s(fx):
    y = ofe.logmel(torch.from_numpy(waves(1234, 1, 32000)))[0, 0].numpy()
    assert y.shape == (101, 64)
    np.testing.assert_allclose(y, fx["logmel_1s"], atol=2e-5)
    np.testing.assert_allclose(y[0, :3], [-14.963374, -17.270578, -16.579090], atol=1e-4)
    y10 = ofe.logmel(torch.from_numpy(waves(4321, 1, 320000)))[0, 0].numpy()
    assert y10.shape == (1001, 64)
    np.testing.assert_allclose(y10, fx["logmel_10s"], atol=2e-5)


def test_tone_known_answer(fx):
    n = np.arange(32000)
    tone = (0.5 * np.cos(2 * np.pi * 1000 * n / 32000)).astype(np.float32)
    p = ofe.power_spectrogram(torch.from_numpy(tone[None]))[0, 0, 50].numpy()
    assert abs(p[32] - 16384.0) < 0.05 and abs(p[31] - 4096.0) < 0.05 and abs(p[33] - 4096.0) < 0.05
    np.testing.assert_allclose(p, fx["tone_power_frame50"], rtol=1e-4, atol=1e-6)
    lm = ofe.logmel(torch.from_numpy(tone[None]))[0, 0].numpy()
    np.testing.assert_allclose(lm, fx["tone_logmel"], atol=2e-3)


def test_silence_is_minus_100_db(fx):
    y = ofe.logmel(torch.zeros(1, 3200))[0, 0].numpy()
    assert np.all(y == -100.0) and np.all(fx["silence_logmel"] == -100.0)


def test_int16_path(fx):
    x = (fx["int16_wave"] / 32767.0).astype(np.float32)
    y = ofe.logmel(torch.from_numpy(x))[0, 0].numpy()
    np.testing.assert_allclose(y, fx["int16_logmel"], atol=2e-5)


def test_power_spectrogram_vs_torch_stft():
    x = torch.from_numpy(waves(9, 2, 16000))
    ours = ofe.power_spectrogram(x)[:, 0]
    st = torch.stft(x, n_fft=1024, hop_length=320, window=torch.hann_window(1024, periodic=True),
                    center=True, pad_mode="reflect", return_complex=True)
    ref = (st.real ** 2 + st.imag ** 2).transpose(1, 2)
    assert ours.shape == ref.shape == (2, 51, 513)
    assert ((ours - ref).abs().max() / ref.abs().max()).item() < 5e-6


def test_frame_count_and_reflect_padding():
    x = torch.arange(2048, dtype=torch.float32)[None] / 2048
    assert ofe.power_spectrogram(x).shape == (1, 1, 1 + 2048 // 320, 513)
    # frame 0 must see x[512], ..., x[1], x[0], ..., x[511]
    z = torch.nn.functional.pad(x[:, None], (512, 512), mode="reflect")[0, 0]
    assert z[0] == x[0, 512] and z[511] == x[0, 1] and z[512] == x[0, 0]


def test_mixup_lambda_stream(golden_dir):
    misc = np.load(os.path.join(golden_dir, "misc.npz"))
    lam = ofe.mixup_lambdas(64, np.random.RandomState(1234))
    np.testing.assert_array_equal(lam, misc["mixup_lambda64"])
    assert lam[0] == 0.23538938957272115 and abs(lam[1] - (1 - lam[0])) == 0
    assert lam[2] == 0.4166419486264242


def test_specaug_draw_order_and_ranges(golden_dir):
    misc = np.load(os.path.join(golden_dir, "misc.npz"))
    torch.manual_seed(7)
    s = ofe.draw_specaug_stripes(4, 1001, 64)
    np.testing.assert_array_equal(s, misc["specaug_seed7_B4_T1001"])
    assert (s[:, [1, 3]] < 64).all() and (s[:, [5, 7]] < 8).all()
    assert (s[:, [0, 2]] + s[:, [1, 3]] <= 1001).all() and (s[:, [4, 6]] + s[:, [5, 7]] <= 64).all()
    x = torch.ones(4, 1, 1001, 64)
    y = ofe.apply_specaug(x, s)
    assert y[0, 0, s[0, 0]:s[0, 0] + s[0, 1]].sum() == 0
    assert y[1, 0, :, s[1, 4]:s[1, 4] + s[1, 5]].sum() == 0


def test_power_spectrogram_vs_numpy_fft_in_float64():
    """A third, independent evaluation of F1 (torchlibrosa Spectrogram as called at models.py:251-253): plain numpy in
    float64 -- np.pad(mode='reflect') by n_fft/2, frames at hop 320, periodic Hann, np.fft.rfft, |.|^2 -- against the oracle's
    conv1d formulation on noise (the reference's own layout: power (B, 1, T, 513))."""
    L = 32000
    x = waves(77, 2, L)
    got = ofe.power_spectrogram(torch.from_numpy(x)).numpy()
    n_fft, hop = 1024, 320
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)
    T = L // hop + 1
    assert got.shape == (2, 1, T, 513)
    for b in range(2):
        xp = np.pad(x[b].astype(np.float64), n_fft // 2, mode="reflect")
        frames = np.stack([xp[t * hop:t * hop + n_fft] * win for t in range(T)])
        want = np.abs(np.fft.rfft(frames, axis=1)) ** 2
        err = np.abs(got[b, 0] - want).max() / want.max()
        assert err < 2e-6, err

"""Model classes of the hot path, drop-in for reference pytorch/models.py: same class names, positional
constructor `(sample_rate, window_size, hop_size, mel_bins, fmin, fmax, classes_num)`, `forward(input,
mixup_lambda=None)` returning {'framewise_output', 'clipwise_output', 'embedding'}, and identical
`state_dict()` keys / shapes (checkpoint compatible), but every op runs as a hand-written HIP kernel
(libsed_hip.so) on NHWC activations.  nn.Module sub-objects (Conv2d, BatchNorm2d, Linear, GRU ...) are used purely
as PARAMETER CONTAINERS so that keys, shapes, `.to()`, `.train()/.eval()` behave exactly like the reference;
their own forward methods are never called.

Covered: Cnn_9layers_FrameMax (:152), Cnn_9layers_FrameAvg (:237), Cnn_9layers_FrameAtt (:322),
Cnn_9layers_Gru_FrameAvg (:403), Cnn_9layers_Gru_FrameAtt (:495), Cnn_9layers_Transformer_FrameAvg (:668),
Cnn_9layers_Transformer_FrameAtt (:762) -- every model type of the reference.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..utils.augmentation import draw_specaug_stripes

__all__ = ['Cnn_9layers_FrameMax', 'Cnn_9layers_FrameAvg', 'Cnn_9layers_FrameAtt', 'Cnn_9layers_Gru_FrameAvg',
           'Cnn_9layers_Gru_FrameAtt', 'Cnn_9layers_Transformer_FrameAvg', 'Cnn_9layers_Transformer_FrameAtt', 'ConvBlock',
           'AttBlock', 'MultiHead', 'init_layer', 'init_bn', 'init_gru', 'interpolate']


def init_layer(layer):
    """models.py:15-21."""
    nn.init.xavier_uniform_(layer.weight)
    if hasattr(layer, 'bias'):
        if layer.bias is not None:
            layer.bias.data.fill_(0.)


def init_bn(bn):
    """models.py:24-27."""
    bn.bias.data.fill_(0.)
    bn.weight.data.fill_(1.)


def init_gru(rnn):
    """models.py:30-55: per-gate uniform(+-sqrt(3/fan_in)) for W_ih and the r,z blocks of W_hh, orthogonal n block, zero
    biases -- for `weight_*_l{i}` / `bias_*_l{i}` ONLY.  The reference loop never names the `_reverse` parameters, so the
    backward direction of its bidirectional GRU keeps nn.GRU's default U(+-1/sqrt(hidden)) weights AND biases; same here."""
    def _concat_init(tensor, init_funcs):
        (length, fan_out) = tensor.shape
        fan_in = length // len(init_funcs)
        for (i, init_func) in enumerate(init_funcs):
            init_func(tensor[i * fan_in: (i + 1) * fan_in, :])

    def _inner_uniform(tensor):
        fan_in = nn.init._calculate_correct_fan(tensor, 'fan_in')
        nn.init.uniform_(tensor, -math.sqrt(3 / fan_in), math.sqrt(3 / fan_in))

    for i in range(rnn.num_layers):
        _concat_init(getattr(rnn, 'weight_ih_l{}'.format(i)), [_inner_uniform] * 3)
        torch.nn.init.constant_(getattr(rnn, 'bias_ih_l{}'.format(i)), 0)
        _concat_init(getattr(rnn, 'weight_hh_l{}'.format(i)), [_inner_uniform, _inner_uniform, nn.init.orthogonal_])
        torch.nn.init.constant_(getattr(rnn, 'bias_hh_l{}'.format(i)), 0)


def interpolate(x, ratio):
    """models.py:58-69: (B, T, C) -> (B, T*ratio, C), each frame repeated `ratio` times."""
    return ops.interpolate(x.contiguous(), ratio)


# ---- frozen front-end parameter containers (state_dict keys of torchlibrosa 0.0.4) ---------------------------

class _STFT(nn.Module):
    def __init__(self, n_fft, hop_length):
        super(_STFT, self).__init__()
        out_channels = n_fft // 2 + 1
        self.conv_real = nn.Conv1d(1, out_channels, n_fft, stride=hop_length, padding=0, bias=False)
        self.conv_imag = nn.Conv1d(1, out_channels, n_fft, stride=hop_length, padding=0, bias=False)
        n = np.arange(n_fft)
        window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)                     # periodic Hann
        W = np.exp(-2j * np.pi * np.outer(np.arange(out_channels), n) / n_fft) * window[None, :]
        self.conv_real.weight.data = torch.Tensor(np.real(W))[:, None, :]
        self.conv_imag.weight.data = torch.Tensor(np.imag(W))[:, None, :]
        for p in self.parameters():
            p.requires_grad = False


class Spectrogram(nn.Module):
    """Holds `stft.conv_real/conv_imag.weight` (513,1,1024).  The kernel uses row 0 of conv_real (= the window)
    and an FFT instead of the dense DFT."""

    def __init__(self, n_fft, hop_length):
        super(Spectrogram, self).__init__()
        self.stft = _STFT(n_fft, hop_length)


def _slaney_mel(sr, n_fft, n_mels, fmin, fmax):
    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * 27.0 / np.log(6.4), f * 3.0 / 200.0)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * np.log(6.4) / 27.0), m * 200.0 / 3.0)

    bins = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    edges = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    W = np.zeros((n_mels, len(bins)), dtype=np.float32)
    for m in range(n_mels):
        up = (bins - edges[m]) / (edges[m + 1] - edges[m])
        down = (edges[m + 2] - bins) / (edges[m + 2] - edges[m + 1])
        W[m] = np.maximum(0.0, np.minimum(up, down))
    W *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return W


class LogmelFilterBank(nn.Module):
    def __init__(self, sr, n_fft, n_mels, fmin, fmax, amin):
        super(LogmelFilterBank, self).__init__()
        self.amin = amin
        self.melW = nn.Parameter(torch.Tensor(_slaney_mel(sr, n_fft, n_mels, fmin, fmax).T), requires_grad=False)


class SpecAugmentation(nn.Module):
    """Parameter-free; the stripes are drawn on the host (same global-RNG stream as torchlibrosa 0.0.4) and applied
    inside the fused bn0 kernel."""

    def __init__(self, time_drop_width, time_stripes_num, freq_drop_width, freq_stripes_num):
        super(SpecAugmentation, self).__init__()
        assert (time_stripes_num, freq_stripes_num) == (2, 2)
        self.time_drop_width, self.freq_drop_width = time_drop_width, freq_drop_width

    def draw(self, batch, frames, mel_bins):
        return draw_specaug_stripes(batch, frames, mel_bins, self.time_drop_width, self.freq_drop_width)


# ---- building blocks ---------------------------------------------------------------------------------------------

class ConvBlock(nn.Module):
    """models.py:72-115.  NHWC in/out: (B,H,W,Cin) -> (B,H//ph,W//pw,Cout); pool_type 'avg' (what every model selects),
    'max' or 'avg+max' (:104-111), anything else raises like the reference."""
    POOL_MODES = {'avg': 0, 'max': 1, 'avg+max': 2}

    def __init__(self, in_channels, out_channels):
        super(ConvBlock, self).__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.init_weight()

    def init_weight(self):
        init_layer(self.conv1)
        init_layer(self.conv2)
        init_bn(self.bn1)
        init_bn(self.bn2)

    def forward(self, input, pool_size=(2, 2), pool_type='avg', pairs_out=False):
        """pairs_out (extension, used by the models' trunk between blocks): the output may be written as split-f16 OPERAND PAIRS
        (ops.ACT_PAIRS) -- same shape and bytes, float32-typed for autograd's sake, but only meaningful to the next ConvBlock,
        which recognises it by the `_sed_pairs` attribute.  A caller that reads the values must leave it False."""
        if pool_type not in self.POOL_MODES:
            raise Exception('Incorrect argument!')
        # the amax of a block's output (left on the device by its pool kernel) rides on the tensor to the next block, whose
        # split-f16 convolution takes its operand scale from it -- no extra pass, no host synchronisation
        out, out_amax = ops.ConvBlockFn.apply(input, self.conv1.weight, self.bn1.weight, self.bn1.bias,
                                              self.bn1.running_mean, self.bn1.running_var, self.conv2.weight,
                                              self.bn2.weight, self.bn2.bias, self.bn2.running_mean, self.bn2.running_var,
                                              self.training, pool_size[0], pool_size[1], getattr(input, '_sed_amax', None),
                                              self.POOL_MODES[pool_type], not torch.is_grad_enabled(),
                                              bool(getattr(input, '_sed_pairs', False)), bool(pairs_out))
        out._sed_amax = out_amax
        out._sed_pairs = bool(pairs_out) and torch.is_grad_enabled() and ops.block_out_pairs_ok(
            self.training, self.POOL_MODES[pool_type], pool_size[0], pool_size[1], input.shape[1], input.shape[2],
            self.conv1.weight.shape[0])
        if self.training and not getattr(self, '_defer_counters', False):
            self.bn1.num_batches_tracked += 1
            self.bn2.num_batches_tracked += 1
        return out


class AttBlock(nn.Module):
    """models.py:118-149: activation 'linear' (the reference's default) or 'sigmoid' (what every model passes), any
    temperature > 0.  Input (B, T, n_in) [time-major]; returns (clip (B,n_out), norm_att (B,n_out,T), cla (B,n_out,T)) like
    the reference (the last two as transposed views).  `bn_att` exists in the state_dict but is unused, as in the
    reference."""

    def __init__(self, n_in, n_out, activation='linear', temperature=1.):
        super(AttBlock, self).__init__()
        if activation not in ('linear', 'sigmoid') or not temperature > 0:
            raise Exception('Incorrect argument!')
        self.activation = activation
        self.temperature = temperature
        self.att = nn.Conv1d(n_in, n_out, kernel_size=1, stride=1, padding=0, bias=True)
        self.cla = nn.Conv1d(n_in, n_out, kernel_size=1, stride=1, padding=0, bias=True)
        self.bn_att = nn.BatchNorm1d(n_out)
        self.init_weights()

    def init_weights(self):
        init_layer(self.att)
        init_layer(self.cla)
        init_bn(self.bn_att)

    def forward(self, x_btc):
        clip, cla, natt = ops.AttHeadFn.apply(x_btc, self.att.weight, self.att.bias, self.cla.weight, self.cla.bias,
                                              self.activation, self.temperature)
        return clip, natt.transpose(1, 2), cla.transpose(1, 2)


class MultiHead(nn.Module):
    """Multi-head self-attention block of the Transformer models (models.py:611-665): parameter container with the
    reference's keys (w_qs, w_ks, w_vs, layer_norm [never used], fc) and inits; forward(x (B,T,d_model), dropout_masks)
    -> (B,T,d_model).  In training mode the two dropouts (attention 0.1, output `dropout`) use the given KEEP masks
    `(attn (n_head*B,T,T), fc (B,T,d_model))` or, when None, masks drawn on the device from torch's generator."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1):
        super(MultiHead, self).__init__()
        if (n_head, d_model, d_k, d_v) != (8, 512, 64, 64):
            # the attention kernels are specialised for the one configuration the reference's models construct (models.py:698, :793)
            raise ValueError('Incorrect argument! MultiHead(n_head, d_model, d_k, d_v) supports (8, 512, 64, 64) only -- the sizes '
                             'of Cnn_9layers_Transformer_* -- got (%r, %r, %r, %r)' % (n_head, d_model, d_k, d_v))
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = nn.Linear(d_model, n_head * d_k)
        self.w_ks = nn.Linear(d_model, n_head * d_k)
        self.w_vs = nn.Linear(d_model, n_head * d_v)
        nn.init.normal_(self.w_qs.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_ks.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_vs.weight, mean=0, std=np.sqrt(2.0 / (d_model + d_v)))
        self.w_qs.bias.data.fill_(0)
        self.w_ks.bias.data.fill_(0)
        self.w_vs.bias.data.fill_(0)
        self.attn_dropout_p = 0.1                      # ScaledDotProductAttention(attn_dropout=0.1), models.py:590
        self.layer_norm = nn.LayerNorm(d_model)        # in the state_dict, never applied (models.py:660-665)
        self.fc = nn.Linear(n_head * d_v, d_model)
        nn.init.xavier_normal_(self.fc.weight)
        self.fc.bias.data.fill_(0)
        self.dropout_p = dropout

    def forward(self, x_btc, dropout_masks=None):
        keep_attn = keep_fc = None
        if self.training:
            B, T, C = x_btc.shape
            if dropout_masks is None:
                # keep masks drawn as bytes in ONE kernel each (a float draw + a compare wrote and re-read 4 bytes per element)
                keep_attn = torch.empty((self.n_head * B, T, T), dtype=torch.uint8, device=x_btc.device).bernoulli_(1.0 - self.attn_dropout_p)
                keep_fc = torch.empty((B, T, C), dtype=torch.uint8, device=x_btc.device).bernoulli_(1.0 - self.dropout_p)
            else:
                keep_attn, keep_fc = (m.to(device=x_btc.device, dtype=torch.bool) for m in dropout_masks)
        return ops.MultiHeadFn.apply(x_btc, self.w_qs.weight, self.w_qs.bias, self.w_ks.weight, self.w_ks.bias,
                                     self.w_vs.weight, self.w_vs.bias, self.fc.weight, self.fc.bias, keep_attn, keep_fc,
                                     self.attn_dropout_p, self.dropout_p)


# ---- models ---------------------------------------------------------------------------------------------------------

class _Cnn9Base(nn.Module):
    interpolate_ratio = 8

    def __init__(self, sample_rate, window_size, hop_size, mel_bins, fmin, fmax, classes_num):
        super(_Cnn9Base, self).__init__()
        if (window_size, hop_size, mel_bins) != (1024, 320, 64):
            # The supported set, precisely: window_size = 1024 (the log-mel kernel is a radix-32 x 32 FFT of two real frames),
            # hop_size = 320, mel_bins = 64 (the reference itself hard-codes BatchNorm2d(64) on the mel axis, models.py:264, so no
            # other value constructs a working model there either); sample_rate, fmin, fmax and classes_num are free (they only
            # shape the mel filter bank and the head).  A ValueError is an Exception: `except Exception` callers of the
            # reference's bare Exception('Incorrect argument!') (models.py:113) keep working.
            raise ValueError('Incorrect argument! This build supports window_size=1024, hop_size=320, mel_bins=64 '
                             '(utils/config.py) with any sample_rate / fmin / fmax / classes_num; got window_size=%r, hop_size=%r, '
                             'mel_bins=%r' % (window_size, hop_size, mel_bins))
        self.classes_num = classes_num
        self.spectrogram_extractor = Spectrogram(n_fft=window_size, hop_length=hop_size)
        self.logmel_extractor = LogmelFilterBank(sr=sample_rate, n_fft=window_size, n_mels=mel_bins, fmin=fmin,
                                                 fmax=fmax, amin=1e-10)
        self.spec_augmenter = SpecAugmentation(time_drop_width=64, time_stripes_num=2, freq_drop_width=8,
                                               freq_stripes_num=2)
        self.bn0 = nn.BatchNorm2d(64)
        self.conv_block1 = ConvBlock(in_channels=1, out_channels=64)
        self.conv_block2 = ConvBlock(in_channels=64, out_channels=128)
        self.conv_block3 = ConvBlock(in_channels=128, out_channels=256)
        self.conv_block4 = ConvBlock(in_channels=256, out_channels=512)
        self._tables = None
        self._tables_key = None
        for blk in (self.conv_block1, self.conv_block2, self.conv_block3, self.conv_block4):
            blk._defer_counters = True          # the trunk bumps all nine num_batches_tracked in ONE launch

    # -- front-end tables: rebuilt when the frozen parameters move device or are reloaded
    def _frontend(self):
        w = self.spectrogram_extractor.stft.conv_real.weight
        mel = self.logmel_extractor.melW
        key = (w.device, w.data_ptr(), w._version, mel.data_ptr(), mel._version)
        if self._tables is None or self._tables_key != key:
            self._tables = ops.frontend_tables(w[0, 0, :], mel, w.device)
            self._tables_key = key
        return self._tables

    def bn_counters(self):
        """`num_batches_tracked` of the nine BatchNorms a training-mode forward pass goes through (att_block.bn_att is never
        applied, models.py:129).  ops.rollback_bn_counters() takes refused steps back out of them."""
        return [self.bn0.num_batches_tracked] + [bn.num_batches_tracked for blk in (
            self.conv_block1, self.conv_block2, self.conv_block3, self.conv_block4) for bn in (blk.bn1, blk.bn2)]

    def extract_logmel(self, input):
        """(B2, L) waveform (float32 or int16) -> (B2, T, 64) log-mel.  models.py:284-285."""
        return ops.logmel(input, self._frontend(), self.logmel_extractor.amin)

    def trunk(self, input, mixup_lambda=None, stripes=None):
        """models.py:284-303 -> features (B, T/8, 512), time-major."""
        if not input.is_cuda:
            raise RuntimeError('the MI355X hot path needs CUDA/HIP tensors (model.to("cuda"), input on the GPU)')
        # the nine BatchNorms PROPOSE their new running statistics; one launch at the end of the pass installs them unless
        # a kernel of the pass met NaN / inf (found-non-finite guard: a refused step leaves the buffers intact)
        ops.begin_bn_commit()
        try:
            feat = self._trunk(input, mixup_lambda, stripes)
        except BaseException:
            ops.commit_bn(drop=True)
            raise
        ops.commit_bn()
        return feat

    def _trunk(self, input, mixup_lambda=None, stripes=None):
        lm = self.extract_logmel(input)
        B2, T, M = lm.shape
        lam = None
        if self.training:
            if stripes is None:
                stripes = self.spec_augmenter.draw(B2, T, M)
            if not (torch.is_tensor(stripes) and stripes.is_cuda):
                stripes = ops.upload_small(stripes, lm.device, torch.int32)      # pinned staging: the host never waits for the GPU
            stripes = stripes.to(device=lm.device, dtype=torch.int32).contiguous()
            if mixup_lambda is not None:
                lam = mixup_lambda.to(device=lm.device, dtype=torch.float32).contiguous()
        else:
            stripes = None
        x = ops.Bn0AugMix.apply(lm, self.bn0.weight, self.bn0.bias, self.bn0.running_mean, self.bn0.running_var,
                                self.training, stripes, lam)
        if self.training:                                                # nn.BatchNorm2d bookkeeping: one foreach launch
            torch._foreach_add_(self.bn_counters(), 1)
        x = x.view(x.shape[0], T, M, 1)                                  # NHWC, C = 1
        if ops.USE_SF16:                 # split-f16 operands of all seven MFMA conv weights: two launches per optimiser step
            blks = (self.conv_block1, self.conv_block2, self.conv_block3, self.conv_block4)
            ops.prepack_sf16([b.conv1.weight for b in blks[1:]] + [b.conv2.weight for b in blks],
                             self.training and torch.is_grad_enabled())
        x = self.conv_block1(x, pool_size=(2, 2), pool_type='avg', pairs_out=True)     # (pairs: only when the next block takes them)
        x = self.conv_block2(x, pool_size=(2, 2), pool_type='avg', pairs_out=True)
        x = self.conv_block3(x, pool_size=(2, 2), pool_type='avg', pairs_out=True)
        # block 4: pool (1,1) followed by torch.mean(dim=3)  ==  one (1, W) average pool
        x = self.conv_block4(x, pool_size=(1, x.shape[2]), pool_type='avg')
        feat = x.view(x.shape[0], x.shape[1], x.shape[3])               # (B, T', 512)
        feat._sed_amax = getattr(x, '_sed_amax', None)                  # block 4's pool left the amax of its output on the device
        return feat


class _FcHead(_Cnn9Base):
    _mode = 0

    def __init__(self, sample_rate, window_size, hop_size, mel_bins, fmin, fmax, classes_num):
        super(_FcHead, self).__init__(sample_rate, window_size, hop_size, mel_bins, fmin, fmax, classes_num)
        self._make_mid()
        self.fc = nn.Linear(512, classes_num, bias=True)
        self.init_weights()

    def _make_mid(self):
        pass

    def _mid(self, feat, dropout_masks=None):
        return feat

    def init_weights(self):
        init_bn(self.bn0)
        init_layer(self.fc)

    def forward(self, input, mixup_lambda=None, specaug_stripes=None, dropout_masks=None):
        """Input: (batch_size, data_length).  `specaug_stripes` (optional, extension) fixes the SpecAugment draws,
        `dropout_masks` (optional, extension, Transformer models) the two MultiHead dropout keep masks."""
        feat = self._mid(self.trunk(input, mixup_lambda, specaug_stripes), dropout_masks)
        frame, clip = ops.FcHeadFn.apply(feat, self.fc.weight, self.fc.bias, self._mode)
        return {'framewise_output': interpolate(frame, self.interpolate_ratio), 'clipwise_output': clip,
                'embedding': feat.transpose(1, 2)}


class Cnn_9layers_FrameAvg(_FcHead):
    """models.py:237-319."""
    _mode = 0


class Cnn_9layers_FrameMax(_FcHead):
    """models.py:152-234."""
    _mode = 1


class _GruMixin(object):
    def _make_mid(self):
        self.gru = nn.GRU(input_size=512, hidden_size=256, num_layers=1, bias=True, batch_first=True, bidirectional=True)

    def _mid(self, feat, dropout_masks=None):
        g = self.gru
        return ops.GruFn.apply(feat, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0, g.weight_ih_l0_reverse,
                               g.weight_hh_l0_reverse, g.bias_ih_l0_reverse, g.bias_hh_l0_reverse,
                               getattr(feat, '_sed_amax', None))


class Cnn_9layers_Gru_FrameAvg(_GruMixin, _FcHead):
    """models.py:403-492."""
    _mode = 0

    def init_weights(self):
        init_bn(self.bn0)
        init_gru(self.gru)
        init_layer(self.fc)


class _AttHead(_Cnn9Base):
    def __init__(self, sample_rate, window_size, hop_size, mel_bins, fmin, fmax, classes_num):
        super(_AttHead, self).__init__(sample_rate, window_size, hop_size, mel_bins, fmin, fmax, classes_num)
        self._make_mid()
        self.att_block = AttBlock(n_in=512, n_out=17, activation='sigmoid')
        self.init_weights()

    _embedding_is_cla = True                          # models.py:398 / :579 return cla; the Transformer variant (:851) x

    def _make_mid(self):
        pass

    def _mid(self, feat, dropout_masks=None):
        return feat

    def init_weights(self):
        init_bn(self.bn0)

    def forward(self, input, mixup_lambda=None, specaug_stripes=None, dropout_masks=None):
        feat = self._mid(self.trunk(input, mixup_lambda, specaug_stripes), dropout_masks)
        (clipwise_output, norm_att, cla) = self.att_block(feat)
        framewise_output = interpolate(cla.transpose(1, 2), self.interpolate_ratio)
        return {'framewise_output': framewise_output, 'clipwise_output': clipwise_output,
                'embedding': cla if self._embedding_is_cla else feat.transpose(1, 2)}


class Cnn_9layers_FrameAtt(_AttHead):
    """models.py:322-400."""


class Cnn_9layers_Gru_FrameAtt(_GruMixin, _AttHead):
    """models.py:495-581."""

    def init_weights(self):
        init_bn(self.bn0)
        init_gru(self.gru)


class _TransformerMixin(object):
    def _make_mid(self):
        self.multihead = MultiHead(n_head=8, d_model=512, d_k=64, d_v=64, dropout=0.2)

    def _mid(self, feat, dropout_masks=None):
        return self.multihead(feat, dropout_masks)


class Cnn_9layers_Transformer_FrameAvg(_TransformerMixin, _FcHead):
    """models.py:668-759."""
    _mode = 0


class Cnn_9layers_Transformer_FrameAtt(_TransformerMixin, _AttHead):
    """models.py:762-853."""
    _embedding_is_cla = False

"""ORACLE (test infrastructure only) — CNN trunk, heads, loss, optimiser, CPU fp32.

Functional restatement of `/root/reference/pytorch/models.py` (ConvBlock :72-115,
AttBlock :118-149, Cnn_9layers_FrameMax :152, _FrameAvg :237-319, _FrameAtt :322-400,
_Gru_FrameAvg :403, _Gru_FrameAtt :495-581, MultiHead :587-665, _Transformer_FrameAvg :668-759,
_Transformer_FrameAtt :762-853), `losses.py:5-12`, `pytorch_utils.py:80-93`
and `optim.Adam(amsgrad=True)` as configured at `main.py:144-145`.  State is a flat dict
keyed exactly like the reference `state_dict()`.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import frontend

MODEL_TYPES = ("Cnn_9layers_FrameMax", "Cnn_9layers_FrameAvg", "Cnn_9layers_FrameAtt",
               "Cnn_9layers_Gru_FrameAvg", "Cnn_9layers_Gru_FrameAtt",
               "Cnn_9layers_Transformer_FrameAvg", "Cnn_9layers_Transformer_FrameAtt")
N_HEAD, D_HEAD = 8, 64                   # MultiHead(n_head=8, d_model=512, d_k=d_v=64, dropout=0.2), models.py:702-707
P_DROP_ATTN, P_DROP_FC = 0.1, 0.2        # ScaledDotProductAttention(attn_dropout=0.1) :590; MultiHead dropout :638
CLASSES = 17
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _bn_keys(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
            (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())]


def state_layout(model_type):
    """Ordered (key, shape) list == reference `state_dict()` order."""
    assert model_type in MODEL_TYPES
    lay = [("spectrogram_extractor.stft.conv_real.weight", (513, 1, 1024)),
           ("spectrogram_extractor.stft.conv_imag.weight", (513, 1, 1024)),
           ("logmel_extractor.melW", (513, 64))]
    lay += _bn_keys("bn0", 64)
    cin = 1
    for i, cout in enumerate((64, 128, 256, 512), start=1):
        p = "conv_block%d" % i
        lay += [(p + ".conv1.weight", (cout, cin, 3, 3)), (p + ".conv2.weight", (cout, cout, 3, 3))]
        lay += _bn_keys(p + ".bn1", cout) + _bn_keys(p + ".bn2", cout)
        cin = cout
    if "Gru" in model_type:
        for sfx in ("", "_reverse"):
            lay += [("gru.weight_ih_l0" + sfx, (768, 512)), ("gru.weight_hh_l0" + sfx, (768, 256)),
                    ("gru.bias_ih_l0" + sfx, (768,)), ("gru.bias_hh_l0" + sfx, (768,))]
    if "Transformer" in model_type:
        for nm in ("w_qs", "w_ks", "w_vs"):
            lay += [("multihead.%s.weight" % nm, (512, 512)), ("multihead.%s.bias" % nm, (512,))]
        lay += [("multihead.layer_norm.weight", (512,)), ("multihead.layer_norm.bias", (512,)),      # never used (:660-665)
                ("multihead.fc.weight", (512, 512)), ("multihead.fc.bias", (512,))]
    if model_type.endswith("FrameAtt"):
        lay += [("att_block.att.weight", (CLASSES, 512, 1)), ("att_block.att.bias", (CLASSES,)),
                ("att_block.cla.weight", (CLASSES, 512, 1)), ("att_block.cla.bias", (CLASSES,))]
        lay += _bn_keys("att_block.bn_att", CLASSES)
    else:
        lay += [("fc.weight", (CLASSES, 512)), ("fc.bias", (CLASSES,))]
    return lay


FROZEN_KEYS = ("spectrogram_extractor.stft.conv_real.weight", "spectrogram_extractor.stft.conv_imag.weight",
               "logmel_extractor.melW")


def is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def recipe_state(model_type, seed=0):
    """Seeded numpy weight recipe iterated over the state_dict keys in order, so that no 19 MB
    weight file has to be committed: the same values are regenerated wherever they are needed
    (golden generation with the reference here; parity tests and smoke() on the GPU box)."""
    wr, wi = frontend.dft_weights()
    st = OrderedDict()
    for idx, (key, shape) in enumerate(state_layout(model_type)):
        rs = np.random.RandomState(seed * 1000 + idx)
        if key == FROZEN_KEYS[0]:
            v = wr
        elif key == FROZEN_KEYS[1]:
            v = wi
        elif key == FROZEN_KEYS[2]:
            v = frontend.mel_matrix()
        elif key.endswith("num_batches_tracked"):
            v = np.array(3, dtype=np.int64)
        elif key.endswith("running_mean"):
            v = (rs.randn(*shape) * 0.2 + (-20.0 if key.startswith("bn0") else 0.1)).astype(np.float32)
        elif key.endswith("running_var"):
            v = ((0.5 + rs.rand(*shape)) * (60.0 if key.startswith("bn0") else 1.0)).astype(np.float32)
        elif ".bn" in key or key.startswith("bn0") or ".layer_norm" in key:
            v = (1.0 + 0.1 * rs.randn(*shape)).astype(np.float32) if key.endswith("weight") \
                else (0.1 * rs.randn(*shape)).astype(np.float32)
        elif key.endswith("bias") or "bias_" in key:
            v = (0.05 * rs.randn(*shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = (rs.uniform(-1.0, 1.0, size=shape) * math.sqrt(3.0 / fan_in) * 1.4).astype(np.float32)
        st[key] = torch.from_numpy(np.ascontiguousarray(v)).reshape(shape)
    return st


def flipfree_state(model_type, seed=0, beta=24.0):
    """recipe_state with every ConvBlock BatchNorm bias at +beta (gamma keeps its ~1): the normalised pre-activations
    (min xhat = -17 on the fixtures' data: the zero padding makes border pixels outliers in proportion to beta) never reach zero, so no ReLU mask (models.py:102-103) can differ between two evaluations of the same
    step -- what is left of a gradient difference is arithmetic alone.  The trunk's features then sit around +beta, which must
    not saturate the head:
      * FrameAvg / FrameMax / FrameAtt (fc, AttBlock read the features directly): the rows of the head matrices are CENTRED
        (zero sum over the 512 channels).  A training-mode BatchNorm pins the batch mean of every channel to exactly beta, so
        a zero-sum row cancels the offset whatever its scale, and the logits keep the recipe's O(1) spread ACROSS FRAMES:
        sigma' varies from frame to frame and the loss gradient that enters the trunk is not frame-constant.  (Round 5 scaled
        these matrices by 0.16 / beta instead: logits within 3e-3 of each other, a frame-constant gradient -- which every
        BatchNorm backward annihilates -- and trunk gradients that were cancellation residue, 1e-9 .. 1e-11 of the head's.)
      * GRU / MultiHead input projections: x 0.4 / beta (bounded by tanh / the softmax behind them; unchanged).
    Used by the flip-free whole-model gradient fixtures (tests/golden/make_golden.py --flipfree) together with flipfree_waves."""
    st = recipe_state(model_type, seed)
    direct_head = "Gru" not in model_type and "Transformer" not in model_type
    for key in list(st.keys()):
        if key.startswith("conv_block") and ".bn" in key and key.endswith(".bias"):
            st[key] = torch.full_like(st[key], float(beta))
        elif direct_head and key in ("fc.weight", "att_block.att.weight", "att_block.cla.weight"):
            st[key] = st[key] - st[key].mean(dim=1, keepdim=True)
        elif key.startswith("gru.weight_ih") or key in ("multihead.w_qs.weight", "multihead.w_ks.weight", "multihead.w_vs.weight"):
            st[key] = st[key] * (0.4 / beta)
    return st


def flipfree_waves(seed, n, length):
    """Input of the flip-free fixtures: noise whose level, slow amplitude envelope and an added tone differ from clip to clip and
    vary WITHIN a clip, so that neither the clips of a batch nor the frames of a clip are statistically alike (stationary noise
    clips have near-identical time-averaged features -- with BatchNorm pinning the batch mean, a clip-level loss then hardly
    depends on the trunk at all).  float32 (n, length), deterministic in `seed`."""
    rs = np.random.RandomState(seed)
    x = rs.randn(n, length) * 0.1
    t = np.arange(length) / 32000.0
    for i in range(n):
        env = 0.35 + 0.65 * (0.5 + 0.5 * np.sin(2 * np.pi * (0.4 + 0.3 * i) * t + rs.rand() * 6.28)) ** 2
        tone = 0.05 * (i % 3) * np.sin(2 * np.pi * (300.0 * (1 + i)) * t)
        x[i] = x[i] * env * (0.5 + 0.25 * i) + tone * env
    return x.astype(np.float32)


# --------------------------------------------------------------------------------------------
# forward pieces

def do_mixup(x, lam):
    """pytorch_utils.py:80-93: out[i] = lam[2i]*x[2i] + lam[2i+1]*x[2i+1]."""
    shape = [-1] + [1] * (x.dim() - 1)
    return x[0::2] * lam[0::2].reshape(shape) + x[1::2] * lam[1::2].reshape(shape)


def _bn(x, st, prefix, training, track):
    """nn.BatchNorm2d forward (eps 1e-5, momentum 0.1; biased var to normalise, unbiased to track)."""
    rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    if training and not track:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, st[prefix + ".weight"], st[prefix + ".bias"], training, BN_MOMENTUM, BN_EPS)
    if training and track:
        st[prefix + ".num_batches_tracked"] = st[prefix + ".num_batches_tracked"] + 1
    return y


def conv_block(x, st, prefix, pool, training, track, pool_type="avg"):
    """models.py:99-115 (pool_type 'avg' is what every model selects; 'max' / 'avg+max': :104-111)."""
    x = F.relu(_bn(F.conv2d(x, st[prefix + ".conv1.weight"], padding=1), st, prefix + ".bn1", training, track))
    x = F.relu(_bn(F.conv2d(x, st[prefix + ".conv2.weight"], padding=1), st, prefix + ".bn2", training, track))
    if pool_type == "max":
        return F.max_pool2d(x, kernel_size=pool)
    if pool_type == "avg":
        return F.avg_pool2d(x, kernel_size=pool)
    if pool_type == "avg+max":
        return F.avg_pool2d(x, kernel_size=pool) + F.max_pool2d(x, kernel_size=pool)
    raise Exception("Incorrect argument!")


def gru_bidir(x, st):
    """nn.GRU(512,256,batch_first,bidirectional) forward, h0=0, gate order (r,z,n), b_hn inside r*(.).
    models.py:529-530, :565-567.  x (B,T,512) -> (B,T,512)."""
    B, T, _ = x.shape
    outs = []
    for sfx, order in (("", range(T)), ("_reverse", range(T - 1, -1, -1))):
        w_ih, w_hh = st["gru.weight_ih_l0" + sfx], st["gru.weight_hh_l0" + sfx]
        b_ih, b_hh = st["gru.bias_ih_l0" + sfx], st["gru.bias_hh_l0" + sfx]
        gi_all = x @ w_ih.t() + b_ih
        h = x.new_zeros(B, 256)
        hs = [None] * T
        for t in order:
            gi = gi_all[:, t]
            gh = h @ w_hh.t() + b_hh
            r = torch.sigmoid(gi[:, :256] + gh[:, :256])
            z = torch.sigmoid(gi[:, 256:512] + gh[:, 256:512])
            n = torch.tanh(gi[:, 512:] + r * gh[:, 512:])
            h = (1.0 - z) * n + z * h
            hs[t] = h
        outs.append(torch.stack(hs, dim=1))
    return torch.cat(outs, dim=2)


def att_block(x, st, activation="sigmoid", temperature=1.0):
    """models.py:118-149 (every model of the reference: 'sigmoid', temperature 1).  x (B,n_in,T) -> clip (B,n_out),
    norm_att (B,n_out,T), cla (B,n_out,T)."""
    tmp = F.conv1d(x, st["att_block.att.weight"], st["att_block.att.bias"])
    tmp = torch.clamp(tmp, -10, 10)
    att = torch.exp(tmp / temperature) + 1e-6
    norm_att = att / torch.sum(att, dim=2)[:, :, None]
    cla = F.conv1d(x, st["att_block.cla.weight"], st["att_block.cla.bias"])
    if activation == "sigmoid":
        cla = torch.sigmoid(cla)
    elif activation != "linear":
        raise ValueError(activation)
    return torch.sum(norm_att * cla, dim=2), norm_att, cla


def dropout_masks(seed, B, T):
    """Seeded KEEP masks of the two dropouts of MultiHead in training mode: attention (N_HEAD*B, T, T) with row index
    head*B + b (the (n*b) layout of models.py:651-657) and fc output (B, T, 512).  The reference draws them from torch's
    RNG; tests fix them (here, in the golden generator and in the product) so that training parity is checkable."""
    rs = np.random.RandomState(seed)
    return (torch.from_numpy(rs.rand(N_HEAD * B, T, T) >= P_DROP_ATTN), torch.from_numpy(rs.rand(B, T, 512) >= P_DROP_FC))


def multihead(x, st, training=False, masks=None):
    """MultiHead.forward(x, x, x) (models.py:641-665) + ScaledDotProductAttention (:596-609): per-head scaled
    dot-product self-attention, output projection, dropout, ReLU; no residual, no layer norm.  x (B,T,512) -> (B,T,512)."""
    B, T, _ = x.shape
    def proj(nm):
        y = F.linear(x, st["multihead.%s.weight" % nm], st["multihead.%s.bias" % nm]).view(B, T, N_HEAD, D_HEAD)
        return y.permute(2, 0, 1, 3).reshape(N_HEAD * B, T, D_HEAD)
    q, k, v = proj("w_qs"), proj("w_ks"), proj("w_vs")
    attn = torch.softmax(torch.bmm(q, k.transpose(1, 2)) / math.sqrt(D_HEAD), dim=2)
    if training:
        assert masks is not None, "training-mode MultiHead needs explicit dropout masks (oracle.model.dropout_masks)"
        attn = attn * masks[0].to(attn.dtype) / (1.0 - P_DROP_ATTN)
    out = torch.bmm(attn, v).view(N_HEAD, B, T, D_HEAD).permute(1, 2, 0, 3).reshape(B, T, N_HEAD * D_HEAD)
    out = F.linear(out, st["multihead.fc.weight"], st["multihead.fc.bias"])
    if training:
        out = out * masks[1].to(out.dtype) / (1.0 - P_DROP_FC)
    return F.relu(out)


def interpolate(x, ratio):
    """models.py:58-69: pure repeat along time."""
    (b, t, c) = x.shape
    return x[:, :, None, :].repeat(1, 1, ratio, 1).reshape(b, t * ratio, c)


def trunk(logmel, st, training, mixup_lambda=None, stripes=None, track=True):
    """models.py:287-303 (identical in every model): bn0 on the mel axis, SpecAugment, mixup,
    four ConvBlocks, freq-mean.  logmel (B2,1,T,64) -> (B,512,T/8)."""
    x = _bn(logmel.transpose(1, 3), st, "bn0", training, track).transpose(1, 3)
    if training:
        if stripes is None:
            stripes = frontend.draw_specaug_stripes(x.shape[0], x.shape[2], x.shape[3])
        x = frontend.apply_specaug(x, stripes)
    if training and mixup_lambda is not None:
        x = do_mixup(x, mixup_lambda)
    x = conv_block(x, st, "conv_block1", (2, 2), training, track)
    x = conv_block(x, st, "conv_block2", (2, 2), training, track)
    x = conv_block(x, st, "conv_block3", (2, 2), training, track)
    x = conv_block(x, st, "conv_block4", (1, 1), training, track)
    return torch.mean(x, dim=3)


def head(model_type, x, st, training=False, masks=None):
    """x (B,512,T') -> output dict.  FrameAvg models.py:306-319, FrameMax :221-234,
    FrameAtt :388-400, Gru_* :565-581, Transformer_* :740-759 / :834-853."""
    if "Gru" in model_type:
        x = gru_bidir(x.transpose(1, 2), st).transpose(1, 2)
    if "Transformer" in model_type:
        x = multihead(x.transpose(1, 2), st, training, masks).transpose(1, 2)
    if model_type.endswith("FrameAtt"):
        clip, _, cla = att_block(x, st)
        return {"framewise_output": interpolate(cla.transpose(1, 2), 8), "clipwise_output": clip,
                "embedding": x if "Transformer" in model_type else cla}      # :398/:579 return cla, :851 returns x
    frame = torch.sigmoid(F.linear(x.transpose(1, 2), st["fc.weight"], st["fc.bias"]))
    frame = interpolate(frame, 8)
    if model_type.endswith("FrameMax"):
        clip = torch.max(frame, dim=1)[0]
    else:
        clip = torch.mean(frame, dim=1)
    return {"framewise_output": frame, "clipwise_output": clip, "embedding": x}


def forward(model_type, st, waveform, training=False, mixup_lambda=None, stripes=None, track=True, dropout_seed=None):
    """Whole-model forward == `Model.forward(input, mixup_lambda)` (models.py:279-319 etc.).  `dropout_seed` fixes the
    MultiHead dropout masks of the Transformer models in training mode (dropout_masks)."""
    lm = frontend.logmel(waveform)
    x = trunk(lm, st, training, mixup_lambda, stripes, track)
    masks = None
    if training and "Transformer" in model_type:
        assert dropout_seed is not None, "Transformer models in training mode need dropout_seed"
        masks = dropout_masks(dropout_seed, x.shape[0], x.shape[2])
    return head(model_type, x, st, training, masks)


def clip_bce(output_dict, target_dict):
    """losses.py:5-12."""
    return F.binary_cross_entropy(output_dict["clipwise_output"], target_dict["target"])


def trainable_keys(model_type):
    return [k for k, _ in state_layout(model_type) if k not in FROZEN_KEYS and not is_buffer(k)]


def unused_keys(model_type):
    """Trainable parameters that never receive a gradient (the reference's Adam skips them: .grad is None)."""
    out = ["att_block.bn_att.weight", "att_block.bn_att.bias"] if model_type.endswith("FrameAtt") else []
    if "Transformer" in model_type:
        out += ["multihead.layer_norm.weight", "multihead.layer_norm.bias"]
    return out


def adam_amsgrad_step(p, g, m, v, vmax, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam(amsgrad=True, weight_decay=0) single-tensor update (main.py:144-145)."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    torch.maximum(vmax, v, out=vmax)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (vmax.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)

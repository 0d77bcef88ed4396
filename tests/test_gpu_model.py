"""GPU parity of the whole models vs the golden vectors (genuine reference outputs) and the CPU oracle.
Gate (BASELINE.json north_star / SURVEY.md §8d): clipwise/framewise max-abs error <= 1e-4 fp32, eval and train
mode (fixed stripes / lambda); gradient slices within 1e-3 relative."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import frontend as ofe
from oracle import model as om

# per (model, convolution path) floor of the gradient gate, where it is NOT the common 2e-3 (round 3 let the split-f16 path have
# 3e-3 everywhere).  FrameMax: arg-max pooling sends the whole gradient of a clip and class through ONE frame, so a single ReLU
# unit of block 4 whose pre-activation is within rounding distance of zero -- active in one fp32 evaluation, inactive in
# another -- moves every tensor below it by 1.5e-3 .. 2.8e-3 (profiles/r02/grad_report_FrameMax.txt; the reference's own float32
# run differs from its float64 run the same way).  Which evaluation flips is decided by the last bit: with the round-4 operand
# pairs the split-f16 path reads 2.1e-3 .. 2.6e-3 on block 1-2 tensors of this fixture, the fp32 path stays below 2e-3.
GRAD_FLOOR = {("Cnn_9layers_FrameMax", "sf16"): 3e-3}
SEEDS = {mt: i + 1 for i, mt in enumerate(om.MODEL_TYPES)}
CTOR = (32000, 1024, 320, 64, 50, 14000, 17)


def waves(seed, n, length):
    return (np.random.RandomState(seed).randn(n, length) * 0.1).astype(np.float32)


def targets(seed, n):
    return (np.random.RandomState(seed).rand(n, 17) < 0.2).astype(np.float32)


def summarize(t):
    f = t.detach().reshape(-1).double().cpu()
    return np.array([f.sum().item(), f.abs().sum().item()] + f[:14].tolist() + [0.0] * max(0, 14 - f.numel()),
                    dtype=np.float64)[:16]


def check_summary(got, want, rtol, what, slack=0.0):
    scale = max(abs(want[1]), 1e-12)
    assert abs(got[0] - want[0]) <= rtol * scale + slack, (what, got[0], want[0])
    assert abs(got[1] - want[1]) <= rtol * scale + slack, (what, got[1], want[1])
    np.testing.assert_allclose(got[2:], want[2:], rtol=rtol * 50, atol=rtol * scale / 10 + slack, err_msg=what)


def dropout_kw(mt, seed, B, T):
    """Product keyword for the fixed MultiHead dropout masks of the Transformer models (same recipe as the fixtures)."""
    if "Transformer" not in mt:
        return {}
    ma, mf = om.dropout_masks(seed, B, T)
    return {"dropout_masks": (ma.cuda(), mf.cuda())}


def fp64_oracle_grads(mt, seed, stripes, unused, dropout_seed=None):
    """Step-0 gradients of the golden training recipe evaluated by the CPU oracle in float64 (ground truth)."""
    st = {k: (v.double() if v.is_floating_point() else v) for k, v in om.recipe_state(mt, seed).items()}
    ofe._CACHE.clear()
    consts = ofe._consts()
    for k in list(consts):
        consts[k] = consts[k].double()
    try:
        keys = [k for k in om.trainable_keys(mt) if k not in unused]
        for k in keys:
            st[k].requires_grad_(True)
        rs = np.random.RandomState(1234)
        xw = torch.from_numpy(waves(700 + 10 * seed, 8, 32000)).double()
        tg = torch.from_numpy(targets(800 + 10 * seed, 8)).double()
        lam = torch.from_numpy(ofe.mixup_lambdas(8, rs).astype(np.float32)).double()
        o = om.forward(mt, st, xw, training=True, mixup_lambda=lam, stripes=stripes, dropout_seed=dropout_seed)
        loss = om.clip_bce(o, {"target": om.do_mixup(tg, lam)})
        grads = torch.autograd.grad(loss, [st[k] for k in keys])
        return {k: g.numpy() for k, g in zip(keys, grads)}
    finally:
        ofe._CACHE.clear()


@pytest.fixture(params=["sf16", "fp32"])
def conv_path(request, monkeypatch):
    """Both convolution paths through the WHOLE model: the default split-f16 MFMA kernels and the fp32 MFMA (Winograd)
    kernels that `ops.USE_SF16 = False` / SED_USE_SF16=0 selects -- the fallback of the non-finite guard."""
    from sound_event_detection_dcase2017_task4_amd import ops
    monkeypatch.setattr(ops, "USE_SF16", request.param == "sf16")
    return request.param


def build(mt):
    from sound_event_detection_dcase2017_task4_amd.pytorch import models
    m = getattr(models, mt)(*CTOR)
    m.load_state_dict(om.recipe_state(mt, SEEDS[mt]))
    return m.to("cuda")


@pytest.mark.parametrize("mt", om.MODEL_TYPES)
def test_eval_forward_matches_reference(mt, golden_dir, conv_path):
    fx = np.load(os.path.join(golden_dir, mt + ".npz"))
    m = build(mt).eval()
    with torch.no_grad():
        o = m(torch.from_numpy(waves(100 + SEEDS[mt], 4, 32000)).cuda())
    assert o["framewise_output"].shape == (4, 96, 17) and o["clipwise_output"].shape == (4, 17)
    assert np.abs(o["clipwise_output"].cpu().numpy() - fx["eval_clip"]).max() < 1e-4
    assert np.abs(o["framewise_output"].cpu().numpy()[:, ::8] - fx["eval_frame"]).max() < 1e-4
    fw = o["framewise_output"].cpu().numpy()
    assert np.array_equal(fw[:, 0::8], fw[:, 7::8])                   # x8 repeat
    check_summary(summarize(o["embedding"].contiguous()), fx["eval_embedding"], 1e-4, "embedding")
    assert o["embedding"].shape == ((4, 17, 12) if (mt.endswith("Att") and "Transformer" not in mt) else (4, 512, 12))


def test_eval_forward_10s_clip(golden_dir):
    mt = "Cnn_9layers_FrameAvg"
    fx = np.load(os.path.join(golden_dir, mt + ".npz"))
    m = build(mt).eval()
    with torch.no_grad():
        o = m(torch.from_numpy(waves(200 + SEEDS[mt], 2, 320000)).cuda())
    assert o["framewise_output"].shape == (2, 1000, 17) and o["embedding"].shape == (2, 512, 125)
    assert np.abs(o["clipwise_output"].cpu().numpy() - fx["eval10_clip"]).max() < 1e-4
    assert np.abs(o["framewise_output"].cpu().numpy()[:, ::8] - fx["eval10_frame"]).max() < 1e-4


@pytest.mark.parametrize("mt", om.MODEL_TYPES)
def test_train_forward_matches_reference(mt, golden_dir, conv_path):
    fx = np.load(os.path.join(golden_dir, mt + ".npz"))
    seed = SEEDS[mt]
    m = build(mt).train()
    lam = torch.from_numpy(fx["train_lambda"]).cuda()
    torch.manual_seed(500 + seed)                                      # same global-RNG stream as the reference run
    with torch.no_grad():
        o = m(torch.from_numpy(waves(300 + seed, 6, 32000)).cuda(), lam, **dropout_kw(mt, int(fx["train_dropout_seed"]), 3, 12))
    assert o["clipwise_output"].shape == (3, 17)
    assert np.abs(o["clipwise_output"].cpu().numpy() - fx["train_clip"]).max() < 1e-4
    assert np.abs(o["framewise_output"].cpu().numpy()[:, ::8] - fx["train_frame"]).max() < 1e-4
    sd = m.state_dict()
    np.testing.assert_allclose(sd["bn0.running_mean"].cpu().numpy(), fx["train_bn0_running_mean"], rtol=1e-5)
    np.testing.assert_allclose(sd["bn0.running_var"].cpu().numpy(), fx["train_bn0_running_var"], rtol=1e-5)
    np.testing.assert_allclose(sd["conv_block4.bn2.running_var"].cpu().numpy(), fx["train_b4bn2_running_var"], rtol=1e-4)
    assert int(sd["bn0.num_batches_tracked"]) == 4


@pytest.mark.parametrize("mt", om.MODEL_TYPES)
def test_three_train_steps_match_reference(mt, golden_dir, conv_path):
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import get_loss_func
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import move_data_to_device, do_mixup
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    fx = np.load(os.path.join(golden_dir, mt + ".npz"))
    seed = SEEDS[mt]
    m = build(mt)
    opt = FusedAdamAmsgrad(m, lr=1e-3, betas=(0.9, 0.999), eps=1e-08)
    loss_func = get_loss_func("clip_bce")
    rs = np.random.RandomState(1234)
    unused = set(fx["grad0_none_keys"].tolist())
    structural_zero = set()        # true gradient is 0: Adam turns the rounding noise into a +-lr walk in BOTH implementations
    for it in range(3):
        xw = move_data_to_device(waves(700 + 10 * seed + it, 8, 32000), "cuda")
        tg = move_data_to_device(targets(800 + 10 * seed + it, 8), "cuda")
        lam = move_data_to_device(ofe.mixup_lambdas(8, rs), "cuda")
        torch.manual_seed(900 + 10 * seed + it)
        m.train()
        o = m(xw, lam, **dropout_kw(mt, int(fx["step_dropout_seeds"][it]), 4, 12))
        loss = loss_func(o, {"target": do_mixup(tg, lam)})
        # step 0 is a pure forward (1e-4 gate); later steps inherit Adam's sign-like amplification of gradient noise
        assert abs(loss.item() - fx["step_losses"][it]) < (1e-4 if it == 0 else 2e-3), (it, loss.item(), fx["step_losses"][it])
        opt.zero_grad()
        loss.backward()
        if it == 0:
            # Gradient gate.  Per-channel gradients are random-sign sums, so ONE ReLU whose pre-activation sits within
            # an fp32 ulp of zero moves a gradient entry by O(1/sqrt(N)) of its value.  On this tiny batch the
            # reference's own fp32 CPU path therefore differs from an fp64 evaluation of the same step by up to
            # 5e-3 (FrameAvg) / 1.2e-2 (FrameMax) of the tensor max, and by a different amount for a different
            # thread count (8 vs 1 threads: relative L2 2.3e-3 on bn0.* of Gru_FrameAtt; vs fp64 up to 3.9e-3 on
            # FrameMax conv_block1.conv1 -- measured while building the fixtures).  When no ReLU flips, this HIP path
            # agrees with fp64 to 3e-6 on every tensor (Gru_FrameAvg fixture).  The yardstick is thus the fp64 oracle
            # with a statistical gate: relative L2 error <= 1e-2 and max error <= 4e-2 of the tensor max (~2.5x the
            # reference's own jitter).  The strict per-kernel gradient checks (<= 3e-4) live in tests/test_gpu_ops.py.
            g64 = fp64_oracle_grads(mt, seed, fx["step_stripes"][0], sorted(unused), int(fx["step_dropout_seeds"][0]))
            report, bad = {}, {}
            for k, p in m.named_parameters():
                if not p.requires_grad or k in unused:
                    continue
                truth = g64[k]
                if np.abs(truth).max() < 1e-7:          # structurally zero (att_block.att.bias)
                    assert p.grad.abs().max().item() < 1e-6
                    structural_zero.add(k)
                    continue
                d = p.grad.double().cpu().numpy() - truth
                l2 = float(np.sqrt((d ** 2).sum() / (truth ** 2).sum()))
                mx = float(np.abs(d).max() / np.abs(truth).max())
                ref_mx = (float(np.abs(fx["gradfull0/" + k].astype(np.float64) - truth).max() / np.abs(truth).max())
                          if ("gradfull0/" + k) in fx.files else None)
                report[k] = (l2, mx, ref_mx)
                if l2 > 1e-2 or mx > 4e-2:
                    bad[k] = report[k]
            print("grad errors vs fp64 (l2, max, reference-fixture max):",
                  sorted(report.items(), key=lambda kv: -kv[1][0])[:4])
            assert not bad, bad
        opt.step()
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    for k, v in m.state_dict().items():
        if k in om.FROZEN_KEYS:
            continue
        got, want = summarize(v.float()), fx["after3/" + k]
        if k not in trainable or k in unused:
            check_summary(got, want, 2e-3, "after3 " + k)              # BN running statistics / untouched tensors: relative
            continue
        # Adam moves every entry by <= lr per step, i.e. by <= 3*lr = 3e-3 in total, in the direction sign(gradient):
        # an entry whose gradient is noise-level (a ReLU flip, see above) may move the other way, and on this tiny batch the
        # later steps inherit gradient differences of up to 6e-3 (printed above).  Tolerances are stated relative to that
        # 3*lr travel: nearly all leading entries agree to 15 % of it (measured: <= 11 %), none differs by more than the
        # 2 x 3*lr two opposite walks can produce, and the sums may differ by the random walk of the few flipped entries.
        # (The low-noise fixture below applies the strict version: mean difference <= 5 % of the travel.)
        travel = 3 * 1e-3
        d = np.abs(got[2:] - want[2:])[:min(14, v.numel())]
        assert d.max() <= 2 * travel * 1.01, ("after3 entries " + k, d)
        if k in structural_zero:
            continue
        assert (d <= 0.15 * travel).sum() >= len(d) - 2, ("after3 entries " + k, d)
        walk = 2 * travel * np.sqrt(0.02 * v.numel()) + 0.15 * travel       # <= 2 % of the entries flipped
        assert abs(got[0] - want[0]) <= walk + 2e-3 * abs(want[1]), ("after3 sum " + k, got[0], want[0], walk)
        assert abs(got[1] - want[1]) <= walk + 2e-3 * abs(want[1]), ("after3 abs-sum " + k, got[1], want[1], walk)


def test_full_size_batch_properties():
    """BASELINE configs[1] shape (B=32 here to bound memory/time; 10 s clips): train-mode outputs are finite
    probabilities, mixup with lambda = (1, 0) pairs equals the un-mixed even clips, and eval is deterministic."""
    mt = "Cnn_9layers_FrameAvg"
    m = build(mt)
    x = torch.from_numpy(waves(1, 16, 320000)).cuda()
    m.eval()
    with torch.no_grad():
        a = m(x)["clipwise_output"]
        b = m(x)["clipwise_output"]
    assert torch.equal(a, b) and torch.isfinite(a).all() and (a >= 0).all() and (a <= 1).all()
    lam = torch.tensor([1.0, 0.0] * 8, device="cuda")
    stripes = np.zeros((16, 8), dtype=np.int32)                       # no stripes
    m.train()
    with torch.no_grad():
        mixed = m(x, lam, specaug_stripes=stripes)["clipwise_output"]
    assert mixed.shape == (8, 17) and torch.isfinite(mixed).all()


def test_full_size_train_step_properties():
    """BASELINE.json configs[1] at FULL size (B=256 post-mixup clips = 512 x 10 s waveforms, mixup + SpecAugment):
    size-independent properties instead of an oracle run (the CPU oracle needs minutes for this batch)."""
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
    mt = "Cnn_9layers_FrameAvg"
    B2, L = 512, 320000
    g = torch.Generator(device="cuda").manual_seed(3)
    wave = (torch.randn((B2, L), generator=g, device="cuda") * 0.1).clamp_(-1, 1)
    target = (torch.rand((B2, 17), generator=g, device="cuda") < 0.2).float()
    lam = torch.from_numpy(ofe.mixup_lambdas(B2, np.random.RandomState(1234)).astype(np.float32)).cuda()

    def one_step():
        m = build(mt).train()
        opt = FusedAdamAmsgrad(m, lr=1e-3)
        torch.manual_seed(77)
        out = m(wave, lam)
        loss = clip_bce(out, {"target": do_mixup(target, lam)})
        opt.zero_grad()
        loss.backward()
        gsum = opt.flat_grad.double().abs().sum().item()
        opt.step()
        return out, loss.item(), gsum, opt.flat.double().sum().item(), m

    out, loss, gsum, psum, m = one_step()
    assert out["clipwise_output"].shape == (256, 17) and out["framewise_output"].shape == (256, 1000, 17)
    assert out["embedding"].shape == (256, 512, 125)
    p = out["clipwise_output"]
    assert torch.isfinite(p).all() and (p >= 0).all() and (p <= 1).all() and np.isfinite(loss) and gsum > 0
    fw = out["framewise_output"]
    assert torch.equal(fw[:, 0::8], fw[:, 7::8])                                   # interpolate = x8 repeat
    assert torch.allclose(fw.mean(dim=1), p, atol=1e-5)                            # FrameAvg: clip = mean over frames
    # BN bookkeeping: batch mean of log-mel noise is far from the recipe's running mean -> moved by momentum 0.1
    assert int(m.bn0.num_batches_tracked) == 4
    # determinism: the same step twice is bit-identical (no atomics on the value path)
    out2, loss2, gsum2, psum2, _ = one_step()
    assert loss2 == loss and gsum2 == gsum and psum2 == psum and torch.equal(out2["clipwise_output"], p)
    # mixup linearity of the targets and of the log-mel stage at full size
    t2 = do_mixup(target, lam)
    assert torch.allclose(t2, target[0::2] * lam[0::2, None] + target[1::2] * lam[1::2, None], atol=1e-6)
    lm = m.extract_logmel(wave[:64])
    lm2 = m.extract_logmel(wave[:64] * 0.5)
    assert (lm - lm2 - 20.0 * np.log10(2.0)).abs().max().item() < 1e-3
    # conv linearity on the largest layer (block-1 conv2, 16.4 M pixels x 64 -> 64): conv(2x) == 2 conv(x)
    x = torch.randn((256, 1001, 64, 64), generator=g, device="cuda")
    wf, _ = ops._pack(m.conv_block1.conv2.weight.detach())
    y1 = ops._conv_igemm(x, wf, 256, 1001, 64, 64, 64)
    x.mul_(2.0)
    y2 = ops._conv_igemm(x, wf, 256, 1001, 64, 64, 64)
    assert (y2 - 2.0 * y1).abs().max().item() <= 1e-5 * y1.abs().max().item()
    del x, y1, y2
    # the same layer on the split-f16 kernel: the operand scale is a power of two taken from the device amax, so doubling the
    # input halves the scale, the (hi, lo) operands are the same bits and the result is EXACTLY twice the first one
    x = torch.randn((256, 1001, 64, 64), generator=g, device="cuda")
    pk = ops.pack_sf16(m.conv_block1.conv2.weight.detach())
    y1 = ops.conv3x3_sf16(x, pk, 256, 1001, 64, 64, 64, x_amax=ops.amax_of(x))
    x.mul_(2.0)
    y2 = ops.conv3x3_sf16(x, pk, 256, 1001, 64, 64, 64, x_amax=ops.amax_of(x))
    assert torch.equal(y2, 2.0 * y1)
    del x, y1, y2


def test_full_size_eval_batch_is_row_consistent():
    """Eval-mode forward of 256 ten-second clips = 8 copies of 32 distinct ones: every copy's outputs are bit-identical to the
    32-clip run (nothing depends on the batch position or on the workgroup a clip lands in), and the gradient of a sum loss
    over the 256 rows w.r.t. the conv weights is 8x the 32-row one (eval-mode BatchNorm does not couple rows)."""
    mt = "Cnn_9layers_FrameAvg"
    m = build(mt).eval()
    x32 = torch.from_numpy(waves(5, 32, 320000)).cuda()
    x256 = x32.repeat(8, 1)
    with torch.no_grad():
        a = m(x32)["clipwise_output"]
        b = m(x256)["clipwise_output"]
    assert torch.equal(b.view(8, 32, 17), a.unsqueeze(0).expand(8, 32, 17))
    grads = []
    for x in (x32[:8], x32[:8].repeat(4, 1)):
        m.zero_grad()
        m(x)["clipwise_output"].sum().backward()
        grads.append([p.grad.double().clone() for p in (m.conv_block2.conv1.weight, m.conv_block4.conv2.weight, m.fc.weight)])
    for g8, g32 in zip(*grads):
        assert (g32 - 4.0 * g8).abs().max().item() <= 2e-5 * g8.abs().max().item() * 4.0


def test_int16_waveforms_end_to_end():
    """The HDF5 storage dtype (int16, utils/features.py) fed straight to the model equals the reference's
    int16_to_float32 path (utilities.py:66-67) to within fp32 rounding of the division."""
    mt = "Cnn_9layers_FrameAtt"
    m = build(mt).eval()
    q = np.round(np.clip(waves(11, 3, 32000), -1, 1) * 32767.0).astype(np.int16)
    with torch.no_grad():
        a = m(torch.from_numpy(q).cuda())["clipwise_output"].cpu()
        b = m(torch.from_numpy((q / 32767.0).astype(np.float32)).cuda())["clipwise_output"].cpu()
    st = om.recipe_state(mt, SEEDS[mt])
    with torch.no_grad():
        ref = om.forward(mt, st, torch.from_numpy((q / 32767.0).astype(np.float32)))["clipwise_output"]
    assert (a - b).abs().max().item() < 1e-6 and (a - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("mt,B,L", [("Cnn_9layers_FrameAvg", 3, 48017), ("Cnn_9layers_Gru_FrameAtt", 5, 20000),
                                    ("Cnn_9layers_FrameMax", 1, 6400)])
def test_ragged_shapes_vs_oracle(mt, B, L):
    """Odd batch sizes and clip lengths that are not multiples of the hop / of the pooling factors (floor-mode pooling
    drops trailing frames; T = L//320 + 1): eval forward and a no-mixup train forward against the CPU oracle."""
    m = build(mt)
    x = waves(B * 7 + L, B, L)
    st = om.recipe_state(mt, SEEDS[mt])
    T8 = ((L // 320 + 1) // 8) * 8
    m.eval()
    with torch.no_grad():
        o = m(torch.from_numpy(x).cuda())
        ref = om.forward(mt, st, torch.from_numpy(x), training=False)
    assert o["framewise_output"].shape == (B, T8, 17) == tuple(ref["framewise_output"].shape)
    assert (o["clipwise_output"].cpu() - ref["clipwise_output"]).abs().max().item() < 1e-4
    assert (o["framewise_output"].cpu() - ref["framewise_output"]).abs().max().item() < 1e-4
    if B > 1:                                         # batch statistics need more than one value per channel
        torch.manual_seed(5)
        stripes = ofe.draw_specaug_stripes(B, L // 320 + 1, 64)
        m.train()
        with torch.no_grad():
            o = m(torch.from_numpy(x).cuda(), None, specaug_stripes=stripes)
            ref = om.forward(mt, st, torch.from_numpy(x), training=True, mixup_lambda=None, stripes=stripes)
        assert (o["clipwise_output"].cpu() - ref["clipwise_output"]).abs().max().item() < 1e-4


LONG_MODELS = ("Cnn_9layers_FrameAvg", "Cnn_9layers_FrameAtt", "Cnn_9layers_Gru_FrameAtt")      # BASELINE.json configs[1..3]


@pytest.mark.parametrize("mt", LONG_MODELS)
def test_full_length_clips_match_reference(mt, golden_dir, conv_path):
    """10 s clips (L = 320000 -> T = 1001 frames, T' = 125: the production sequence length of the attention pooling and of
    the BiGRU) through the genuine reference model: eval forward and train forward (mixup + SpecAugment), 1e-4 gate."""
    fx = np.load(os.path.join(golden_dir, mt + "__big.npz"))
    seed = SEEDS[mt]
    m = build(mt).eval()
    with torch.no_grad():
        o = m(torch.from_numpy(waves(200 + seed, 2, 320000)).cuda())
    assert o["framewise_output"].shape == (2, 1000, 17)
    assert np.abs(o["clipwise_output"].cpu().numpy() - fx["eval10_clip"]).max() < 1e-4
    assert np.abs(o["framewise_output"].cpu().numpy()[:, ::8] - fx["eval10_frame"]).max() < 1e-4
    m = build(mt).train()
    with torch.no_grad():
        o = m(torch.from_numpy(waves(250 + seed, 4, 320000)).cuda(), torch.from_numpy(fx["train10_lambda"]).cuda(),
              specaug_stripes=fx["train10_stripes"])
    assert o["clipwise_output"].shape == (2, 17) and o["framewise_output"].shape == (2, 1000, 17)
    assert np.abs(o["clipwise_output"].cpu().numpy() - fx["train10_clip"]).max() < 1e-4
    assert np.abs(o["framewise_output"].cpu().numpy()[:, ::8] - fx["train10_frame"]).max() < 1e-4
    np.testing.assert_allclose(m.state_dict()["conv_block4.bn2.running_mean"].cpu().numpy(), fx["train10_b4bn2_running_mean"],
                               rtol=1e-4, atol=1e-6)


def sample_index(numel, cap=2048):
    return np.arange(0, numel, max(1, -(-numel // cap)))


@pytest.mark.parametrize("mt", om.MODEL_TYPES)
def test_low_noise_training_fixture_vs_float64_reference(mt, golden_dir, conv_path):
    """Gradient gate of SURVEY.md 8(d) -- relative error <= 1e-3 -- against the genuine reference model evaluated in
    FLOAT64 on a batch large enough (32 x 2 s waveforms) that single ReLU flips no longer dominate.

    What the fixture shows about the reference itself: its own float32 gradients differ from float64 by 0.7e-3 .. 3.8e-3
    relative L2 on every tensor below block 4's second BatchNorm (`big_ref32err/*`, written by make_golden.py).  The cause
    is cancellation in the BatchNorm backward (g - mean(g) - xhat*mean(g*xhat)): the clip-level loss gradient is constant
    over the frames of a clip, so the mean that is subtracted is ~100x larger than what remains.  Whatever rounding noise
    the gradient carries into a BatchNorm backward is amplified by that factor; the 2-D Winograd dgrad kernels carry ~3x
    the rounding noise of a direct fp32 convolution, and this path indeed measures 1.0 .. 2.4x the
    reference's own float32 error on the tensors where that error is cancellation noise.  The second mechanism is the ReLU
    flip: a pre-activation within rounding distance of zero has a different mask in fp32 than in fp64, and ONE flipped
    element among the 10^6..10^7 of a layer moves every upstream gradient by ~1e-3 relative L2 (tools/convblock_precision.py:
    a ConvBlock whose forward agrees to 5e-7 shows 5e-4..2.5e-3 gradient differences, or 1e-6 when no mask flips; the
    reference shows the same between 1 and 8 CPU threads).  The number of flips scales with the forward rounding error, so
    this path flips somewhat more often than a direct fp32 convolution.  The gate is therefore, per tensor,
    max(3e-3, 3 x the reference's own float32 error) -- an explicit, data-driven allow-list (ONE flip in block 4 of this
    16-clip fixture moves every upstream tensor by 1.5e-3 .. 2.8e-3: FrameMax with the split-f16 kernels; the Winograd and the
    direct fp32 kernels flip elsewhere and read 0.8e-3 .. 3.0e-3 and 0.2e-3 .. 0.8e-3, profiles/r02/grad_report_FrameMax.txt;
    without flips every ConvBlock gradient is within 4e-6 of float64, tests/test_gpu_ops.py); the test prints how many
    tensors pass the plain 1e-3 (FrameAvg: 10 to 28 of 28 depending on where the last bit of the log-mel falls) -- and
    tensors whose true gradient is structurally zero (softmax / attention shift invariance) are checked absolutely.
    Then: three optimisation steps against the float64 reference, tolerances relative to Adam's 3*lr travel."""
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import get_loss_func
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    fx = np.load(os.path.join(golden_dir, mt + "__big.npz"))
    seed = SEEDS[mt]
    rows, L = 32, 64000
    T = L // 320 + 1
    m = build(mt)
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    opt = FusedAdamAmsgrad(m, lr=1e-3, betas=(0.9, 0.999), eps=1e-08)
    loss_func = get_loss_func("clip_bce")
    rs = np.random.RandomState(1234)
    report = {}
    structural_zero = set()        # true gradient is 0: Adam turns the rounding noise into a +-lr walk in BOTH implementations
    for it in range(3):
        xw = torch.from_numpy(waves(1700 + 10 * seed + it, rows, L)).cuda()
        tg = torch.from_numpy(targets(1800 + 10 * seed + it, rows)).cuda()
        lam = torch.from_numpy(ofe.mixup_lambdas(rows, rs).astype(np.float32)).cuda()
        m.train()
        o = m(xw, lam, specaug_stripes=fx["big_stripes"][it],
              **dropout_kw(mt, int(fx["big_dropout_seeds"][it]), rows // 2, ((T // 2) // 2) // 2))
        loss = loss_func(o, {"target": do_mixup(tg, lam)})
        # later steps inherit Adam's sign-like amplification of noise-level gradient entries.  FrameMax (arg-max pooling: one
        # frame per clip and class carries the gradient) is the most sensitive: the CPU oracle in float32 itself lands
        # 8e-5 (8 threads) or 5.8e-3 (1 thread) from the float64 loss at step 2 of this fixture, so its gate is 8e-3
        later = 8e-3 if mt.endswith("FrameMax") else 2e-3
        assert abs(loss.item() - fx["big_losses64"][it]) < (2e-5 if it == 0 else later), (it, loss.item(), fx["big_losses64"][it])
        opt.zero_grad()
        loss.backward()
        if it == 0:
            bad = {}
            for k, p in m.named_parameters():
                if ("big_g64/" + k) not in fx.files:
                    continue
                want = fx["big_g64/" + k].astype(np.float64)
                l2, _, _, mx = fx["big_g64n/" + k]
                g = p.grad.detach().double().reshape(-1).cpu().numpy()
                if mx < 1e-7:                                        # structurally zero in exact arithmetic
                    assert np.abs(g).max() < 1e-6, (k, np.abs(g).max())
                    structural_zero.add(k)
                    continue
                got = g[sample_index(g.size)]
                err = float(np.sqrt(((got - want) ** 2).sum() / max((want ** 2).sum(), 1e-300)))
                ref = float(fx["big_ref32err/" + k][0])
                gate = max(GRAD_FLOOR.get((mt, conv_path), 2e-3), 3.0 * ref)     # ONE floor for both convolution paths
                report[k] = (err, ref)
                if err > gate:
                    bad[k] = (err, ref, gate)
                # full-tensor norm against the float64 norm
                assert abs(float(np.sqrt((g ** 2).sum())) - l2) <= gate * l2, (k, float(np.sqrt((g ** 2).sum())), l2)
            top = sorted(report.items(), key=lambda kv: -kv[1][0])[:5]
            print("gradient relative L2 vs float64 (ours, reference-fp32):", top)
            n_1e3 = sum(e <= 1e-3 for e, _ in report.values())
            print("tensors where we are below 1e-3: %d of %d; below the reference's own error: %d"
                  % (n_1e3, len(report), sum(e <= r for e, r in report.values())))
            assert not bad, bad
            # the plain SURVEY 8(d) figure (1e-3) is ASSERTED where no ReLU mask lies between the tensor and the loss -- the head
            # (and the GRU / attention stack above block 4), where the error is pure arithmetic -- and for at least two trunk
            # tensors besides: below a flipped mask every upstream tensor moves by ~1e-3 in ANY fp32 evaluation (the reference's own
            # float32 run is 0.7e-3 .. 3.8e-3 from float64 on them), so which tensors pass is decided by the last bit of the
            # log-mel, not by the kernels
            head = {k: e for k, (e, _) in report.items() if not k.startswith(("conv_block", "bn0"))}
            assert head and max(head.values()) <= 1e-3, head
            assert n_1e3 >= len(head) + 2, (n_1e3, len(report))         # measured 5 .. 28 of 28-38 over models / paths / boxes
        opt.step()
    travel = 3 * 1e-3
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    worst = {"bn_rel_l2": 0.0, "mean/travel (>=1024)": 0.0, "mean/travel (<1024)": 0.0, "frac>0.1travel": 0.0}
    for k, v in m.state_dict().items():
        if ("big_after3/" + k) not in fx.files:
            continue
        want = fx["big_after3/" + k].astype(np.float64)
        got = v.detach().double().reshape(-1).cpu().numpy()[sample_index(v.numel())]
        if k not in trainable:                                       # BatchNorm running statistics (of layers whose weights
            # have walked by +-lr per step, a few per cent of the entries the other way): relative L2 of the vector
            err = float(np.sqrt(((got - want) ** 2).sum() / max((want ** 2).sum(), 1e-300)))
            worst["bn_rel_l2"] = max(worst["bn_rel_l2"], err)
            assert err <= 1e-2, (k, err)
            continue
        d = np.abs(got - want)
        assert d.max() <= 2 * travel * 1.01, (k, d.max())
        if k in structural_zero:                                     # rounding noise through Adam's normalisation: a +-lr walk
            continue
        moved = np.abs(want - before[k].double().reshape(-1).cpu().numpy()[sample_index(v.numel())])
        if moved.max() == 0:                                         # never receives a gradient (att_block.bn_att.*)
            assert d.max() == 0, k
            continue
        # (per-channel vectors of 64..512 entries: a handful of near-zero gradients whose Adam direction differs is a
        # large FRACTION of such a tensor, so the fraction criterion applies to the >= 1024-entry samples only)
        worst["mean/travel (>=1024)" if d.size >= 1024 else "mean/travel (<1024)"] = max(
            worst["mean/travel (>=1024)" if d.size >= 1024 else "mean/travel (<1024)"], float(d.mean() / travel))
        if d.size >= 1024:
            worst["frac>0.1travel"] = max(worst["frac>0.1travel"], float((d > 0.1 * travel).mean()))
            assert (d > 0.1 * travel).mean() <= 0.10, (k, float((d > 0.1 * travel).mean()))
        assert d.mean() <= (0.05 if d.size >= 1024 else 0.10) * travel, (k, float(d.mean()))
    print("after three steps, worst over tensors (gates: 1e-2, 0.05, 0.10, 0.10):", worst)


FF_ROWS, FF_L = 8, 64000


FLIPFREE_GATE = 1e-4


@pytest.mark.parametrize("mt", ["Cnn_9layers_FrameAvg", "Cnn_9layers_FrameAtt", "Cnn_9layers_Gru_FrameAtt"])
def test_flip_free_whole_model_gradients_vs_float64_reference(mt, golden_dir, conv_path):
    """The SURVEY 8(d) gradient gate (relative error <= 1e-3) on a WHOLE-MODEL training step in which no ReLU mask can flip:
    `oracle.model.flipfree_state` puts every ConvBlock BatchNorm bias at +24 (the generator asserts that the smallest
    pre-activation of the step is > 6), so what separates this path from the genuine reference evaluated in float64
    (tests/golden/<model>__flipfree.npz, make_golden.py --flipfree) is arithmetic alone -- the measurement behind the "the
    other fixtures' gate is loose because of flips" argument.  EVERY trainable tensor of the headline model (FrameAvg), of
    FrameAtt and of Gru_FrameAtt is held to relative L2 <= 1e-4 -- ten times below the 8(d) figure -- with no allow-list, on both
    convolution paths; the reference's own float32 run reads up to 1.7e-4 on the same fixtures (`ff_ref32err/*`).
    Round 6: the fixtures are no longer degenerate for the models whose head reads the features directly -- centred head rows
    (the BatchNorm-pinned +24 offset cancels, the logits keep an O(1) spread across frames) and non-stationary clips
    (`oracle.model.flipfree_waves`); round 5's FrameAvg fixture had a frame-constant loss gradient, which every BatchNorm
    backward annihilates: its trunk gradients were cancellation residue (1e-9 .. 1e-11 of the head's) and its gate 8x the
    reference's own float32 error.  The one tensor whose true gradient is structurally zero (`att_block.att.bias`: softmax
    shift invariance) is checked absolutely."""
    from sound_event_detection_dcase2017_task4_amd.pytorch import models
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import get_loss_func
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    fx = np.load(os.path.join(golden_dir, mt + "__flipfree.npz"))
    seed = SEEDS[mt]
    assert float(fx["ff_min_preact_f64"]) > 6.0 and float(fx["ff_min_preact_f32"]) > 6.0       # the premise: no mask can flip
    m = getattr(models, mt)(*CTOR)
    m.load_state_dict(om.flipfree_state(mt, seed))
    m = m.to("cuda").train()
    opt = FusedAdamAmsgrad(m, lr=1e-3)
    xw = torch.from_numpy(om.flipfree_waves(2700 + seed, FF_ROWS, FF_L)).cuda()
    tg = torch.from_numpy(targets(2800 + seed, FF_ROWS)).cuda()
    lam = torch.from_numpy(fx["ff_lambda"]).cuda()
    o = m(xw, lam, specaug_stripes=fx["ff_stripes"])
    loss = get_loss_func("clip_bce")(o, {"target": do_mixup(tg, lam)})
    assert abs(loss.item() - float(fx["ff_loss64"])) < 2e-5, (loss.item(), float(fx["ff_loss64"]))
    assert np.abs(o["clipwise_output"].detach().cpu().numpy() - fx["ff_clip64"]).max() < 1e-4
    opt.zero_grad()
    loss.backward()
    report, bad = {}, {}
    norms = {}
    for k, p in m.named_parameters():
        if ("ff_g64/" + k) not in fx.files:
            continue
        want = fx["ff_g64/" + k].astype(np.float64)
        l2, _, _, mx = fx["ff_g64n/" + k]
        g = p.grad.detach().double().reshape(-1).cpu().numpy()
        ref = float(fx["ff_ref32err/" + k])
        if k == "att_block.att.bias":                            # structurally zero in exact arithmetic (the reference's fp32 run
            assert np.abs(g).max() < 1e-6, (k, np.abs(g).max())  # is noise there): absolute check
            continue
        got = g[sample_index(g.size)]
        err = float(np.sqrt(((got - want) ** 2).sum() / max((want ** 2).sum(), 1e-300)))
        report[k] = (err, ref, FLIPFREE_GATE)
        norms[k] = float(l2)
        if err > FLIPFREE_GATE:
            bad[k] = (err, ref, FLIPFREE_GATE)
    # the fixture is NOT degenerate: every trunk tensor carries a gradient far above cancellation residue (the reference's own
    # float32 run resolves it to better than 2e-4), none is noise-level relative to the head's
    assert max(v[1] for v in report.values()) < 3e-4, max(report.items(), key=lambda kv: kv[1][1])
    print("flip-free gradient relative L2 vs float64 (ours, reference-fp32, gate), worst five:",
          sorted(report.items(), key=lambda kv: -kv[1][0] / kv[1][2])[:5])
    print("FLIPFREE_REPORT", mt, conv_path, {k: ("%.2e" % v[0], "%.2e" % v[1], "%.1e" % norms[k]) for k, v in report.items()})
    assert len(report) >= 26 and not bad, bad


@pytest.mark.parametrize("mt", ["Cnn_9layers_FrameAvg", "Cnn_9layers_Gru_FrameAtt"])
def test_twenty_step_loss_trajectory_vs_oracle(mt):
    """Training dynamics, not just one step: 20 optimiser steps on one fixed batch of 8 waveforms (fresh mixup lambdas and
    SpecAugment stripes every step) on the HIP path and on the CPU oracle (main.py:281-319 loop body, Adam amsgrad).
    Adam's first updates are +-lr * sign(gradient) on EVERY entry, so entries whose gradient is rounding noise walk
    differently in any two fp32 evaluations, and on this tiny batch the curves separate after a few steps.  Yardstick,
    measured with the oracle alone on this recipe (FrameAvg): fp32 with 8 threads vs fp32 with 1 thread differ by up to
    9.9e-3 in loss (2e-4 within the first three steps, 8e-4 within the first six), fp32 vs float64 by 1.7e-2 .. 2.7e-2.
    Gate: step 0 within 2e-5, the first three steps within 5e-4, the first six within 2e-3, every step within 3e-2
    (measured 4e-3 .. 1.2e-2) and the mean difference within 1e-2; both runs must have
    learnt the batch; the BatchNorm running statistics are compared against the oracle's own spread (below)."""
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import get_loss_func
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    seed, rows, L, steps = SEEDS[mt], 8, 32000, 20
    T = L // 320 + 1
    xw = waves(2700 + seed, rows, L)
    tg = targets(2800 + seed, rows)
    rs = np.random.RandomState(4321)
    lams = [ofe.mixup_lambdas(rows, rs).astype(np.float32) for _ in range(steps)]
    torch.manual_seed(77 + seed)
    stripes = [ofe.draw_specaug_stripes(rows, T) for _ in range(steps)]

    m = build(mt)
    opt = FusedAdamAmsgrad(m, lr=1e-3, betas=(0.9, 0.999), eps=1e-08)
    loss_func = get_loss_func("clip_bce")
    xg, tgg = torch.from_numpy(xw).cuda(), torch.from_numpy(tg).cuda()
    ours = []
    m.train()
    for it in range(steps):
        lam = torch.from_numpy(lams[it]).cuda()
        o = m(xg, lam, specaug_stripes=stripes[it])
        loss = loss_func(o, {"target": do_mixup(tgg, lam)})
        opt.zero_grad()
        loss.backward()
        opt.step()
        ours.append(loss.item())

    st = om.recipe_state(mt, seed)
    unused = set(om.unused_keys(mt))
    keys = [k for k in om.trainable_keys(mt) if k not in unused]
    for k in keys:
        st[k].requires_grad_(True)
    state = {k: [torch.zeros_like(st[k]) for _ in range(3)] for k in keys}
    xc, tc = torch.from_numpy(xw), torch.from_numpy(tg)
    want = []
    for it in range(steps):
        lam = torch.from_numpy(lams[it])
        o = om.forward(mt, st, xc, training=True, mixup_lambda=lam, stripes=stripes[it])
        loss = om.clip_bce(o, {"target": om.do_mixup(tc, lam).clamp(max=1.0)})
        grads = torch.autograd.grad(loss, [st[k] for k in keys])
        with torch.no_grad():
            for k, g in zip(keys, grads):
                mm, v, vmax = state[k]
                om.adam_amsgrad_step(st[k], g, mm, v, vmax, it + 1, 1e-3)
        want.append(loss.item())

    diff = np.abs(np.array(ours) - np.array(want))
    print("loss curve ours  :", np.round(ours, 4).tolist())
    print("loss curve oracle:", np.round(want, 4).tolist())
    print("max |difference| %.2e at step %d" % (diff.max(), int(diff.argmax())))
    assert diff[0] < 2e-5 and diff[:3].max() < 5e-4 and diff[:6].max() < 2e-3, diff
    assert diff.max() < 3e-2 and diff.mean() < 1e-2, diff
    assert ours[-1] < 0.8 * ours[0] and want[-1] < 0.8 * want[0]
    # The BatchNorm running statistics integrate the whole trajectory.  The oracle's own spread on this recipe (8 threads
    # vs 1 thread vs float64): bn0.running_mean 4e-7, conv_block1.bn1.running_var 6e-4, conv_block4.bn2.running_var
    # 7e-2 .. 8e-2 (lr = 1e-3 is 4 % of a block-4 weight's scale PER STEP, so sign walks of noise-level entries show there).
    sd = m.state_dict()
    for k, gate in (("bn0.running_mean", 1e-5), ("conv_block1.bn1.running_var", 5e-3), ("conv_block4.bn2.running_var", 0.25)):
        a, b = sd[k].double().cpu().numpy(), st[k].double().numpy()
        err = np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum())
        print("%s relative L2 %.2e (gate %.0e)" % (k, err, gate))
        assert err < gate, (k, err)


def test_edge_inputs_fail_loudly_or_match_the_oracle():
    """Empty / too short / mis-sized inputs: the reference raises from torch (reflect padding needs L > n_fft/2, avg_pool2d
    refuses an empty output, do_mixup cannot broadcast); this path must raise too -- never read out of bounds or return
    garbage -- and the shortest clip the architecture admits (8 STFT frames -> one output frame) must still match the oracle."""
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
    mt = "Cnn_9layers_FrameAvg"
    m = build(mt).eval()
    for bad in (torch.zeros(0, 32000), torch.zeros(2, 1), torch.randn(2, 319), torch.randn(2, 512), torch.randn(2, 1000)):
        with pytest.raises((RuntimeError, ValueError)):
            with torch.no_grad():
                m(bad.cuda())
            torch.cuda.synchronize()
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 32000))                                        # host tensor: no CPU fallback
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 1, 32000).cuda())
    # shortest admissible clip: T = 8 frames
    x = waves(4242, 3, 7 * 320)
    with torch.no_grad():
        o = m(torch.from_numpy(x).cuda())
    want = om.forward(mt, om.recipe_state(mt, SEEDS[mt]), torch.from_numpy(x), training=False)
    assert o["framewise_output"].shape == tuple(want["framewise_output"].shape) == (3, 8, 17)
    assert np.abs(o["clipwise_output"].cpu().numpy() - want["clipwise_output"].numpy()).max() < 1e-4
    m.train()
    xw = torch.randn(4, 32000).cuda()
    with pytest.raises(ValueError):
        m(xw, torch.rand(2).cuda())                                     # one lambda per waveform
    with pytest.raises(ValueError):
        m(xw[:3], torch.rand(3).cuda())                                 # mixup pairs need an even batch
    with pytest.raises(ValueError):
        m(xw, torch.rand(4).cuda(), specaug_stripes=np.zeros((2, 8), np.int32))
    with pytest.raises(ValueError):
        do_mixup(torch.rand(4, 17).cuda(), torch.rand(3).cuda())
    torch.cuda.synchronize()
    from sound_event_detection_dcase2017_task4_amd import ops
    ops.check_device_errors(synchronize=True)

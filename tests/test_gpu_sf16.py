"""Split-f16 3x3 convolution (csrc/conv_sf16.hip) against a float64 convolution and against the fused epilogues of the
fp32 direct kernel: the three-term split must be as accurate as a direct fp32 convolution
(tools/split_f16_study.py: 1.5e-7 relative L2 per 1152-term sum)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x_nhwc, w, scale=None, shift=None):
    x = x_nhwc.double()
    if scale is not None:
        x = torch.relu(x * scale.double() + shift.double())
    y = F.conv2d(x.permute(0, 3, 1, 2), w.double(), padding=1)
    return y.permute(0, 2, 3, 1).contiguous()


def _err(y, want):
    d = y.double().cpu() - want
    return float(d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()), float(d.abs().max() / want.abs().max())


SHAPES = [(2, 37, 64, 64, 128, False), (3, 21, 32, 128, 128, True), (2, 250, 16, 128, 256, False),
          (3, 13, 8, 256, 512, True), (1, 1, 8, 16, 128, False), (2, 5, 64, 32, 256, True),
          (2, 9, 64, 64, 64, True), (2, 33, 32, 128, 64, False), (1, 7, 8, 64, 192, True), (2, 3, 16, 48, 64, False)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,inT", SHAPES)
def test_sf16_conv_matches_float64(B, H, W, Cin, Cout, inT):
    from sound_event_detection_dcase2017_task4_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn((B, H, W, Cin), generator=g) * 1.5
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * float(np.sqrt(6.0 / (9 * Cin + 9 * Cout)))
    st = scale = shift = None
    if inT:
        scale = torch.rand(Cin, generator=g) + 0.5
        shift = torch.randn(Cin, generator=g) * 0.3
        st = ops.BnStats(Cin, "cuda")
        st.scale.copy_(scale); st.shift.copy_(shift)
    want = _ref(x, w, scale, shift)
    y = ops.conv3x3_sf16(x.cuda(), ops.pack_sf16(w.cuda()), B, H, W, Cin, Cout, in_st=st)
    torch.cuda.synchronize()
    rel, mx = _err(y, want)
    print("sf16 %s: relative L2 %.2e, max %.2e of the output max" % ((B, H, W, Cin, Cout, inT), rel, mx))
    assert rel < 1e-6 and mx < 1e-5      # fp32 accumulation over K = 9*Cin terms: same as a direct fp32 convolution


@pytest.mark.parametrize("B,H,W,Cin,Cout,inT", [(2, 37, 64, 64, 128, True), (3, 11, 32, 128, 64, False), (2, 1001, 64, 64, 64, True),
                                               (2, 125, 8, 256, 512, False)])
def test_sf16_statistics_epilogue(B, H, W, Cin, Cout, inT):
    """Epilogue 1: the per-part (sum, M2, count) partials merge (sed_bn_finalize) to the batch statistics of the output."""
    from sound_event_detection_dcase2017_task4_amd import ops, _lib
    g = torch.Generator().manual_seed(77 + H)
    x = torch.randn((B, H, W, Cin), generator=g)
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * 0.05
    st = None
    if inT:
        st = ops.BnStats(Cin, "cuda")
        st.scale.copy_(torch.rand(Cin, generator=g) + 0.5); st.shift.copy_(torch.randn(Cin, generator=g) * 0.3)
    P = int(_lib.lib().sed_conv_sf16_num_parts(B, H, W, Cout))
    part = torch.full((P * 2 * Cout + P,), float("nan"), device="cuda")
    y = ops.conv3x3_sf16(x.cuda(), ops.pack_sf16(w.cuda()), B, H, W, Cin, Cout, in_st=st, epi=1, partials=part)
    gam, bet = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    bst = ops.bn_finalize(part, P, -1, B * H * W, gam, bet, rm, rv)
    torch.cuda.synchronize()
    assert float(part[P * 2 * Cout:].sum()) == B * H * W
    y2 = y.double().reshape(-1, Cout)
    mean, var = y2.mean(0), y2.var(0, unbiased=False)
    assert (bst.mean.double() - mean).abs().max() < 1e-5 * max(1.0, float(mean.abs().max()))
    assert ((bst.invstd.double() - 1.0 / torch.sqrt(var + 1e-5)).abs() / (1.0 / torch.sqrt(var + 1e-5))).max() < 1e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 37, 64, 128, 64), (3, 21, 32, 128, 128), (2, 1001, 64, 64, 64), (2, 13, 8, 512, 512)])
def test_sf16_dgrad_epilogue_and_dynamic_scale(B, H, W, Cin, Cout):
    """dgrad with epilogue 2 (ReLU mask of the previous activation + BatchNorm-backward sums) on gradients of magnitude 1e-6,
    scaled by the amax a producer left on the device: against float64."""
    from sound_event_detection_dcase2017_task4_amd import ops, _lib
    g = torch.Generator().manual_seed(91 + H)
    # w: (Cout, Cin): forward maps Cin -> Cout; this dgrad maps gy (Cout channels) -> gx (Cin channels)
    gy = torch.randn((B, H, W, Cout), generator=g) * 1e-6 * torch.exp(torch.randn((B, H, W, 1), generator=g))
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * 0.04
    yprev = torch.randn((B, H, W, Cin), generator=g)
    pst = ops.BnStats(Cin, "cuda")
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    mu, istd = torch.randn(Cin, generator=g) * 0.1, torch.rand(Cin, generator=g) + 0.5
    pst.scale.copy_(sc); pst.shift.copy_(sh); pst.mean.copy_(mu); pst.invstd.copy_(istd)
    full = F.conv_transpose2d(gy.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    mask = (torch.addcmul(sh, yprev, sc) > 0)            # fp32 fma, like bn_relu_active
    want = full * mask
    P = int(_lib.lib().sed_conv_sf16_num_parts(B, H, W, Cin))
    part = torch.full((P * 2 * Cin,), float("nan"), device="cuda")
    gyc = gy.cuda()
    gx = ops.conv3x3_sf16(gyc, ops.pack_sf16(w.cuda(), dgrad=True), B, H, W, Cout, Cin, epi=2, partials=part,
                          yprev=yprev.cuda(), p_st=pst, x_amax=ops.amax_of(gyc))
    torch.cuda.synchronize()
    rel, mx = _err(gx, want)
    print("sf16 dgrad %s: relative L2 %.2e, max %.2e" % ((B, H, W, Cin, Cout), rel, mx))
    flips = int(((gx.cpu() != 0) != (want != 0)).sum())
    assert flips <= 2, flips                               # addcmul vs fma may differ in the last bit at t == 0
    assert rel < 1e-6 and mx < 2e-5
    p = part.view(P, 2, Cin).double().sum(0).cpu()
    s1 = want.reshape(-1, Cin).sum(0)
    s2 = (want * ((yprev.double() - mu.double()) * istd.double())).reshape(-1, Cin).sum(0)
    assert (p[0] - s1).abs().max() <= 2e-5 * want.abs().reshape(-1, Cin).sum(0).max()
    assert (p[1] - s2).abs().max() <= 2e-5 * (want.abs() * ((yprev.double() - mu.double()) * istd.double()).abs()).reshape(-1, Cin).sum(0).max()


@pytest.mark.parametrize("B,H,W,Cin,Cout,inT", [(2, 21, 64, 64, 64, True), (3, 37, 32, 64, 128, False), (2, 50, 16, 128, 256, True),
                                               (5, 13, 8, 256, 512, False), (1, 1, 8, 32, 64, False), (3, 1001, 64, 64, 64, False),
                                               (9, 9, 16, 32, 64, True)])
def test_sf16_wgrad_matches_float64(B, H, W, Cin, Cout, inT):
    from sound_event_detection_dcase2017_task4_amd import ops
    g = torch.Generator().manual_seed(B * 77 + H)
    x = torch.randn((B, H, W, Cin), generator=g)
    gy = torch.randn((B, H, W, Cout), generator=g) * 1e-6 * torch.exp(torch.randn((B, H, W, 1), generator=g))
    st = None
    a = x.double()
    if inT:
        sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        st = ops.BnStats(Cin, "cuda")
        st.scale.copy_(sc); st.shift.copy_(sh)
        a = torch.relu(torch.addcmul(sh, x, sc)).double()
    want = torch.nn.grad.conv2d_weight(a.permute(0, 3, 1, 2), (Cout, Cin, 3, 3), gy.double().permute(0, 3, 1, 2), padding=1)
    dw = ops._wgrad_sf16(x.cuda(), gy.cuda(), B, H, W, Cin, Cout, in_st=st)
    torch.cuda.synchronize()
    rel, mx = _err(dw, want)
    print("sf16 wgrad %s: relative L2 %.2e, max %.2e" % ((B, H, W, Cin, Cout, inT), rel, mx))
    assert rel < 1e-6 and mx < 1e-5


def _chan_rel(y, want):
    """Worst over the output channels of rms(error) / rms(reference) of that channel -- NOT relative to the tensor max, so
    a channel (or a whole tensor) of small magnitude cannot hide behind a large one."""
    d = (y.double().cpu() - want).reshape(-1, want.shape[-1])
    w2 = want.reshape(-1, want.shape[-1])
    return float((d.pow(2).mean(0).sqrt() / w2.pow(2).mean(0).sqrt().clamp_min(1e-300)).max())


@pytest.mark.parametrize("inT", [False, True])
@pytest.mark.parametrize("mag", [1e-6, 1e-4, 1e-3, 1.0, 1e3, 1e6])
def test_sf16_is_magnitude_safe(mag, inT):
    """Operand scales come from device-side amax values: forward, dgrad and weight gradient keep the accuracy of a direct
    fp32 convolution at ANY activation magnitude (a fixed scale lost the low half to f16 subnormals below ~1e-3 -- 1e-5
    relative at 1e-4 -- and overflowed above 4094).  Error per output channel relative to that channel's RMS."""
    from sound_event_detection_dcase2017_task4_amd import ops
    B, H, W, Cin, Cout = 2, 19, 16, 128, 128
    g = torch.Generator().manual_seed(int(abs(np.log10(mag)) * 10) + inT)
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * float(np.sqrt(6.0 / (9 * Cin + 9 * Cout)))
    if inT:                                                       # operand relu(scale*y + shift) of magnitude `mag`
        x = torch.randn((B, H, W, Cin), generator=g)
        scale, shift = (torch.rand(Cin, generator=g) + 0.5) * mag, torch.randn(Cin, generator=g) * 0.3 * mag
        st = ops.BnStats(Cin, "cuda")
        st.scale.copy_(scale); st.shift.copy_(shift)
        a = torch.relu(torch.addcmul(shift, x, scale))
    else:
        x = torch.relu(torch.randn((B, H, W, Cin), generator=g)) * mag
        scale = shift = st = None
        a = x
    want = _ref(a, w)
    y = ops.conv3x3_sf16(x.cuda(), ops.pack_sf16(w.cuda()), B, H, W, Cin, Cout, in_st=st)
    gy = torch.randn((B, H, W, Cout), generator=g) * mag * 1e-3
    want_dw = torch.nn.grad.conv2d_weight(a.double().permute(0, 3, 1, 2), (Cout, Cin, 3, 3), gy.double().permute(0, 3, 1, 2),
                                          padding=1)
    dw = ops._wgrad_sf16(x.cuda(), gy.cuda(), B, H, W, Cin, Cout, in_st=st)
    ops.check_device_errors(synchronize=True)
    e_y, e_ych = _err(y, want)[0], _chan_rel(y, want)
    e_dw = _err(dw, want_dw)[0]
    print("sf16 at magnitude %g (fused affine %s): forward %.2e relative L2 (worst channel %.2e), wgrad %.2e"
          % (mag, inT, e_y, e_ych, e_dw))
    assert e_y < 5e-7 and e_ych < 1e-6 and e_dw < 5e-7       # the same at every magnitude: 4.8e-7 .. 5.6e-7 worst channel


def test_sf16_hot_channel():
    """One input channel 1000x the others (and, separately, one output channel's weights 1000x): the per-tensor scales
    follow the hot channel, the others still sit inside the 2^16 window in which hi + lo carries 22 bits."""
    from sound_event_detection_dcase2017_task4_amd import ops
    B, H, W, Cin, Cout = 2, 23, 32, 64, 128
    g = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn((B, H, W, Cin), generator=g))
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * 0.05
    x[..., 7] *= 1000.0
    w[11] *= 1000.0
    x_cold = x.clone(); x_cold[..., 7] = 0                        # what the other 63 channels contribute on their own
    y = ops.conv3x3_sf16(x.cuda(), ops.pack_sf16(w.cuda()), B, H, W, Cin, Cout)
    y_cold = ops.conv3x3_sf16(x_cold.cuda(), ops.pack_sf16(w.cuda()), B, H, W, Cin, Cout)
    ops.check_device_errors(synchronize=True)
    e, e_cold = _chan_rel(y, _ref(x, w)), _chan_rel(y_cold, _ref(x_cold, w))
    print("sf16 hot channel: %.2e per-channel, cold channels alone %.2e" % (e, e_cold))
    assert e < 5e-7 and e_cold < 5e-7


def test_act_amax_from_range_partials_is_exact():
    """The amax of the never-materialised operand relu(bn1(y1)) comes from the per-part per-channel (max, min) the producing
    convolution leaves: equal BIT FOR BIT to a pass over the tensor (the affine + ReLU is monotone), for the split-f16
    kernel's epilogue and for the direct Cin = 1 kernel."""
    from sound_event_detection_dcase2017_task4_amd import ops, _lib
    g = torch.Generator().manual_seed(3)
    B, H, W, Cin, Cout = 3, 37, 32, 64, 128
    x = torch.randn((B, H, W, Cin), generator=g).cuda()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * 0.05).cuda()
    st = ops.BnStats(Cout, "cuda")
    st.scale.copy_(torch.randn(Cout, generator=g)); st.shift.copy_(torch.randn(Cout, generator=g) * 0.5)   # both signs
    P = int(_lib.lib().sed_conv_sf16_num_parts(B, H, W, Cout))
    mm = torch.full((P, 2, Cout), float("nan"), device="cuda")
    y = ops.conv3x3_sf16(x, ops.pack_sf16(w), B, H, W, Cin, Cout, minmax=mm)
    a, b, c = ops.act_amax(mm, P, Cout, st), ops.act_amax_full(y, st), ops.act_amax(mm, P, Cout)
    torch.cuda.synchronize()
    a, b, c = ops.amax_value(a), ops.amax_value(b), ops.amax_value(c)          # device amax vectors: max over the 64 slots
    assert a == b > 0 and c == float(y.abs().max())
    ref = torch.relu(y.double() * st.scale.double() + st.shift.double()).max()
    assert abs(a - float(ref)) <= 1e-6 * float(ref)
    # direct kernel of block 1 (Cin = 1)
    B, H, W = 2, 101, 64
    x0 = torch.randn((B, H, W, 1), generator=g).cuda()
    w1 = (torch.randn((64, 1, 3, 3), generator=g) * 0.3).cuda()
    rpp = _lib.lib().sed_conv1_rows_per_part()
    n = (B * H * W + rpp - 1) // rpp
    mm1 = torch.full((n, 2, 64), float("nan"), device="cuda")
    y1 = torch.empty((B, H, W, 64), device="cuda")
    ops._call("sed_conv1_fwd", ops._ptr(x0), ops._ptr(w1), ops._ptr(y1), B, H, W, None, ops._ptr(mm1), ops._stream())
    st1 = ops.BnStats(64, "cuda")
    st1.scale.copy_(torch.randn(64, generator=g)); st1.shift.copy_(torch.randn(64, generator=g))
    assert ops.amax_value(ops.act_amax(mm1, n, 64, st1)) == ops.amax_value(ops.act_amax_full(y1, st1)) > 0


def test_sf16_nonfinite_operand_is_reported_and_adam_refuses_the_step():
    """Large FINITE operands are fine now (device-side amax scale).  A NaN / inf operand raises NonFiniteOperand at the
    next check and -- through the device word -- makes the Adam kernel leave parameters and moments untouched."""
    from sound_event_detection_dcase2017_task4_amd import ops
    B, H, W, Cin, Cout = 1, 4, 16, 32, 64
    x = torch.randn((B, H, W, Cin), device="cuda")
    w = torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05
    ops.check_device_errors(synchronize=True)
    x[0, 2, 3, 5] = 5000.0
    y = ops.conv3x3_sf16(x, ops.pack_sf16(w), B, H, W, Cin, Cout)
    ops.check_device_errors(synchronize=True)                          # finite: no error at any magnitude ...
    assert _err(y, _ref(x.cpu(), w.cpu()))[0] < 1e-6                   # ... and accurate
    p = torch.randn(1000, device="cuda"); p0 = p.clone()
    gr, m, v, vm = torch.randn(1000, device="cuda"), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    for bad, fn in ((float("inf"), "conv"), (float("nan"), "wgrad")):
        x[0, 2, 3, 5] = bad
        if fn == "conv":
            ops.conv3x3_sf16(x, ops.pack_sf16(w), B, H, W, Cin, Cout)
        else:
            ops._wgrad_sf16(x, torch.randn((B, H, W, Cout), device="cuda"), B, H, W, Cin, Cout)
        ops.adam_amsgrad_(p, gr, m, v, vm, 1, 1e-3, guard=True)        # refused: the device word is set
        ops.adam_amsgrad_(p, gr, m, v, vm, 1, 1e-3, guard=True)
        with pytest.raises(ops.NonFiniteOperand) as ei:
            ops.check_device_errors(synchronize=True)
        assert ei.value.skipped_steps == 2
        assert torch.equal(p, p0) and float(m.abs().max()) == 0.0
        ops.check_device_errors(synchronize=True)                      # reported once, then clear
    # a non-finite GRADIENT (e.g. arriving through the all-reduce from another rank) is refused as well
    gbad = gr.clone(); gbad[777] = float("nan")
    ops.adam_amsgrad_(p, gbad, m, v, vm, 1, 1e-3, guard=True)
    with pytest.raises(ops.NonFiniteOperand):
        ops.check_device_errors(synchronize=True)
    assert torch.equal(p, p0)
    ops.adam_amsgrad_(p, gr, m, v, vm, 1, 1e-3, guard=True)            # and a clean step goes through again
    torch.cuda.synchronize()
    assert not torch.equal(p, p0)
    ops.adam_amsgrad_(p, gbad, m, v, vm, 2, 1e-3, guard=False)         # guard off = torch.optim.Adam: NaN in, NaN out
    torch.cuda.synchronize()
    assert torch.isnan(p[777]) and not torch.isnan(p[0])


def test_multi_tensor_pack_equals_the_single_packs():
    """sed_pack_conv_weights_sf16_multi (all conv weights of a model in two launches) writes bit-identical operands, scales
    and amax slots to one sed_pack_conv_weights_sf16 call per weight, and leaves cache entries sf16_packs() finds."""
    from sound_event_detection_dcase2017_task4_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(128, 64), (256, 128), (512, 256), (64, 64), (128, 128), (256, 256), (512, 512)]
    ws = [torch.randn((co, ci, 3, 3), device="cuda", generator=g) * (10.0 ** (i - 3)) for i, (co, ci) in enumerate(shapes)]
    ops.invalidate_weight_caches()
    assert ops.prepack_sf16(ws, True) == 7
    assert ops.prepack_sf16(ws, True) == 0                       # fresh: nothing to do
    for w in ws:
        (f_m, s_m), (d_m, _) = ops.sf16_packs(w, True)           # cache hit: the multi-tensor result
        (f_1, s_1), (d_1, _) = ops.pack_sf16(w, both=True)
        assert torch.equal(f_m, f_1) and torch.equal(d_m, d_1)
        assert float(s_m[64]) == float(s_1[64]) and float(s_m[:64].max()) == float(s_1[:64].max()) == float(w.abs().max())
    # forward-only packs (inference) and a stale entry after the parameters moved
    ops.invalidate_weight_caches()
    assert ops.prepack_sf16(ws[:2], False) == 2
    f_m, d_m = ops.sf16_packs(ws[0], False)
    assert d_m is None and torch.equal(f_m[0], ops.pack_sf16(ws[0])[0])


@pytest.mark.parametrize("B,H,W,C,ph,pw", [(2, 37, 64, 64, 2, 2), (3, 21, 32, 128, 2, 2), (2, 50, 16, 256, 2, 2), (3, 13, 8, 512, 1, 8),
                                          (1, 1001, 64, 64, 2, 2), (2, 125, 8, 512, 1, 8), (1, 2, 16, 64, 2, 2)])
def test_sf16_eval_pool_epilogue(B, H, W, C, ph, pw):
    """Inference epilogue: avg_pool(relu(bn2(conv(relu(bn1(y1)))))) in one kernel equals the conv + pool kernels of the training
    path and a float64 evaluation; the amax it leaves is the maximum of the pooled tensor."""
    from sound_event_detection_dcase2017_task4_amd import ops, _lib
    g = torch.Generator().manual_seed(5 + H)
    y1 = torch.randn((B, H, W, C), generator=g)
    w = (torch.rand((C, C, 3, 3), generator=g) * 2 - 1) * 0.05
    st1, st2 = ops.BnStats(C, "cuda"), ops.BnStats(C, "cuda")
    s1, h1 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    s2, h2 = torch.rand(C, generator=g) * 2 - 1, torch.randn(C, generator=g) * 0.3          # both signs
    st1.scale.copy_(s1); st1.shift.copy_(h1); st2.scale.copy_(s2); st2.shift.copy_(h2)
    assert _lib.lib().sed_conv3x3_sf16_eval_pool_supported(H, W, C, C, ph, pw) == 1
    y1c = y1.cuda()
    pk = ops.pack_sf16(w.cuda())
    a1 = ops.act_amax_full(y1c, st1)
    out = torch.full((B, H // ph, W // pw, C), float("nan"), device="cuda")
    oa = ops._amax_buf("cuda")
    ops._call("sed_conv3x3_sf16_eval_pool", ops._ptr(y1c), ops._ptr(pk[0]), ops._ptr(pk[1]), ops._ptr(out), B, H, W, C, C,
              ops._ptr(st1.scale), ops._ptr(st1.shift), ops._ptr(st2.scale), ops._ptr(st2.shift), ph, pw, ops._ptr(a1), ops._ptr(oa),
              None, None, ops._stream())
    # the two-kernel path
    y2 = ops.conv3x3_sf16(y1c, pk, B, H, W, C, C, in_st=st1, x_amax=a1)
    ref2 = torch.empty_like(out)
    oa2 = ops._amax_buf("cuda")
    ops._call("sed_bn_relu_pool_fwd", ops._ptr(y2), B, H, W, C, ph, pw, ops._ptr(st2.scale), ops._ptr(st2.shift), ops._ptr(ref2),
              ops._ptr(oa2), ops._stream())
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert float((out - ref2).abs().max()) <= 1e-6 * max(1.0, float(ref2.abs().max()))
    assert ops.amax_value(oa) == float(out.max()) and abs(ops.amax_value(oa) - ops.amax_value(oa2)) <= 1e-6 * ops.amax_value(oa2)
    # float64
    a = torch.relu(y1.double() * s1.double() + h1.double()).permute(0, 3, 1, 2)
    z = F.conv2d(a, w.double(), padding=1)
    z = torch.relu(z * s2.double()[None, :, None, None] + h2.double()[None, :, None, None])
    want = F.avg_pool2d(z, kernel_size=(ph, pw)).permute(0, 2, 3, 1)
    assert float((out.cpu().double() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("B,H", [(3, 21), (2, 1001), (5, 7)])
def test_block1_without_a_materialised_conv1_output_is_bit_identical(B, H):
    """Round 4: in training, block 1 never writes the raw conv1 output y1 (4.2 GB at batch 256).  A statistics pass feeds bn1,
    sed_conv1_act_sf16 writes relu(bn1(y1)) once as split-f16 operand pairs (plain-copy staging in conv2's forward and weight
    gradient), and the two backward kernels that need the RAW y1 recompute it from the one-channel input with conv1's own
    fma sequence.  Every MFMA operand, ReLU mask and BatchNorm sum is therefore the same number as in the round-3 dataflow
    (ops.B1_ACT_PAIRS = False): outputs, amax, all gradients and the running statistics must be EQUAL, bit for bit --
    including ragged image bottoms (H = 21: tiles of 4 rows; H = 7: one odd tile) and the full 10 s frame count."""
    from sound_event_detection_dcase2017_task4_amd import ops
    W, Cout = 64, 64
    g = torch.Generator().manual_seed(100 + H)
    x = torch.randn(B, H, W, 1, generator=g).cuda()
    base = [(torch.randn(Cout, 1, 3, 3, generator=g) * 0.5), 1 + 0.2 * torch.randn(Cout, generator=g), 0.2 * torch.randn(Cout, generator=g),
            0.1 * torch.randn(Cout, generator=g), 0.5 + torch.rand(Cout, generator=g),
            (torch.randn(Cout, Cout, 3, 3, generator=g) * (1.5 / np.sqrt(9 * Cout))), 1 + 0.2 * torch.randn(Cout, generator=g),
            0.2 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g), 0.5 + torch.rand(Cout, generator=g)]
    gout = torch.randn(B, H // 2, W // 2, Cout, generator=g).cuda()
    results = []
    prev = (ops.B1_ACT_PAIRS, ops.USE_SF16)
    try:
        ops.USE_SF16 = True
        for flag in (False, True):
            ops.B1_ACT_PAIRS = flag
            params = [t.clone().cuda() for t in base]
            for i in (0, 1, 2, 5, 6, 7):
                params[i].requires_grad_(True)
            xg = x.clone().requires_grad_(True)
            out, out_amax = ops.ConvBlockFn.apply(xg, *params, True, 2, 2)
            out.backward(gout)
            torch.cuda.synchronize()
            ops.check_device_errors()
            results.append([out.detach(), out_amax, xg.grad] + [params[i].grad for i in (0, 1, 2, 5, 6, 7)] + [params[i] for i in (3, 4, 8, 9)])
    finally:
        ops.B1_ACT_PAIRS, ops.USE_SF16 = prev
    names = ["out", "out_amax", "dx", "dw1", "dgamma1", "dbeta1", "dw2", "dgamma2", "dbeta2", "rm1", "rv1", "rm2", "rv2"]
    for n, a, b in zip(names, results[0], results[1]):
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0, n
        assert torch.equal(a, b), (n, float((a - b).abs().max()))


def test_conv1_act_pairs_decode_to_relu_bn_conv1():
    """sed_conv1_act_sf16: hi + lo of every pair, unscaled by the power of two of the amax, is relu(scale * conv1(x0) + shift)
    to 2^-22 of the amax; a NaN input raises the found-non-finite words (the ReLU's fmaxf would swallow it)."""
    import ctypes
    from sound_event_detection_dcase2017_task4_amd import ops
    B, H, W = 2, 13, 64
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, H, W, 1, generator=g).cuda()
    w = (torch.randn(64, 1, 3, 3, generator=g) * 0.5).cuda()
    sc, sh = (1 + 0.3 * torch.randn(64, generator=g)).cuda(), (0.3 * torch.randn(64, generator=g)).cuda()
    y = torch.empty((B, H, W, 64), device="cuda")
    ops._call("sed_conv1_fwd", ops._ptr(x), ops._ptr(w), ops._ptr(y), B, H, W, None, None, ops._stream())
    ref = torch.relu(y * sc + sh)
    amax = ops.amax_of(ref)
    pairs = torch.empty((B, H, W, 64), dtype=torch.int32, device="cuda")
    ops.check_device_errors(synchronize=True)
    ops._call("sed_conv1_act_sf16", ops._ptr(x), ops._ptr(w), B, H, W, ops._ptr(sc), ops._ptr(sh), ops._ptr(amax), ops._ptr(pairs),
              ops._sf16_err_ptr(), ops._sf16_err_dev_ptr(x.device), ops._stream())
    halves = pairs.view(torch.float16).view(B, H, W, 32, 2, 2).float()          # [.., channel pair, {hi, lo} dword, channel of the pair]
    s = 2.0 ** (14 - int(np.ceil(np.log2(float(amax.max()) * (1 + 1e-7)))))
    dec = (halves[..., 0, :] + halves[..., 1, :]).reshape(B, H, W, 64) / s
    assert float((dec - ref).abs().max()) <= float(amax.max()) * 2.0 ** -21
    ops.check_device_errors(synchronize=True)
    xb = x.clone(); xb[1, 5, 9, 0] = float("nan")
    ops._call("sed_conv1_act_sf16", ops._ptr(xb), ops._ptr(w), B, H, W, ops._ptr(sc), ops._ptr(sh), ops._ptr(amax), ops._ptr(pairs),
              ops._sf16_err_ptr(), ops._sf16_err_dev_ptr(x.device), ops._stream())
    with pytest.raises(ops.NonFiniteOperand):
        ops.check_device_errors(synchronize=True)


@pytest.mark.parametrize("Cin,Cout,H,W,ph,pw", [(64, 128, 22, 32, 2, 2), (128, 256, 20, 16, 2, 2), (256, 512, 12, 8, 1, 8), (1, 64, 21, 64, 2, 2)])
def test_gradients_as_operand_pairs_match_the_fp32_tensors(Cin, Cout, H, W, ph, pw):
    """ops.GRAD_PAIRS: the BatchNorm-backward apply kernels write gy2 / gy1 as split-f16 pairs scaled by a device-side BOUND
    of their amax (sed_grad_bound); dgrad and weight-gradient kernels copy them into LDS.  hi and lo are floating-point
    halves, so a scale that is a few powers of two lower than the exact-amax scale changes nothing but the exponent:
    every gradient must agree with the fp32-tensor dataflow to rounding noise -- at gradient magnitudes 1e-6 .. 1e+3 --
    and the bound must really bound (no overflow, nothing non-finite)."""
    from sound_event_detection_dcase2017_task4_amd import ops
    B = 3
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).abs_().cuda() if Cin != 1 else torch.randn(B, H, W, 1, generator=g).cuda()
    base = [(torch.randn(Cout, Cin, 3, 3, generator=g) * (1.5 / np.sqrt(9 * Cin))), 1 + 0.2 * torch.randn(Cout, generator=g),
            0.2 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g), 0.5 + torch.rand(Cout, generator=g),
            (torch.randn(Cout, Cout, 3, 3, generator=g) * (1.5 / np.sqrt(9 * Cout))), 1 + 0.2 * torch.randn(Cout, generator=g),
            0.2 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g), 0.5 + torch.rand(Cout, generator=g)]
    g0 = torch.randn(B, H // ph, W // pw, Cout, generator=g).cuda()
    prev = (ops.GRAD_PAIRS, ops.USE_SF16)
    try:
        ops.USE_SF16 = True
        for mag in (1.0, 1e-6, 1e3):
            res = []
            for flag in (False, True):
                ops.GRAD_PAIRS = flag
                params = [t.clone().cuda() for t in base]
                for i in (0, 1, 2, 5, 6, 7):
                    params[i].requires_grad_(True)
                xg = x.clone().requires_grad_(True)
                out, _ = ops.ConvBlockFn.apply(xg, *params, True, ph, pw)
                out.backward(g0 * mag)
                torch.cuda.synchronize()
                ops.check_device_errors()
                res.append([xg.grad] + [params[i].grad for i in (0, 1, 2, 5, 6, 7)])
            for n, a, b in zip(["dx", "dw1", "dgamma1", "dbeta1", "dw2", "dgamma2", "dbeta2"], res[0], res[1]):
                assert torch.isfinite(b).all(), (n, mag)
                err = float((a - b).double().norm() / max(float(a.double().norm()), 1e-300))
                assert err < 2e-6, (n, mag, err)
    finally:
        ops.GRAD_PAIRS, ops.USE_SF16 = prev


def test_grad_bound_bounds_and_is_tight_enough():
    """sed_grad_bound >= max |a*dy + b*y + c| over any dy with |dy| <= G * ginv and y inside its per-channel range, and within
    a small factor of the attained maximum for typical coefficients (it only picks a power-of-two scale)."""
    from sound_event_detection_dcase2017_task4_amd import ops
    C, N, nparts = 128, 4096, 16
    g = torch.Generator().manual_seed(3)
    y = (torch.randn(N, C, generator=g) * (0.5 + torch.rand(C, generator=g)) + torch.randn(C, generator=g)).cuda()
    dy = (torch.randn(N, C, generator=g) * 3e-3).cuda()
    coef = torch.stack([1 + 0.3 * torch.randn(C, generator=g), 1e-3 * torch.randn(C, generator=g), 1e-3 * torch.randn(C, generator=g)]).cuda()
    yp = y.view(nparts, N // nparts, C)
    mm = torch.stack([yp.max(1)[0], yp.min(1)[0]], dim=1).contiguous()           # [nparts][2][C]
    G = ops.amax_of(dy)
    out = ops._amax_buf(y.device)
    ops._call("sed_grad_bound", ops._ptr(mm), nparts, C, ops._ptr(coef), ops._ptr(G), 1.0, ops._ptr(out), None, ops._stream())
    bound = float(out.max())
    true = float((coef[0] * dy + coef[1] * y + coef[2]).abs().max())
    assert true <= bound <= 4.0 * true, (true, bound)
    out2 = ops._amax_buf(y.device)                      # the looser form: |y| <= amax(y) for every channel
    ops._call("sed_grad_bound", None, 0, C, ops._ptr(coef), ops._ptr(G), 1.0, ops._ptr(out2), ops._ptr(ops.amax_of(y)), ops._stream())
    assert bound <= float(out2.max()) <= 8.0 * true


@pytest.mark.parametrize("mt", ["Cnn_9layers_FrameAvg", "Cnn_9layers_Gru_FrameAtt"])
def test_operand_pair_dataflows_leave_a_training_step_unchanged(mt):
    """Whole model, one training step: with gradients and pooled block outputs written as operand pairs (ops.GRAD_PAIRS,
    ops.ACT_PAIRS) -- and with block 1's activation pairs (ops.B1_ACT_PAIRS) -- loss, outputs and every gradient equal the
    fp32-tensor dataflow's to rounding noise (a pair differs from the fp32 value it replaces by 2^-22 of the tensor's amax
    bound at most, and only the power-of-two scale depends on how tight that bound is)."""
    from oracle import model as om
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch import models
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    rs = np.random.RandomState(5)
    x = torch.from_numpy((rs.randn(8, 64000) * 0.1).astype(np.float32)).cuda()
    y = torch.from_numpy((rs.rand(8, 17) < 0.2).astype(np.float32)).cuda()
    stripes = torch.zeros((8, 8), dtype=torch.int32)
    prev = (ops.GRAD_PAIRS, ops.ACT_PAIRS, ops.B1_ACT_PAIRS, ops.USE_SF16)
    res = []
    try:
        ops.USE_SF16 = True
        for gp, ap, b1 in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
            ops.GRAD_PAIRS, ops.ACT_PAIRS, ops.B1_ACT_PAIRS = gp, ap, b1
            m = getattr(models, mt)(32000, 1024, 320, 64, 50, 14000, 17)
            m.load_state_dict(om.recipe_state(mt, 4))
            m = m.cuda().train()
            opt = FusedAdamAmsgrad(m, lr=1e-3)
            out = m(x, None, specaug_stripes=stripes)
            loss = clip_bce(out, {"target": y})
            opt.zero_grad(); loss.backward()
            torch.cuda.synchronize()
            ops.check_device_errors()
            res.append((float(loss), out["clipwise_output"].detach().clone(), opt.flat_grad.clone(), opt.offsets, [n for n, p in m.named_parameters() if p.requires_grad],
                        [p.numel() for p in m.parameters() if p.requires_grad]))
    finally:
        ops.GRAD_PAIRS, ops.ACT_PAIRS, ops.B1_ACT_PAIRS, ops.USE_SF16 = prev
    ref = res[0]
    for k, r in enumerate(res[1:], start=1):
        assert abs(r[0] - ref[0]) < 2e-6 and float((r[1] - ref[1]).abs().max()) < 2e-6, k
        for name, off, n in zip(ref[4], ref[3], ref[5]):
            a, b = ref[2][off:off + n].double(), r[2][off:off + n].double()
            if float(a.norm()) < 1e-6:          # structurally zero gradients (attention shift invariance): rounding noise only
                assert float((a - b).norm()) < 1e-6, (k, name)
                continue
            assert float((a - b).norm() / a.norm()) < 5e-5, (k, name, float((a - b).norm() / a.norm()))


@pytest.mark.parametrize("M,N,K,mag", [(1000, 128, 32, 1.0), (4099, 1536, 512, 1.0), (32000, 512, 1536, 1e-6), (777, 256, 96, 1e5)])
def test_split_f16_gemm_vs_float64(M, N, K, mag):
    """sed_gemm_nt_sf16 (csrc/gemm_sf16.hip: the GRU input projection and its input gradient) against a float64 matmul: ragged M,
    one and many K stages, operand magnitudes 1e-6 ... 1e+5 (scales from device-side amax values), bias; relative L2 <= 1e-6 and
    the worst output column within 2e-6 of its RMS -- the gate of the split-f16 convolutions.  The fp32-MFMA sed_gemm_nt measures
    3e-7 on the same inputs."""
    from sound_event_detection_dcase2017_task4_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * mag).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05 * (1 + 10 * (torch.rand(N, 1, generator=g) < 0.02).float())).cuda()   # a few hot rows
    bias = (torch.randn(N, generator=g) * mag).cuda()
    assert ops.gemm_nt_sf16_ok(M, N, K)
    pack = ops.gemm_pack_sf16(w)
    out_amax = ops._amax_buf(x.device)
    y = ops.gemm_nt_sf16(x, pack, N, bias, out_amax=out_amax)
    ops.check_device_errors(synchronize=True)
    ref = x.double() @ w.double().t() + bias.double()
    err = (y.double() - ref)
    assert float(err.norm() / ref.norm()) < 1e-6
    col = err.pow(2).mean(0).sqrt() / ref.pow(2).mean(0).sqrt().clamp_min(1e-300)
    assert float(col.max()) < 2e-6, float(col.max())
    assert abs(float(out_amax.max()) - float(y.abs().max())) <= 1e-6 * float(y.abs().max())
    y32 = ops.gemm_nt(x, w, bias) if (N % 64 == 0 and K % 32 == 0) else None
    if y32 is not None:
        assert float((y - y32).double().norm() / ref.norm()) < 2e-6


def test_split_f16_gemm_reports_a_non_finite_operand():
    from sound_event_detection_dcase2017_task4_amd import ops
    x = torch.randn(512, 64).cuda()
    x[17, 3] = float("nan")
    w = torch.randn(128, 64).cuda()
    ops.check_device_errors(synchronize=True)
    ops.gemm_nt_sf16(x, ops.gemm_pack_sf16(w), 128)
    with pytest.raises(ops.NonFiniteOperand):
        ops.check_device_errors(synchronize=True)
    ops.check_device_errors(synchronize=True)


@pytest.mark.parametrize("M,N,K,mag", [(64, 128, 128, 1.0), (1000, 128, 256, 1.0), (31744, 768, 256, 1e-6), (32000, 1536, 512, 1e4),
                                       (4099, 512, 512, 1.0)])
def test_split_f16_tn_gemm_vs_float64(M, N, K, mag):
    """sed_gemm_tn_sf16 (the weight gradients of the GRU / MultiHead dense layers: dw[n][k] = sum_m gy[m][n] x[m][k], both operands
    converted when staged, LDS transpose reads, row slices reduced in fp64) against float64: ragged M (a partial last stage, fewer
    stages than slices), gradient magnitudes 1e-6 ... 1e+4, one operand given by an upper BOUND of its amax; relative L2 <= 1e-6,
    worst output row within 3e-6 of its RMS."""
    from sound_event_detection_dcase2017_task4_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.tanh(torch.randn(M, K, generator=g)).cuda()                      # |x| <= 1 (hidden states)
    gy = (torch.randn(M, N, generator=g) * mag * (1 + 5 * (torch.rand(1, N, generator=g) < 0.03).float())).cuda()
    dw = ops.gemm_tn(x, gy, x_amax=ops.unit_amax("cuda"))                      # a bound of amax(x), amax(gy) by one pass
    ops.check_device_errors(synchronize=True)
    ref = gy.double().t() @ x.double()
    err = dw.double() - ref
    assert float(err.norm() / ref.norm()) < 1e-6
    row = err.pow(2).mean(1).sqrt() / ref.pow(2).mean(1).sqrt().clamp_min(1e-300)
    assert float(row.max()) < 3e-6, float(row.max())


def test_gradient_amax_hand_over_never_outlives_its_backward_pass():
    """A ConvBlock's backward leaves the amax of the input gradient it returns for the PREVIOUS block's backward (ops._GRAD_AMAX,
    keyed by the gradient's address).  A standalone block whose gradient nobody consumes must not leave that entry behind: the
    allocator reuses addresses, and a later gradient at the same address would be scaled by a dead tensor's amax."""
    from sound_event_detection_dcase2017_task4_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 2, 8, 16, 64, 128
    x = torch.randn(B, H, W, Cin, generator=g).cuda().requires_grad_(True)
    ps = [(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05), torch.ones(Cout), torch.zeros(Cout), torch.zeros(Cout), torch.ones(Cout),
          (torch.randn(Cout, Cout, 3, 3, generator=g) * 0.05), torch.ones(Cout), torch.zeros(Cout), torch.zeros(Cout), torch.ones(Cout)]
    ps = [t.cuda() for t in ps]
    for i in (0, 1, 2, 5, 6, 7):
        ps[i].requires_grad_(True)
    out, _ = ops.ConvBlockFn.apply(x, *ps, True, 2, 2)
    out.backward(torch.randn(out.shape, generator=g).cuda())
    torch.cuda.synchronize()
    assert x.grad is not None and len(ops._GRAD_AMAX) == 0


def _random_shapes(n, seed):
    """Seeded sweep over what the split-f16 kernels accept: every width, ragged heights around the tile sizes (256 / W rows per
    tile; 64 / W rows per weight-gradient stage), odd batch sizes, every channel count of the model and a few others."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        W = int(rs.choice([8, 16, 32, 64]))
        tr = 256 // W
        H = int(rs.choice([1, 2, tr - 1, tr, tr + 1, 2 * tr + 3, 3 * tr - 2, int(rs.randint(1, 5 * tr))]))
        B = int(rs.randint(1, 6))
        Cin = int(rs.choice([32, 64, 96, 128, 256, 512]))
        Cout = int(rs.choice([64, 128, 192, 256, 512]))
        out.append((B, max(H, 1), W, Cin, Cout, bool(rs.randint(0, 2))))
    return out


@pytest.mark.parametrize("B,H,W,Cin,Cout,inT", _random_shapes(14, 20250930))
def test_sf16_random_shape_sweep_forward_dgrad_wgrad_vs_float64(B, H, W, Cin, Cout, inT):
    """Forward (with / without the fused operand transform), dgrad and weight gradient of ONE random shape against float64 --
    shapes nobody hand-picked (ragged last tiles, one-row images, tile-boundary heights, channel counts between the model's)."""
    from sound_event_detection_dcase2017_task4_amd import ops
    g = torch.Generator().manual_seed(B * 1009 + H * 31 + W + Cin + Cout)
    x = torch.randn((B, H, W, Cin), generator=g) * 1.3
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * float(np.sqrt(6.0 / (9 * Cin + 9 * Cout)))
    gy = torch.randn((B, H, W, Cout), generator=g) * 1e-5
    st = scale = shift = None
    if inT:
        scale, shift = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        st = ops.BnStats(Cin, "cuda")
        st.scale.copy_(scale); st.shift.copy_(shift)
    xd, gyd, wd = x.cuda(), gy.cuda(), w.cuda()
    # forward
    want = _ref(x, w, scale, shift)
    y = ops.conv3x3_sf16(xd, ops.pack_sf16(wd), B, H, W, Cin, Cout, in_st=st)
    rel, mx = _err(y, want)
    assert rel < 1e-6 and mx < 1e-5, ("forward", rel, mx)
    # dgrad (the transposed convolution; needs Cin % 64 == 0 as ITS output channel count and Cout % 16 == 0 as its K)
    if Cin % 64 == 0:
        xr = torch.zeros((B, Cin, H, W), dtype=torch.float64, requires_grad=True)
        F.conv2d(xr, w.double(), padding=1).backward(gy.double().permute(0, 3, 1, 2))
        gx = ops.conv3x3_sf16(gyd, ops.pack_sf16(wd, dgrad=True), B, H, W, Cout, Cin)
        rel, mx = _err(gx, xr.grad.permute(0, 2, 3, 1).contiguous())
        assert rel < 1e-6 and mx < 1e-5, ("dgrad", rel, mx)
    # weight gradient
    L = ops._lib.lib()
    if L.sed_wgrad_sf16_supported(H, W, Cin, Cout):
        a = x.double() if not inT else torch.relu(torch.addcmul(shift, x, scale)).double()
        wantw = torch.nn.grad.conv2d_weight(a.permute(0, 3, 1, 2), (Cout, Cin, 3, 3), gy.double().permute(0, 3, 1, 2), padding=1)
        dw = ops._wgrad_sf16(xd, gyd, B, H, W, Cin, Cout, in_st=st)
        rel, mx = _err(dw, wantw)
        assert rel < 1e-6 and mx < 1e-5, ("wgrad", rel, mx)
    torch.cuda.synchronize()
    ops.check_device_errors(synchronize=True)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 125, 8, 512, 512), (2, 125, 8, 256, 512), (1, 250, 16, 256, 256), (1, 37, 32, 256, 128),
                                            (3, 13, 8, 512, 256)])
def test_sf16_split_k_small_m_launches(B, H, W, Cin, Cout, monkeypatch):
    """Small-M launches (the reference's `--batch_size 32` over 8 GPUs leaves 4 clips per GPU: the 512-channel layers are 128
    workgroups of 96 dependent stages on 256 CUs) split their K range over `ksplit` workgroups per output tile; the last one to
    arrive adds the others' accumulators and runs the UNCHANGED epilogue.  Checked against float64 and against the un-split launch
    for every epilogue the training step uses: plain / statistics + range (with and without the fused operand transform), and the
    dgrad form with ReLU mask + BatchNorm-backward sums; the tickets are back at zero afterwards, and a second launch agrees with
    the first bit for bit (same partial sums, same order)."""
    from sound_event_detection_dcase2017_task4_amd import ops
    L = ops._lib.lib()
    ks = int(L.sed_conv_sf16_ksplit(B, H, W, Cin, Cout))
    assert ks > 1, "pick a shape the library splits"
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randn((B, H, W, Cin), generator=g)
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * float(np.sqrt(6.0 / (9 * Cin + 9 * Cout)))
    scale, shift = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    st = ops.BnStats(Cin, "cuda"); st.scale.copy_(scale); st.shift.copy_(shift)
    xd, wd = x.cuda(), w.cuda()
    pack = ops.pack_sf16(wd)
    P = int(L.sed_conv_sf16_num_parts(B, H, W, Cout))

    def run(split, in_st, epi):
        monkeypatch.setattr(ops, "CONV_SPLITK", split)
        parts = torch.zeros((P * 2 * Cout + P,), device="cuda") if epi else None
        mm = torch.zeros((P, 2, Cout), device="cuda")
        y = ops.conv3x3_sf16(xd, pack, B, H, W, Cin, Cout, in_st=in_st, epi=epi, partials=parts, minmax=mm)
        torch.cuda.synchronize()
        return y, parts, mm

    for in_st, sc, sh in ((None, None, None), (st, scale, shift)):
        want = _ref(x, w, sc, sh)
        for epi in (0, 1):
            y1, p1, m1 = run(True, in_st, epi)
            y0, p0, m0 = run(False, in_st, epi)
            rel, mx = _err(y1, want)
            assert rel < 1e-6 and mx < 1e-5, (in_st is not None, epi, rel, mx)
            assert (y1 - y0).abs().max().item() <= 2e-6 * want.abs().max().item()
            assert torch.allclose(m1, m0, rtol=1e-5, atol=2e-6 * float(want.abs().max()))
            if epi:
                n = P * Cout
                assert torch.allclose(p1[:n], p0[:n], rtol=1e-4, atol=1e-3) and torch.equal(p1[2 * n:], p0[2 * n:])     # sums, counts
                assert torch.allclose(p1[n:2 * n], p0[n:2 * n], rtol=1e-3, atol=1e-3)                                   # M2
            y2, _, _ = run(True, in_st, epi)
            assert torch.equal(y1, y2)
    # dgrad form: transposed convolution of a gradient-sized tensor, masked by relu'(bn(yprev)), with the BatchNorm-backward sums
    ks_d = int(L.sed_conv_sf16_ksplit(B, H, W, Cout, Cin))
    if ks_d > 1:
        gy = (torch.randn((B, H, W, Cout), generator=g) * 1e-5).cuda()
        yprev = torch.randn((B, H, W, Cin), generator=g).cuda()
        pst = ops.BnStats(Cin, "cuda"); pst.scale.copy_(scale); pst.shift.copy_(shift); pst.mean.fill_(0.1); pst.invstd.fill_(0.9)
        packd = ops.pack_sf16(wd, dgrad=True)
        Pd = int(L.sed_conv_sf16_num_parts(B, H, W, Cin))
        outs = []
        for split in (True, False):
            monkeypatch.setattr(ops, "CONV_SPLITK", split)
            parts = torch.zeros((Pd * 2 * Cin,), device="cuda")
            gx = ops.conv3x3_sf16(gy, packd, B, H, W, Cout, Cin, epi=2, partials=parts, yprev=yprev, p_st=pst)
            torch.cuda.synchronize()
            outs.append((gx, parts))
        (g1, q1), (g0, q0) = outs
        assert torch.equal(g1 == 0, g0 == 0)                                   # the same ReLU mask
        assert (g1 - g0).abs().max().item() <= 2e-6 * g0.abs().max().item()
        assert torch.allclose(q1, q0, rtol=1e-4, atol=1e-6 * float(q0.abs().max()))
    assert ops._TICKETS and all(int(t.abs().sum()) == 0 for t in ops._TICKETS.values())
    ops.check_device_errors(synchronize=True)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(100, 8, 8, 256, 512), (26, 125, 8, 512, 512), (13, 250, 16, 256, 256)])
def test_sf16_tail_split_fills_the_last_round(B, H, W, Cin, Cout, monkeypatch):
    """Round 6: a launch of more than one round of the chip's 768 resident workgroups (the 125 x 8 layers at the metric's batch: 1024
    workgroups = 1.33 rounds) can split the K range of the tiles of its last round only, so that their shares fill it; the full
    rounds in front run exactly as un-split.  The library's own rule never selects this form (measured slower:
    profiles/r06/tail_split_ab.txt) -- it stays reachable for A/B runs (SED_CONV_TAIL) and must stay correct: against float64,
    against the un-split launch (the un-split tiles bit for bit, the split ones to a rounding), every training epilogue, tickets
    back at zero, bit-reproducible."""
    import ctypes
    from sound_event_detection_dcase2017_task4_amd import ops
    L = ops._lib.lib()
    nfull = ctypes.c_int(0)
    assert int(L.sed_conv_sf16_split_plan(B, H, W, Cin, Cout, 0, ctypes.byref(nfull))) == 1        # the library's rule: un-split
    monkeypatch.setattr(ops, "CONV_TAIL", 3)
    ks = int(L.sed_conv_sf16_split_plan(B, H, W, Cin, Cout, 3, ctypes.byref(nfull)))
    tr = 256 // W
    tiles = B * ((H + tr - 1) // tr) * (Cout // 64)
    assert ks > 1 and nfull.value > 0 and nfull.value % 768 == 0 and nfull.value < tiles, (ks, nfull.value, tiles)
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randn((B, H, W, Cin), generator=g)
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * float(np.sqrt(6.0 / (9 * Cin + 9 * Cout)))
    scale, shift = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    st = ops.BnStats(Cin, "cuda"); st.scale.copy_(scale); st.shift.copy_(shift)
    xd, wd = x.cuda(), w.cuda()
    pack = ops.pack_sf16(wd)
    P = int(L.sed_conv_sf16_num_parts(B, H, W, Cout))

    def run(split, in_st, epi):
        monkeypatch.setattr(ops, "CONV_SPLITK", split)
        parts = torch.zeros((P * 2 * Cout + P,), device="cuda") if epi else None
        mm = torch.zeros((P, 2, Cout), device="cuda")
        y = ops.conv3x3_sf16(xd, pack, B, H, W, Cin, Cout, in_st=in_st, epi=epi, partials=parts, minmax=mm)
        torch.cuda.synchronize()
        return y, parts, mm

    for in_st, sc, sh in ((None, None, None), (st, scale, shift)):
        want = _ref(x, w, sc, sh)
        for epi in (0, 1):
            y1, p1, m1 = run(True, in_st, epi)
            y0, p0, m0 = run(False, in_st, epi)
            rel, mx = _err(y1, want)
            assert rel < 1e-6 and mx < 1e-5, (in_st is not None, epi, rel, mx)
            same = (y1 == y0).reshape(B, -1).all(dim=1)
            assert int(same.sum()) >= nfull.value // ((H + tr - 1) // tr * (Cout // 64))       # the images of the full rounds: identical bits
            assert not bool(same.all())                                                         # ... and the tail WAS summed differently
            assert (y1 - y0).abs().max().item() <= 2e-6 * want.abs().max().item()
            assert torch.allclose(m1, m0, rtol=1e-5, atol=2e-6 * float(want.abs().max()))
            if epi:
                n = P * Cout
                assert torch.allclose(p1[:n], p0[:n], rtol=1e-4, atol=1e-3) and torch.equal(p1[2 * n:], p0[2 * n:])
                assert torch.allclose(p1[n:2 * n], p0[n:2 * n], rtol=1e-3, atol=1e-3)
            y2, _, _ = run(True, in_st, epi)
            assert torch.equal(y1, y2)
    if Cin == Cout:        # dgrad form (conv2's backward: same channel count both ways)
        gy = (torch.randn((B, H, W, Cout), generator=g) * 1e-5).cuda()
        yprev = torch.randn((B, H, W, Cin), generator=g).cuda()
        pst = ops.BnStats(Cin, "cuda"); pst.scale.copy_(scale); pst.shift.copy_(shift); pst.mean.fill_(0.1); pst.invstd.fill_(0.9)
        packd = ops.pack_sf16(wd, dgrad=True)
        outs = []
        for split in (True, False, True):
            monkeypatch.setattr(ops, "CONV_SPLITK", split)
            parts = torch.zeros((P * 2 * Cin,), device="cuda")
            gx = ops.conv3x3_sf16(gy, packd, B, H, W, Cout, Cin, epi=2, partials=parts, yprev=yprev, p_st=pst)
            torch.cuda.synchronize()
            outs.append((gx, parts))
        (g1, q1), (g0, q0), (g2, q2) = outs
        assert torch.equal(g1 == 0, g0 == 0) and torch.equal(g1, g2) and torch.equal(q1, q2)
        assert (g1 - g0).abs().max().item() <= 2e-6 * g0.abs().max().item()
        assert torch.allclose(q1, q0, rtol=1e-4, atol=1e-6 * float(q0.abs().max()))
    assert ops._TICKETS and all(int(t.abs().sum()) == 0 for t in ops._TICKETS.values())
    ops.check_device_errors(synchronize=True)


def test_sf16_split_k_exchange_across_xcds():
    """The split-K exchange carries no release / acquire fence (an agent-scope fence pair costs ~10 us per workgroup on this 8-XCD
    part): shares travel with relaxed agent-scope stores behind `s_waitcnt vmcnt(0)`, a relaxed ticket, and sc1 loads
    (csrc/conv_sf16.hip).  That rests on write-through-and-acknowledge behaviour the memory model does not spell out, so it is held
    to evidence: 100 tiles x 4 shares = 400 workgroups, 50 per XCD -- 50 % 4 != 0, so a share group straddles EVERY XCD boundary --
    launched 300 times with two alternating inputs (a stale or half-written share of the previous launch would carry the OTHER
    input's sums).  Every launch must reproduce the first result of its input bit for bit, and the tickets end at zero."""
    from sound_event_detection_dcase2017_task4_amd import ops
    L = ops._lib.lib()
    B, H, W, Cin, Cout = 25, 8, 8, 512, 256
    ks = int(L.sed_conv_sf16_ksplit(B, H, W, Cin, Cout))
    assert ks == 4 and (B * (Cout // 64) * ks // 8) % ks != 0
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn((B, H, W, Cin), generator=g).cuda() for _ in range(2)]
    w = ((torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * 0.02).cuda()
    pack = ops.pack_sf16(w)
    ams = [ops.amax_of(x) for x in xs]
    first = [ops.conv3x3_sf16(x, pack, B, H, W, Cin, Cout, x_amax=a).clone() for x, a in zip(xs, ams)]
    torch.cuda.synchronize()
    assert not torch.equal(first[0], first[1])
    want = _ref(xs[0].cpu(), w.cpu())
    rel, mx = _err(first[0], want)
    assert rel < 1e-6 and mx < 1e-5
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for it in range(300):
        y = ops.conv3x3_sf16(xs[it & 1], pack, B, H, W, Cin, Cout, x_amax=ams[it & 1])
        bad += (y != first[it & 1]).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0
    assert all(int(t.abs().sum()) == 0 for t in ops._TICKETS.values())
    ops.check_device_errors(synchronize=True)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 37, 32, 64, 128), (40, 125, 8, 64, 128), (9, 1001, 64, 64, 64)])
def test_finalize_launches_carry_the_operand_amax_and_the_gradient_bound(B, H, W, Cin, Cout):
    """Round 6 (launch diet): the amax of the never-materialised operand relu(bn1(y1)) and the bound of bn1's backward output
    used to be launches of their own (sed_act_amax, sed_grad_bound) behind the BatchNorm finalize launches; now the finalize
    launches leave them as by-products -- inside the launch while one wave per channel walks the parts in a few trips (<= 512
    parts), by the chunked follow-up launch beyond (the third shape: 2259 parts).  Both must equal the stand-alone launches BIT
    FOR BIT (max / min and the monotone affine are exact), and the statistics / coefficients must not move."""
    from sound_event_detection_dcase2017_task4_amd import ops, _lib
    g = torch.Generator().manual_seed(B + H)
    x = torch.randn((B, H, W, Cin), generator=g).cuda()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * 0.05).cuda()
    P = int(_lib.lib().sed_conv_sf16_num_parts(B, H, W, Cout))
    parts = torch.zeros((P * 2 * Cout + P,), device="cuda")
    mm = torch.full((P, 2, Cout), float("nan"), device="cuda")
    y = ops.conv3x3_sf16(x, ops.pack_sf16(w), B, H, W, Cin, Cout, epi=1, partials=parts, minmax=mm)
    gam = (torch.randn(Cout, generator=g)).cuda()                 # both signs
    bet = (torch.randn(Cout, generator=g) * 0.5).cuda()
    M = B * H * W
    st0 = ops.bn_finalize(parts, P, -1, M, gam, bet, None, None)
    a_sep = ops.act_amax(mm, P, Cout, st0)
    a_fold = ops._amax_buf("cuda")
    st1 = ops.bn_finalize(parts, P, -1, M, gam, bet, None, None, minmax=mm, act_amax_out=a_fold)
    torch.cuda.synchronize()
    for k in ("mean", "invstd", "scale", "shift"):
        assert torch.equal(getattr(st0, k), getattr(st1, k)), k
    assert ops.amax_value(a_fold) == ops.amax_value(a_sep) == ops.amax_value(ops.act_amax_full(y, st0)) > 0
    # backward: (sum dy, sum dy * xhat) partials of a made-up gradient -> coefficients + the bound of |a*dy + b*y + c|
    partb = (torch.randn((P, 2, Cout), generator=g) * 1e-3).cuda()
    g_amax = ops.amax_of((torch.randn(1000, generator=g) * 1e-4).cuda())
    dg0, db0, coef0 = ops.bn_bwd_finalize(partb, P, M, st0)
    b_sep = ops._amax_buf("cuda")
    ops._call("sed_grad_bound", ops._ptr(mm), P, Cout, ops._ptr(coef0), ops._ptr(g_amax), 1.0, ops._ptr(b_sep), None, ops._stream())
    b_fold = ops._amax_buf("cuda")
    dg1, db1, coef1 = ops.bn_bwd_finalize(partb, P, M, st0, bound=(None, g_amax, 1.0, b_fold), minmax=mm)
    torch.cuda.synchronize()
    assert torch.equal(coef0, coef1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert ops.amax_value(b_fold) == ops.amax_value(b_sep) > 0
    # the bound really bounds: |a*dy + b*y + c| <= bound for |dy| <= G and y in the tensor
    G = ops.amax_value(g_amax)
    a, b, c = coef0[0].double(), coef0[1].double(), coef0[2].double()
    worst = (a.abs() * G + (b * y.double().reshape(-1, Cout) + c).abs().max(dim=0).values).max()
    assert float(worst) <= ops.amax_value(b_fold)

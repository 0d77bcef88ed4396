"""FusedAdamAmsgrad's direct gradient sinks, the derived-weight caches and the differentiability of the secondary model
outputs (framewise_output / embedding), on the GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import frontend as ofe
from oracle import model as om

CTOR = (32000, 1024, 320, 64, 50, 14000, 17)


def _build(mt, seed=5):
    from sound_event_detection_dcase2017_task4_amd.pytorch import models
    m = getattr(models, mt)(*CTOR)
    m.load_state_dict(om.recipe_state(mt, seed))
    return m.to("cuda").train()


def _batch(rows=8, L=32000, seed=11):
    rs = np.random.RandomState(seed)
    x = torch.from_numpy((rs.randn(rows, L) * 0.1).astype(np.float32)).cuda()
    y = torch.from_numpy((rs.rand(rows, 17) < 0.2).astype(np.float32)).cuda()
    lam = torch.from_numpy(ofe.mixup_lambdas(rows, np.random.RandomState(1234)).astype(np.float32)).cuda()
    torch.manual_seed(7)
    stripes = ofe.draw_specaug_stripes(rows, L // 320 + 1, 64)
    return x, y, lam, stripes


@pytest.mark.parametrize("mt", ["Cnn_9layers_FrameAvg", "Cnn_9layers_FrameMax", "Cnn_9layers_Gru_FrameAtt",
                                "Cnn_9layers_Transformer_FrameAtt"])
def test_direct_sinks_equal_autograd_accumulation(mt):
    """Gradients written straight into the flat buffer by the backward kernels (direct_grads=True: autograd gets None,
    no accumulate kernels) are BIT-identical to the ordinary autograd path accumulating into the zeroed views, for every
    parameter of the model; three steps of both variants leave identical parameters."""
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
    x, y, lam, stripes = _batch()
    kw = {}
    if "Transformer" in mt:
        ma, mf = om.dropout_masks(3, 4, 12)
        kw = {"dropout_masks": (ma.cuda(), mf.cuda())}
    flats, grads = [], []
    for direct in (True, False):
        m = _build(mt)
        opt = FusedAdamAmsgrad(m, lr=1e-3, direct_grads=direct)
        for it in range(3):
            out = m(x, lam, specaug_stripes=stripes, **kw)
            loss = clip_bce(out, {"target": do_mixup(y, lam)})
            opt.zero_grad()
            loss.backward()
            if it == 0:
                grads.append(opt.flat_grad.clone())
                for p in opt.params:                       # .grad stays a view of the flat buffer in both modes
                    assert p.grad is not None and p.grad.data_ptr() >= opt.flat_grad.data_ptr()
            opt.step()
        flats.append(opt.flat.clone())
    assert grads[0].abs().sum().item() > 0
    assert torch.equal(grads[0], grads[1])
    assert torch.equal(flats[0], flats[1])


def test_second_backward_without_step_is_refused():
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    m = _build("Cnn_9layers_FrameAvg")
    opt = FusedAdamAmsgrad(m, lr=1e-3)
    x, y, lam, stripes = _batch(rows=4)
    for _ in range(2):
        loss = clip_bce(m(x, None, specaug_stripes=stripes), {"target": y})
        loss.backward()
    with pytest.raises(RuntimeError, match="direct_grads=False"):
        opt.step()
    # zero_grad() starts a fresh cycle
    loss = clip_bce(m(x, None, specaug_stripes=stripes), {"target": y})
    opt.zero_grad()
    loss.backward()
    opt.step()


def test_derived_weight_caches_follow_the_parameters():
    """The padded head matrix / stacked GRU operands and (round 3) the split-f16 packs of the conv weights are cached between
    steps; an optimiser step (raw-pointer update), an in-place torch update and load_state_dict must each invalidate them."""
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    mt = "Cnn_9layers_Gru_FrameAtt"
    m = _build(mt).eval()
    x = _batch(rows=4)[0]

    def clip():
        with torch.no_grad():
            return m(x)["clipwise_output"].clone()

    def oracle():
        st = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        with torch.no_grad():
            return om.forward(mt, st, x.cpu(), training=False)["clipwise_output"]

    a = clip()
    assert torch.equal(clip(), a)                                     # cache hit: same operands, same result
    with torch.no_grad():
        m.att_block.cla.weight.mul_(1.5)                              # in-place torch update bumps _version
        m.gru.weight_ih_l0.add_(0.01)
        m.conv_block3.conv2.weight.mul_(1.7)                          # a cached split-f16 pack (eval mode: forward layout)
        m.conv_block2.conv1.weight.add_(0.02)
    b = clip()
    assert (b.cpu() - oracle()).abs().max().item() < 1e-4 and (a - b).abs().max().item() > 1e-4
    sd = om.recipe_state(mt, 9)
    m.load_state_dict(sd)
    c = clip()
    assert (c.cpu() - oracle()).abs().max().item() < 1e-4
    # optimiser step: parameters change through the flat buffer only
    m.train()
    opt = FusedAdamAmsgrad(m, lr=1e-2)
    xb, y, lam, stripes = _batch(rows=8)
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
    loss = clip_bce(m(xb, lam, specaug_stripes=stripes), {"target": do_mixup(y, lam)})
    opt.zero_grad(); loss.backward(); opt.step()
    m.eval()
    d = clip()
    assert (d.cpu() - oracle()).abs().max().item() < 1e-4 and (c - d).abs().max().item() > 1e-5
    ops.invalidate_weight_caches()
    assert torch.equal(clip(), d)


@pytest.mark.parametrize("mt", ["Cnn_9layers_FrameAvg", "Cnn_9layers_FrameMax", "Cnn_9layers_FrameAtt"])
def test_framewise_output_and_embedding_are_differentiable(mt):
    """The reference modules are ordinary differentiable nn.Modules: a loss on `framewise_output` (strong labels) or on
    `embedding` must produce the same parameter gradients as the oracle's autograd -- not a silent zero."""
    m = _build(mt)
    x, y, lam, stripes = _batch(rows=6)
    rs = np.random.RandomState(3)
    out = m(x, lam, specaug_stripes=stripes)
    wf = torch.from_numpy(rs.randn(*out["framewise_output"].shape).astype(np.float32)).cuda()
    we = torch.from_numpy(rs.randn(*out["embedding"].shape).astype(np.float32)).cuda()
    wc = torch.from_numpy(rs.randn(*out["clipwise_output"].shape).astype(np.float32)).cuda()
    loss = (out["framewise_output"] * wf).sum() + (out["embedding"] * we).sum() * 1e-2 + (out["clipwise_output"] * wc).sum()
    keys = ["fc.weight", "fc.bias"] if "fc.weight" in dict(m.named_parameters()) else \
        ["att_block.att.weight", "att_block.att.bias", "att_block.cla.weight", "att_block.cla.bias"]
    keys += ["conv_block4.bn2.weight", "bn0.weight"]
    params = dict(m.named_parameters())
    got = torch.autograd.grad(loss, [params[k] for k in keys])
    st = {k: v.double() if v.is_floating_point() else v for k, v in om.recipe_state(mt, 5).items()}
    ofe._CACHE.clear()
    consts = ofe._consts()
    for k in list(consts):
        consts[k] = consts[k].double()
    try:
        for k in keys:
            st[k].requires_grad_(True)
        o = om.forward(mt, st, x.cpu().double(), training=True, mixup_lambda=lam.cpu().double(), stripes=stripes)
        lo = (o["framewise_output"] * wf.cpu().double()).sum() + (o["embedding"] * we.cpu().double()).sum() * 1e-2 + \
            (o["clipwise_output"] * wc.cpu().double()).sum()
        want = torch.autograd.grad(lo, [st[k] for k in keys])
    finally:
        ofe._CACHE.clear()
    assert abs(loss.item() - lo.item()) < 1e-3 * max(1.0, abs(lo.item()))
    for k, g, w in zip(keys, got, want):
        if w.norm().item() < 1e-6:                     # structurally zero (att.bias: the attention is shift invariant)
            assert g.abs().max().item() < 1e-5, (k, g.abs().max().item())
            continue
        err = (g.double().cpu() - w).norm().item() / max(w.norm().item(), 1e-12)
        assert err < 2e-3, (k, err)


def test_att_head_gradients_through_all_three_outputs():
    """AttHeadFn (models.py:118-149) with gradients arriving through clip, cla AND norm_att (the clamp active on part of
    the logits) against the same arithmetic written with torch ops on the CPU in float64."""
    from sound_event_detection_dcase2017_task4_amd import ops
    g = torch.Generator().manual_seed(21)
    B, T, C, K = 5, 23, 512, 17
    feat = torch.randn(B, T, C, generator=g)
    w_att = torch.randn(K, C, 1, generator=g) * 0.35            # |logit| reaches > 10 on a few percent of the entries
    w_cla = torch.randn(K, C, 1, generator=g) * 0.05
    b_att, b_cla = torch.randn(K, generator=g) * 0.1, torch.randn(K, generator=g) * 0.1
    wc, wl, wn = torch.randn(B, K, generator=g), torch.randn(B, T, K, generator=g), torch.randn(B, T, K, generator=g)

    def ref(feat, w_att, b_att, w_cla, b_cla):
        z = torch.einsum("btc,kc->btk", feat, w_att[:, :, 0]) + b_att
        att = torch.exp(torch.clamp(z, -10, 10)) + 1e-6
        natt = att / att.sum(dim=1, keepdim=True)
        cla = torch.sigmoid(torch.einsum("btc,kc->btk", feat, w_cla[:, :, 0]) + b_cla)
        return (natt * cla).sum(dim=1), cla, natt, z

    leaves = [t.double().requires_grad_(True) for t in (feat, w_att, b_att, w_cla, b_cla)]
    clip, cla, natt, z = ref(*leaves)
    assert ((z.abs() > 10).float().mean().item() > 0.005) and ((z.abs() < 10).float().mean().item() > 0.5)
    loss = (clip * wc.double()).sum() + (cla * wl.double()).sum() + (natt * wn.double()).sum() * 3.0
    want = torch.autograd.grad(loss, leaves)
    dl = [t.cuda().requires_grad_(True) for t in (feat, w_att, b_att, w_cla, b_cla)]
    c2, l2, n2 = ops.AttHeadFn.apply(*dl)
    assert (c2.detach().cpu() - clip.detach()).abs().max().item() < 1e-5
    assert (n2.detach().cpu() - natt.detach()).abs().max().item() < 1e-5
    loss2 = (c2 * wc.cuda()).sum() + (l2 * wl.cuda()).sum() + (n2 * wn.cuda()).sum() * 3.0
    got = torch.autograd.grad(loss2, dl)
    for name, a, b in zip(("feat", "w_att", "b_att", "w_cla", "b_cla"), got, want):
        err = (a.double().cpu() - b).norm().item() / b.norm().item()
        assert err < 2e-4, (name, err)


def test_side_stream_weight_gradients_are_joined_and_bit_identical(monkeypatch):
    """The weight-gradient kernels run on a side HIP stream (ops._fork_wgrad).  With kernels long enough for a missing join
    to show (32 ten-second waveforms: 5-20 ms per weight gradient), the flat gradient read on the MAIN stream right after
    backward() must be complete -- bit-identical to the one-stream schedule -- on every repetition, nothing may be left
    pending, and the following optimiser steps must agree bit for bit as well."""
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
    x, y, lam, stripes = _batch(rows=32, L=320000, seed=21)
    results = {}
    for side in (False, True, True):
        monkeypatch.setattr(ops, "WGRAD_SIDE_STREAM", side)
        m = _build("Cnn_9layers_FrameAvg")
        opt = FusedAdamAmsgrad(m, lr=1e-3)
        snaps = []
        for it in range(3):
            out = m(x, lam, specaug_stripes=stripes)
            loss = clip_bce(out, {"target": do_mixup(y, lam)})
            opt.zero_grad()
            loss.backward()
            assert not ops._PENDING
            snaps.append(opt.flat_grad.clone())           # main stream, no device synchronisation in between
            opt.step()
        snaps.append(opt.flat.clone())
        torch.cuda.synchronize()
        if side in results:
            for a, b in zip(results[side], snaps):
                assert torch.equal(a, b)
        results[side] = snaps
    for a, b in zip(results[False], results[True]):
        assert torch.equal(a, b)
    assert float(results[True][0].abs().sum()) > 0


def test_model_zero_grad_does_not_wipe_direct_gradients():
    """`model.zero_grad()` sets every `.grad` to None (torch default); the backward kernels still write through their sinks
    into the flat buffer.  `_gather()` must re-attach those slices, not zero them: the parameters move exactly as with
    `optimizer.zero_grad()`."""
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    x, y, lam, stripes = _batch(rows=4)
    flats = []
    for use_model_zero_grad in (False, True):
        m = _build("Cnn_9layers_FrameAtt")
        opt = FusedAdamAmsgrad(m, lr=1e-3)
        start = opt.flat.clone()
        for _ in range(2):
            loss = clip_bce(m(x, None, specaug_stripes=stripes), {"target": y})
            if use_model_zero_grad:
                m.zero_grad()
                assert all(p.grad is None for p in opt.params)
            else:
                opt.zero_grad()
            loss.backward()
            opt.step()
        assert (opt.flat - start).abs().max().item() > 5e-4            # two Adam steps of lr 1e-3
        got_none = [p for p in opt.params if p.grad is None]            # only the reference's unused att_block.bn_att.*
        assert len(got_none) == (2 if use_model_zero_grad else 0)
        assert all(p.grad is None or p.grad.data_ptr() == opt.flat_grad.data_ptr() + 4 * off
                   for p, off in zip(opt.params, opt.offsets))
        flats.append(opt.flat.clone())
    assert torch.equal(flats[0], flats[1])


def test_backward_that_raises_leaves_no_stale_side_stream_state(monkeypatch):
    """A backward pass that dies between a side-stream fork and its join must not poison the next one (stale sinks
    reporting ready -> a spurious 'backward() ran more than once')."""
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    m = _build("Cnn_9layers_FrameAvg")
    opt = FusedAdamAmsgrad(m, lr=1e-3)
    x, y, lam, stripes = _batch(rows=4)
    real = ops.bn_bwd_finalize
    calls = {"n": 0}

    def dying(*a, **k):
        calls["n"] += 1
        if calls["n"] == 4:                                  # somewhere in the middle of the conv stack's backward
            raise RuntimeError("injected failure")
        return real(*a, **k)

    monkeypatch.setattr(ops, "bn_bwd_finalize", dying)
    loss = clip_bce(m(x, None, specaug_stripes=stripes), {"target": y})
    opt.zero_grad()
    with pytest.raises(RuntimeError, match="injected failure"):
        loss.backward()
    monkeypatch.setattr(ops, "bn_bwd_finalize", real)
    before = opt.flat.clone()
    loss = clip_bce(m(x, None, specaug_stripes=stripes), {"target": y})
    opt.zero_grad()                                          # drops what the dead pass left behind
    assert not ops._PENDING
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert (opt.flat - before).abs().max().item() > 5e-4


def test_nan_input_is_refused_by_the_guard_and_fp32_fallback_matches_torch_semantics():
    """A NaN waveform: with the split-f16 kernels the optimiser refuses the step and raises NonFiniteOperand (parameters
    intact); with ops.USE_SF16 = False the step goes through like torch.optim.Adam would (NaN parameters) -- which is what
    the train CLI switches to."""
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    m = _build("Cnn_9layers_FrameAvg")
    opt = FusedAdamAmsgrad(m, lr=1e-3)
    x, y, lam, stripes = _batch(rows=4)
    xb = x.clone(); xb[1, 5000] = float("nan")
    start = opt.flat.clone()
    buffers_before = {k: v.clone() for k, v in m.named_buffers()}
    with torch.no_grad():
        eval_before = m.eval()(x, None)["clipwise_output"].clone()
    m.train()
    ops.check_device_errors(synchronize=True)
    loss = clip_bce(m(xb, None, specaug_stripes=stripes), {"target": y})
    opt.zero_grad(); loss.backward()
    with pytest.raises(ops.NonFiniteOperand):
        opt.step()
        ops.check_device_errors(synchronize=True)            # (the flag is host-mapped: normally seen by step() itself)
    assert torch.equal(opt.flat, start) and opt.step_count == 0 and opt.skipped_steps == 1
    # the BatchNorm buffers are as intact as the parameters: the poisoned forward pass did NOT blend its NaN batch statistics
    # into running_mean / running_var (bn_finalize is guarded on the device), the counters can be rolled back, and an
    # eval-mode forward -- which uses the running statistics -- is finite and equal to the one before the refused step
    for name, buf in m.named_buffers():
        assert torch.isfinite(buf.float()).all(), name
        assert torch.equal(buf, buffers_before[name]) or name.endswith("num_batches_tracked"), name
    n0 = int(buffers_before["bn0.num_batches_tracked"])
    assert int(m.bn0.num_batches_tracked) == n0 + 1
    ops.rollback_bn_counters(m, 1)
    assert all(int(c) == n0 for c in m.bn_counters())
    with torch.no_grad():
        ev = m.eval()(x, None)["clipwise_output"]
    m.train()
    assert torch.isfinite(ev).all() and torch.equal(ev, eval_before)
    # a clean batch trains normally afterwards
    loss = clip_bce(m(x, None, specaug_stripes=stripes), {"target": y})
    opt.zero_grad(); loss.backward(); opt.step()
    ops.check_device_errors(synchronize=True)
    assert opt.step_count == 1 and torch.isfinite(opt.flat).all() and not torch.equal(opt.flat, start)
    # reference semantics on the fp32 kernels
    prev, ops.USE_SF16 = ops.USE_SF16, False
    try:
        loss = clip_bce(m(xb, None, specaug_stripes=stripes), {"target": y})
        opt.zero_grad(); loss.backward(); opt.step()
        ops.check_device_errors(synchronize=True)
        assert torch.isnan(opt.flat).any()
    finally:
        ops.USE_SF16 = prev



def test_lagged_poll_reports_refused_steps_at_a_deterministic_step():
    """FusedAdamAmsgrad(poll_lag=2): the host learns about a refused step exactly two optimizer.step() calls later -- by
    waiting for that step's event and reading the status word its Adam kernel wrote -- never earlier, never later, whatever
    the timing (this is what lets every rank of a data-parallel job raise from the SAME call).  The exception carries the
    number of refused steps (the poisoned one + the two issued since: the device flag is sticky), step_count is corrected,
    parameters / moments / BatchNorm buffers are untouched, and training continues afterwards."""
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    m = _build("Cnn_9layers_FrameAvg")
    opt = FusedAdamAmsgrad(m, lr=1e-3, poll_lag=2)
    x, y, lam, stripes = _batch(rows=4)
    xb = x.clone(); xb[1, 5000] = float("nan")
    ops.check_device_errors(synchronize=True)

    def step(inp):
        loss = clip_bce(m(inp, None, specaug_stripes=stripes), {"target": y})
        opt.zero_grad(); loss.backward(); opt.step()

    n0 = int(m.bn0.num_batches_tracked)
    for _ in range(3):
        step(x)                                              # clean steps: nothing to report
    assert opt.step_count == 3
    before = opt.flat.clone()
    bufs = {k: v.clone() for k, v in m.named_buffers() if not k.endswith("num_batches_tracked")}
    step(xb)                                                 # poisoned step 4: refused on the device, not yet reported
    torch.cuda.synchronize()                                 # (even with the flag long visible to the host ...)
    step(x)                                                  # ... step 5 -- refused too: the flag is sticky -- does not report
    with pytest.raises(ops.NonFiniteOperand) as ei:
        step(x)                                              # step 6 polls step 4: reports 4, 5 and 6
    assert ei.value.skipped_steps == 3 and opt.step_count == 3 and opt.skipped_steps == 3
    assert torch.equal(opt.flat, before)
    for k, v in m.named_buffers():
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(v, bufs[k]), k
    assert int(m.bn0.num_batches_tracked) == n0 + 6
    ops.rollback_bn_counters(m, ei.value.skipped_steps)
    assert int(m.bn0.num_batches_tracked) == n0 + 3
    for _ in range(3):
        step(x)                                              # flags were cleared: the run goes on
    opt.poll(0)
    assert opt.step_count == 6 and torch.isfinite(opt.flat).all() and not torch.equal(opt.flat, before)
    # poll(0) drains: a poisoned LAST step is reported without further steps
    step(xb)
    with pytest.raises(ops.NonFiniteOperand) as ei:
        opt.poll(0)
    assert ei.value.skipped_steps == 1 and opt.step_count == 6
    ops.check_device_errors(synchronize=True, nonfinite=True)


@pytest.mark.parametrize("graphed", [False, True])
def test_step_refused_after_its_forward_pass_takes_the_batchnorm_statistics_back(graphed):
    """The forward pass of a step is clean and installs its BatchNorm running statistics; the NaN arrives afterwards (here: a
    non-finite loss gradient, which is what a poisoned backward pass, a non-finite all-reduced gradient or another rank's
    flag look like to this rank).  The Adam kernel refuses the step -- and the running statistics the forward pass had
    already blended in are taken back behind it (sed_bn_restore), so a refused step leaves the buffers exactly as they were
    and the re-run of the same batch does not blend it twice.  Eager and under the HIP graph (replays run no Python)."""
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.graph import GraphedTrainStep
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    m = _build("Cnn_9layers_FrameAvg")
    opt = FusedAdamAmsgrad(m, lr=1e-3)
    x, y, lam, stripes = _batch(rows=4)
    poison = torch.ones((), device="cuda")

    def loss_func(out, tgt):
        return clip_bce(out, tgt) * poison                   # poison = NaN: finite forward pass, NaN gradient everywhere

    ops.check_device_errors(synchronize=True)
    if graphed:
        stepper = GraphedTrainStep(m, opt, loss_func, mixup=False, eager_steps=1)
        step = lambda: stepper(x, y, None, stripes)
    else:
        def step():
            loss = loss_func(m(x, None, specaug_stripes=stripes), {"target": y})
            opt.zero_grad(); loss.backward(); opt.step()
    for _ in range(3):
        step()                                               # clean steps (the third one replays the graph)
    ops.check_device_errors(synchronize=True)
    assert opt.step_count == 3 and (not graphed or stepper.replays >= 1)
    before = opt.flat.clone()
    bufs = {k: v.clone() for k, v in m.named_buffers() if not k.endswith("num_batches_tracked")}
    poison.fill_(float("nan"))
    with pytest.raises(ops.NonFiniteOperand):
        step()
        ops.check_device_errors(synchronize=True)
    assert torch.equal(opt.flat, before) and opt.step_count == 3
    for k, v in m.named_buffers():
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(v, bufs[k]), k                # restored bit for bit
    ops.rollback_bn_counters(m, 1)
    poison.fill_(1.0)
    step()                                                   # the run goes on, and now the statistics DO move
    ops.check_device_errors(synchronize=True)
    assert opt.step_count == 4 and not torch.equal(m.bn0.running_mean, bufs["bn0.running_mean"])


def test_refused_step_takes_back_every_forward_pass_since_the_last_step():
    """Gradient accumulation (direct_grads=False) or any second train-mode forward before optimizer.step(): every one of those
    passes installed running statistics, and a refused step must take ALL of them back (newest first), not only the newest
    pass's -- the buffers end up exactly as they were before the first pass of the cycle (round-5 advisor, ops._BN_LAST)."""
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    m = _build("Cnn_9layers_FrameAvg")
    opt = FusedAdamAmsgrad(m, lr=1e-3, direct_grads=False)
    x, y, lam, stripes = _batch(rows=4)
    x2 = x * 0.5
    ops.check_device_errors(synchronize=True)
    loss = clip_bce(m(x, None, specaug_stripes=stripes), {"target": y})
    opt.zero_grad(); loss.backward(); opt.step()                 # one clean step
    ops.check_device_errors(synchronize=True)
    before = opt.flat.clone()
    bufs = {k: v.clone() for k, v in m.named_buffers() if not k.endswith("num_batches_tracked")}
    opt.zero_grad()
    clip_bce(m(x, None, specaug_stripes=stripes), {"target": y}).backward()          # pass 1 of the cycle: clean
    mid = m.bn0.running_mean.clone()
    assert not torch.equal(mid, bufs["bn0.running_mean"])
    (clip_bce(m(x2, None, specaug_stripes=stripes), {"target": y}) * float("nan")).backward()    # pass 2: NaN gradient
    assert not torch.equal(m.bn0.running_mean, mid)
    with pytest.raises(ops.NonFiniteOperand):
        opt.step()
        ops.check_device_errors(synchronize=True)
    assert torch.equal(opt.flat, before)
    for k, v in m.named_buffers():
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(v, bufs[k]), k                    # BOTH passes taken back, bit for bit


def test_optimizer_state_interchanges_with_torch_adam_amsgrad():
    """Checkpoint interchange (reference main.py:144-145, :222-230: the 'optimizer' entry is torch.optim.Adam(amsgrad=True)'s
    state_dict over ALL model.parameters(), the frozen front-end tensors included).  Two steps of stock torch Adam and of
    FusedAdamAmsgrad on the same gradients agree; the stock state_dict then loads into a fresh FusedAdamAmsgrad (moments, amsgrad
    maximum, step), this class's torch_state_dict() loads into stock Adam, and a third step agrees everywhere.  Parameters that
    never receive a gradient (`att_block.bn_att.*`) carry no torch state and load as zero moments.  A state that is not
    Adam-amsgrad for this model is refused with the reason."""
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    mt = "Cnn_9layers_FrameAtt"
    m = _build(mt)
    every = list(m.parameters())
    names = [n for n, _ in m.named_parameters()]
    ref = [p.detach().clone().requires_grad_(p.requires_grad) for p in every]
    topt = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.999), eps=1e-08, weight_decay=0., amsgrad=True)
    opt = FusedAdamAmsgrad(m, lr=1e-3, direct_grads=False)
    unused = [i for i, n in enumerate(names) if ".bn_att." in n]
    assert unused and not every[0].requires_grad
    g = torch.Generator(device="cuda").manual_seed(3)

    def fused_step(o, model_params, grads):
        o.zero_grad()
        for p, gr in zip(model_params, grads):
            if gr is not None:
                p.grad.copy_(gr)
        o.step()

    def torch_step(o, params, grads):
        for p, gr in zip(params, grads):
            p.grad = gr.clone() if gr is not None else None
        o.step()

    def grads():
        return [torch.randn(p.shape, device="cuda", generator=g) * 1e-2 if (p.requires_grad and i not in unused) else None
                for i, p in enumerate(every)]

    for _ in range(2):
        gr = grads()
        torch_step(topt, ref, gr)
        fused_step(opt, every, gr)
    torch.cuda.synchronize()
    for p, q, n in zip(every, ref, names):
        assert torch.allclose(p, q, rtol=0, atol=2e-7), n
    sd = topt.state_dict()
    assert sorted(sd["state"].keys()) == [i for i, p in enumerate(every) if p.requires_grad and i not in unused]
    # torch state -> a fresh fused optimiser
    m2 = _build(mt)
    with torch.no_grad():
        for p, q in zip(m2.parameters(), ref):
            p.copy_(q)
    opt2 = FusedAdamAmsgrad(m2, lr=1e-3, direct_grads=False)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 2
    # (torch forms exp_avg as lerp(m, g, 1 - beta1), the fused kernel as beta1 * m + (1 - beta1) * g: a rounding of the 1e-3-sized
    # moments apart, i.e. up to 1e-10 absolute where the two terms cancel)
    for a, b, atol in ((opt2.exp_avg, opt.exp_avg, 1e-9), (opt2.exp_avg_sq, opt.exp_avg_sq, 1e-13), (opt2.max_exp_avg_sq, opt.max_exp_avg_sq, 1e-13)):
        assert torch.allclose(a, b, rtol=1e-5, atol=atol), float((a - b).abs().max())
    # fused state -> stock Adam
    ref3 = [q.detach().clone().requires_grad_(q.requires_grad) for q in ref]
    topt3 = torch.optim.Adam(ref3, lr=1e-3, betas=(0.9, 0.999), eps=1e-08, weight_decay=0., amsgrad=True)
    topt3.load_state_dict(opt.torch_state_dict())
    gr = grads()
    torch_step(topt, ref, gr)
    torch_step(topt3, ref3, gr)
    fused_step(opt, every, gr)
    fused_step(opt2, list(m2.parameters()), gr)
    torch.cuda.synchronize()
    for p, p2, q, q3, n in zip(every, m2.parameters(), ref, ref3, names):
        assert torch.allclose(p, q, rtol=0, atol=3e-7) and torch.allclose(p2, q, rtol=0, atol=3e-7) and torch.allclose(q3, q, rtol=0, atol=3e-7), n
    for i in unused:                                                   # never moved, anywhere
        assert torch.equal(every[i], ref[i])
    # refusals carry the reason
    plain = torch.optim.Adam([q for q in ref], lr=1e-3)
    with pytest.raises(ValueError, match="WITHOUT amsgrad"):
        opt2.load_state_dict(plain.state_dict())
    short = torch.optim.Adam(ref[:-1], lr=1e-3, amsgrad=True)
    with pytest.raises(ValueError, match="holds %d parameters" % (len(ref) - 1)):
        opt2.load_state_dict(short.state_dict())
    with pytest.raises(ValueError, match="unknown optimiser state layout"):
        opt2.load_state_dict({"momentum_buffer": 1})
    assert opt2.step_count == 3                                        # a refused load changed nothing

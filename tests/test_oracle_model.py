"""CPU: pin the oracle's trunk/heads/loss/Adam against golden vectors produced by the genuine
reference code (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import frontend as ofe
from oracle import model as om

torch.set_num_threads(8)


def waves(seed, n, length):
    return (np.random.RandomState(seed).randn(n, length) * 0.1).astype(np.float32)


def targets(seed, n):
    return (np.random.RandomState(seed).rand(n, 17) < 0.2).astype(np.float32)


def summarize(t):
    f = t.detach().reshape(-1).double()
    return np.array([f.sum().item(), f.abs().sum().item()] + f[:14].tolist() +
                    [0.0] * max(0, 14 - f.numel()), dtype=np.float64)[:16]


def check_summary(got, want, rtol, what, slack=0.0):
    """`slack`: absolute allowance for Adam's sign-like updates on noise-level gradients."""
    scale = max(abs(want[1]), 1e-12)
    assert abs(got[0] - want[0]) <= rtol * scale + slack, (what, got[0], want[0])
    assert abs(got[1] - want[1]) <= rtol * scale + slack, (what, got[1], want[1])
    np.testing.assert_allclose(got[2:], want[2:], rtol=rtol * 50, atol=rtol * scale / 10 + slack, err_msg=what)


SEEDS = {mt: i + 1 for i, mt in enumerate(om.MODEL_TYPES)}


@pytest.mark.parametrize("mt", om.MODEL_TYPES)
def test_state_layout_matches_survey_counts(mt):
    st = om.recipe_state(mt, 1)
    n_train = sum(st[k].numel() for k in om.trainable_keys(mt))
    n_frozen = sum(st[k].numel() for k in om.FROZEN_KEYS)
    want = {"Cnn_9layers_FrameMax": 4694993, "Cnn_9layers_FrameAvg": 4694993, "Cnn_9layers_FrameAtt": 4703748,
            "Cnn_9layers_Gru_FrameAvg": 5877713, "Cnn_9layers_Gru_FrameAtt": 5886468,
            "Cnn_9layers_Transformer_FrameAvg": 5746641, "Cnn_9layers_Transformer_FrameAtt": 5755396}[mt]
    assert n_train == want and n_frozen == 1083456


@pytest.mark.parametrize("mt", om.MODEL_TYPES)
def test_eval_forward_matches_reference(mt, golden_dir):
    fx = np.load(os.path.join(golden_dir, mt + ".npz"))
    seed = SEEDS[mt]
    st = om.recipe_state(mt, seed)
    with torch.no_grad():
        o = om.forward(mt, st, torch.from_numpy(waves(100 + seed, 4, 32000)), training=False)
    assert o["framewise_output"].shape == (4, 96, 17)
    np.testing.assert_allclose(o["clipwise_output"].numpy(), fx["eval_clip"], atol=2e-6)
    np.testing.assert_allclose(o["framewise_output"].numpy()[:, ::8], fx["eval_frame"], atol=2e-6)
    check_summary(summarize(o["embedding"]), fx["eval_embedding"], 1e-5, "embedding")


def test_eval_forward_10s(golden_dir):
    mt = "Cnn_9layers_FrameAvg"
    fx = np.load(os.path.join(golden_dir, mt + ".npz"))
    st = om.recipe_state(mt, SEEDS[mt])
    with torch.no_grad():
        o = om.forward(mt, st, torch.from_numpy(waves(200 + SEEDS[mt], 2, 320000)), training=False)
    assert o["framewise_output"].shape == (2, 1000, 17) and o["embedding"].shape == (2, 512, 125)
    np.testing.assert_allclose(o["clipwise_output"].numpy(), fx["eval10_clip"], atol=2e-6)
    np.testing.assert_allclose(o["framewise_output"].numpy()[:, ::8], fx["eval10_frame"], atol=2e-6)


@pytest.mark.parametrize("mt", om.MODEL_TYPES)
def test_train_forward_matches_reference(mt, golden_dir):
    fx = np.load(os.path.join(golden_dir, mt + ".npz"))
    seed = SEEDS[mt]
    st = om.recipe_state(mt, seed)
    lam = torch.from_numpy(fx["train_lambda"])
    with torch.no_grad():
        o = om.forward(mt, st, torch.from_numpy(waves(300 + seed, 6, 32000)), training=True,
                       mixup_lambda=lam, stripes=fx["train_stripes"], dropout_seed=int(fx["train_dropout_seed"]))
    assert o["clipwise_output"].shape == (3, 17)
    np.testing.assert_allclose(o["clipwise_output"].numpy(), fx["train_clip"], atol=1e-5)
    np.testing.assert_allclose(o["framewise_output"].numpy()[:, ::8], fx["train_frame"], atol=1e-5)
    np.testing.assert_allclose(st["bn0.running_mean"].numpy(), fx["train_bn0_running_mean"], rtol=1e-5)
    np.testing.assert_allclose(st["bn0.running_var"].numpy(), fx["train_bn0_running_var"], rtol=1e-5)
    np.testing.assert_allclose(st["conv_block4.bn2.running_var"].numpy(), fx["train_b4bn2_running_var"],
                               rtol=1e-4)
    assert int(st["bn0.num_batches_tracked"]) == 4


@pytest.mark.parametrize("mt", om.MODEL_TYPES)
def test_three_train_steps_match_reference(mt, golden_dir):
    fx = np.load(os.path.join(golden_dir, mt + ".npz"))
    seed = SEEDS[mt]
    st = om.recipe_state(mt, seed)
    keys = om.trainable_keys(mt)
    for k in keys:
        st[k].requires_grad_(True)
    opt_state = {k: [torch.zeros_like(st[k]) for _ in range(3)] for k in keys}
    rs = np.random.RandomState(1234)
    unused = set(fx["grad0_none_keys"].tolist())
    for it in range(3):
        xw = torch.from_numpy(waves(700 + 10 * seed + it, 8, 32000))
        tg = torch.from_numpy(targets(800 + 10 * seed + it, 8))
        lam = torch.from_numpy(ofe.mixup_lambdas(8, rs).astype(np.float32))
        o = om.forward(mt, st, xw, training=True, mixup_lambda=lam, stripes=fx["step_stripes"][it],
                       dropout_seed=int(fx["step_dropout_seeds"][it]))
        loss = om.clip_bce(o, {"target": om.do_mixup(tg, lam)})
        assert abs(loss.item() - fx["step_losses"][it]) < 2e-5, (it, loss.item(), fx["step_losses"][it])
        used = [k for k in keys if k not in unused]
        grads = torch.autograd.grad(loss, [st[k] for k in used], allow_unused=False)
        with torch.no_grad():
            for k, g in zip(used, grads):
                if it == 0:
                    check_summary(summarize(g), fx["grad0/" + k], 2e-3, "grad " + k, slack=1e-7)  # att.bias grad is structurally ~0
                    if ("gradfull0/" + k) in fx.files:
                        ref = fx["gradfull0/" + k]
                        np.testing.assert_allclose(g.numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max() + 1e-7)
                m, v, vmax = opt_state[k]
                om.adam_amsgrad_step(st[k], g, m, v, vmax, it + 1, 1e-3)
    assert unused == set(om.unused_keys(mt))
    for k, _ in om.state_layout(mt):
        if k in om.FROZEN_KEYS:
            continue
        check_summary(summarize(st[k].float()), fx["after3/" + k], 3e-4, "after3 " + k, slack=1e-3)


def test_misc_known_answers(golden_dir):
    misc = np.load(os.path.join(golden_dir, "misc.npz"))
    lam = torch.from_numpy(misc["mixup_lambda64"][:6].astype(np.float32))
    np.testing.assert_allclose(om.do_mixup(torch.from_numpy(misc["do_mixup_in"]), lam).numpy(),
                               misc["do_mixup_out"], atol=1e-7)
    loss = om.clip_bce({"clipwise_output": torch.from_numpy(misc["bce_p"])},
                       {"target": torch.from_numpy(misc["bce_y"])})
    np.testing.assert_allclose(loss.numpy(), misc["bce_loss"], rtol=1e-6)
    # log clamp at -100: p=0 with y=1 contributes exactly 100
    assert abs(float(misc["bce_loss"]) * 5 - (100 + 100 + 0.6931472 + 100 + 16.118) ) < 0.5


def test_oracle_conv_block_pool_types_vs_reference(golden_dir):
    """oracle.model.conv_block with pool_type 'avg' / 'max' / 'avg+max' against the genuine reference ConvBlock
    (tests/golden/convblock_pool.npz, made by make_golden_pool.py): output and input gradient."""
    import os
    src = open(os.path.join(golden_dir, "make_golden_pool.py")).read()
    ns = {"np": np}
    exec(src[src.index("CIN, COUT, B, H, W"):src.index("def main():")], ns)
    p, x, gout = ns["recipe"]()
    fx = np.load(os.path.join(golden_dir, "convblock_pool.npz"))
    for pt in ("avg", "max", "avg+max"):
        st = {"cb." + k: torch.from_numpy(v) for k, v in p.items()}
        for i in (1, 2):
            st["cb.bn%d.running_mean" % i] = torch.zeros(ns["COUT"])
            st["cb.bn%d.running_var" % i] = torch.ones(ns["COUT"])
            st["cb.bn%d.num_batches_tracked" % i] = torch.tensor(0)
        xt = torch.from_numpy(x).requires_grad_(True)
        y = om.conv_block(xt, st, "cb", (2, 2), True, True, pool_type=pt)
        y.backward(torch.from_numpy(gout))
        tag = pt.replace("+", "_")
        assert np.abs(y.detach().numpy() - fx[tag + "/out"]).max() < 1e-5
        assert np.abs(xt.grad.numpy().reshape(-1)[::7] - fx[tag + "/dx/sample7"]).max() <= 1e-4 * np.abs(fx[tag + "/dx/sample7"]).max()


def _att_case(fx, i):
    act, temp = str(fx["cases"][i]).split(",")
    st = {"att_block.att.weight": torch.from_numpy(fx["w/att.weight"]), "att_block.att.bias": torch.from_numpy(fx["w/att.bias"]),
          "att_block.cla.weight": torch.from_numpy(fx["w/cla.weight"]), "att_block.cla.bias": torch.from_numpy(fx["w/cla.bias"])}
    return act, float(temp), st


def sample_index(numel, cap=2048):
    return np.arange(0, numel, max(1, -(-numel // cap)))


def test_att_block_generic_arguments_vs_reference(golden_dir):
    """AttBlock with the constructor arguments no model of the reference passes ('linear' activation -- its default --, temperature
    != 1) besides ('sigmoid', 1): the oracle against outputs and gradients of the genuine class in float64 (attblock.npz)."""
    fx = np.load(os.path.join(golden_dir, "attblock.npz"))
    assert float(fx["clamped_frac"].min()) > 0.005
    for i in range(len(fx["cases"])):
        act, temp, st = _att_case(fx, i)
        st = {k: v.double().requires_grad_(True) for k, v in st.items()}
        x = torch.from_numpy(fx["x"]).double().requires_grad_(True)
        clip, natt, cla = om.att_block(x, st, activation=act, temperature=temp)
        for got, key in ((clip, "clip"), (natt, "norm_att"), (cla, "cla")):
            np.testing.assert_allclose(got.detach().numpy(), fx["%d/%s" % (i, key)], rtol=2e-6, atol=1e-7, err_msg="%s %s" % (fx["cases"][i], key))
        loss = (clip * torch.from_numpy(fx["g_clip"]).double()).sum() + (natt * torch.from_numpy(fx["g_norm_att"]).double()).sum() + \
               (cla * torch.from_numpy(fx["g_cla"]).double()).sum()
        loss.backward()
        for got, key in ((x.grad, "g_x"), (st["att_block.att.weight"].grad, "g_att.weight"), (st["att_block.cla.bias"].grad, "g_cla.bias")):
            g = got.numpy().reshape(-1)
            np.testing.assert_allclose(g[sample_index(g.size)], fx["%d/%s" % (i, key)], rtol=2e-6, atol=1e-9 + 2e-7 * float(fx["%d/%s/l2" % (i, key)]))

"""graph.GraphedTrainStep: forward + loss + backward replayed as one HIP graph must train exactly like the eager loop body
(reference pytorch/main.py:233-258) -- same losses, same parameters, same SpecAugment random stream, same non-finite guard."""
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


def _batches(n, rows=8, L=32000, seed=11):
    rs = np.random.RandomState(seed)
    lam_rs = np.random.RandomState(1234)
    out = []
    for _ in range(n):
        x = torch.from_numpy((rs.randn(rows, L) * 0.1).astype(np.float32)).cuda()
        y = torch.from_numpy((rs.rand(rows, 17) < 0.2).astype(np.float32)).cuda()
        out.append((x, y, ofe.mixup_lambdas(rows, lam_rs).astype(np.float32)))
    return out


def _run(mt, graphed, batches, eager_steps=2, seed=99):
    from sound_event_detection_dcase2017_task4_amd.graph import GraphedTrainStep
    from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad
    from sound_event_detection_dcase2017_task4_amd.pytorch.losses import clip_bce
    m = _build(mt)
    opt = FusedAdamAmsgrad(m, lr=1e-3)
    step = GraphedTrainStep(m, opt, clip_bce, mixup=True, eager_steps=eager_steps, enabled=graphed)
    torch.manual_seed(seed)                     # SpecAugment stripes come from the global CPU generator, call by call
    losses = [float(step(x, y, lam)) for (x, y, lam) in batches]
    return m, opt, step, losses


@pytest.mark.parametrize("mt", ["Cnn_9layers_FrameAvg", "Cnn_9layers_Gru_FrameAtt"])
def test_graphed_steps_equal_eager_steps(mt):
    batches = _batches(7)
    me, oe, se, le = _run(mt, False, batches)
    mg, og, sg, lg = _run(mt, True, batches)
    assert se.replays == 0 and sg.replays == 5 and sg.graph is not None
    np.testing.assert_allclose(lg, le, rtol=2e-6, atol=0)
    d = float((og.flat - oe.flat).abs().max())
    assert d <= 2e-6, d                          # same kernels on the same data; only atomics' order may differ
    assert og.step_count == oe.step_count == 7
    for (k, a), (_, b) in zip(sorted(mg.state_dict().items()), sorted(me.state_dict().items())):
        if "running" in k or "num_batches" in k:
            np.testing.assert_allclose(a.float().cpu().numpy(), b.float().cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
    # the eager path still works on the same model afterwards (amax pool rows are not shared with the graph)
    x, y, lam = batches[0]
    mg.eval()
    with torch.no_grad():
        a = mg(x[:4], None)["clipwise_output"]
        b = me.eval()(x[:4], None)["clipwise_output"]
    assert float((a - b).abs().max()) <= 2e-6


def test_graph_is_rebuilt_when_the_shape_changes_and_operand_scales_follow_the_data():
    """A replay must recompute every device-side operand scale from THIS batch (the amax rows are re-zeroed inside the graph):
    train on quiet audio, then on audio 1000x louder, and compare with the eager run."""
    batches = _batches(4) + [(x * 1000.0, y, lam) for (x, y, lam) in _batches(3, seed=12)] + _batches(2, rows=4, seed=13)
    me, oe, se, le = _run("Cnn_9layers_FrameAvg", False, batches)
    mg, og, sg, lg = _run("Cnn_9layers_FrameAvg", True, batches)
    np.testing.assert_allclose(lg, le, rtol=5e-6, atol=0)
    assert float((og.flat - oe.flat).abs().max()) <= 5e-6
    assert sg.replays == 5                       # 7 calls at the first shape (2 eager + 5 replays), 2 eager at the second


def test_non_finite_batch_is_refused_under_the_graph():
    from sound_event_detection_dcase2017_task4_amd import ops
    batches = _batches(5)
    m, opt, step, _ = _run("Cnn_9layers_FrameAvg", True, batches[:4])
    before = opt.flat.clone()
    x, y, lam = batches[4]
    x = x.clone()
    x[1, 1000] = float("nan")
    with pytest.raises(ops.NonFiniteOperand):
        step(x, y, lam)
        ops.check_device_errors(synchronize=True)
    assert torch.equal(opt.flat, before)         # the Adam kernel behind the graph skipped the update
    assert opt.step_count == 4
    loss = step(*batches[4])                     # the next clean batch trains again
    assert np.isfinite(float(loss))
    assert opt.step_count == 5 and not torch.equal(opt.flat, before)

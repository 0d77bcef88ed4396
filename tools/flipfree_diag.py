#!/usr/bin/env python
"""Which stage of the BatchNorm backward sets the error of the bn*.weight gradients on the flip-free fixtures?
Runs the fixture step of tests/test_gpu_model.py::test_flip_free_whole_model_gradients_vs_float64_reference with the pooled-window
form of the bn2 + ReLU + pool backward sums on / off (ops.POOL_BWD_WINDOWED) and with gradients as operand pairs on / off
(ops.GRAD_PAIRS), and prints the relative L2 error of every BatchNorm tensor against the float64 reference beside the reference's
own float32 error.    python tools/flipfree_diag.py [model_type]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import model as om                                                        # noqa: E402  (checker data only)
from sound_event_detection_dcase2017_task4_amd import ops                              # noqa: E402
from sound_event_detection_dcase2017_task4_amd.optim import FusedAdamAmsgrad          # noqa: E402
from sound_event_detection_dcase2017_task4_amd.pytorch import models                  # noqa: E402
from sound_event_detection_dcase2017_task4_amd.pytorch.losses import get_loss_func    # noqa: E402
from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup  # noqa: E402
import test_gpu_model as T                                                            # noqa: E402


def run(mt, windowed, pairs):
    ops.POOL_BWD_WINDOWED, ops.GRAD_PAIRS = windowed, pairs
    fx = np.load(os.path.join(REPO, "tests", "golden", mt + "__flipfree.npz"))
    seed = T.SEEDS[mt]
    m = getattr(models, mt)(*T.CTOR)
    m.load_state_dict(om.flipfree_state(mt, seed))
    m = m.to("cuda").train()
    opt = FusedAdamAmsgrad(m, lr=1e-3)
    xw = torch.from_numpy(om.flipfree_waves(2700 + seed, T.FF_ROWS, T.FF_L)).cuda()
    tg = torch.from_numpy(T.targets(2800 + seed, T.FF_ROWS)).cuda()
    lam = torch.from_numpy(fx["ff_lambda"]).cuda()
    o = m(xw, lam, specaug_stripes=fx["ff_stripes"])
    loss = get_loss_func("clip_bce")(o, {"target": do_mixup(tg, lam)})
    opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    out = {}
    for k, p in m.named_parameters():
        if ("ff_g64/" + k) in fx.files and (".bn" in k or k.startswith("bn0")):
            want = fx["ff_g64/" + k].astype(np.float64)
            g = p.grad.detach().double().reshape(-1).cpu().numpy()[T.sample_index(p.numel())]
            out[k] = (float(np.sqrt(((g - want) ** 2).sum() / (want ** 2).sum())), float(fx["ff_ref32err/" + k]))
    return out


def main():
    mt = sys.argv[1] if len(sys.argv) > 1 else "Cnn_9layers_FrameAvg"
    rows = {}
    for windowed, pairs in ((True, True), (False, True), (True, False), (False, False)):
        rows[(windowed, pairs)] = run(mt, windowed, pairs)
    keys = list(rows[(True, True)].keys())
    print("%s: relative L2 error of the BatchNorm gradients vs float64 (flip-free fixture)" % mt)
    print("%-26s %11s | %11s %11s %11s %11s" % ("tensor", "ref fp32", "win+pairs", "full+pairs", "win", "full"))
    for k in keys:
        print("%-26s %11.2e | %11.2e %11.2e %11.2e %11.2e" % (k, rows[(True, True)][k][1], rows[(True, True)][k][0], rows[(False, True)][k][0],
                                                             rows[(True, False)][k][0], rows[(False, False)][k][0]))


if __name__ == "__main__":
    main()

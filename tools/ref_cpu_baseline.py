#!/usr/bin/env python
"""Re-measure the REFERENCE's own CPU path in the build container.

Imports /root/reference/pytorch/{models,losses,pytorch_utils}.py unmodified (with the test-only torchlibrosa stand-in of
tests/golden/, because the real third-party package is not installed) and times BASELINE.json configs[0]:
Cnn_9layers_FrameAvg, batch_size=32, clip_bce, no mixup (SpecAugment on in train mode), Adam(amsgrad), synthetic
10 s / 32 kHz clips -- `steps` optimisation steps, the first one reported separately as warm-up.
The reference's Python cannot travel to the GPU box, so this number is produced HERE and committed:

    python tools/ref_cpu_baseline.py --steps 4 > profiles/r02/ref_cpu_baseline.json
"""
import argparse
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "tests", "golden", "_torchlibrosa_standin"))
sys.path.insert(1, "/root/reference/pytorch")
sys.path.insert(2, "/root/reference/utils")

import numpy as np
import torch
import torch.optim as optim

import models as ref_models            # /root/reference/pytorch/models.py
import losses as ref_losses            # /root/reference/pytorch/losses.py
import pytorch_utils as ref_utils      # /root/reference/pytorch/pytorch_utils.py
import config as ref_config            # /root/reference/utils/config.py


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    m = ref_models.Cnn_9layers_FrameAvg(ref_config.sample_rate, ref_config.window_size, ref_config.hop_size, ref_config.mel_bins,
                                        ref_config.fmin, ref_config.fmax, ref_config.classes_num)
    opt = optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-08, weight_decay=0., amsgrad=True)   # main.py:144-145
    loss_func = ref_losses.get_loss_func('clip_bce')
    rs = np.random.RandomState(1234)
    times = []
    for it in range(args.steps):
        x = (rs.randn(args.batch_size, 320000) * 0.1).astype(np.float32)
        y = (rs.rand(args.batch_size, 17) < 0.2).astype(np.float32)
        t0 = time.time()
        xw = ref_utils.move_data_to_device(x, 'cpu')
        tg = ref_utils.move_data_to_device(y, 'cpu')
        m.train()                                                          # main.py:243-258
        out = m(xw, None)
        loss = loss_func(out, {'target': tg})
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.time() - t0)
    cpu = ""
    try:
        cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    timed = times[1:] if len(times) > 1 else times
    print(json.dumps({"what": "reference CPU path (genuine /root/reference code + torchlibrosa stand-in), BASELINE.json configs[0]",
                      "value": round(args.batch_size * len(timed) / sum(timed), 3), "unit": "clips/s", "kind": "reference",
                      "cores": args.threads, "cpu": cpu, "machine": platform.machine(), "torch": torch.__version__,
                      "batch_size": args.batch_size, "step_seconds": [round(t, 2) for t in times],
                      "warmup_steps": len(times) - len(timed), "loss": float(loss.item())}))


if __name__ == "__main__":
    main()

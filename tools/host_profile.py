#!/usr/bin/env python
"""Where the HOST time of one training step goes (~33 us per launch through Python + ctypes + autograd).

    python tools/host_profile.py [--batch_size 4] [--model_type Cnn_9layers_FrameAvg] [--steps 5] [--top 45]

Runs `steps` (<= 5: below the depth of the pinned upload ring, so the host never waits for the GPU) training steps from a
drained device under cProfile and prints (a) the un-profiled host enqueue time per step, (b) the cumulative / own-time
table.  GPU box only (the product has no CPU path)."""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=4)
    ap.add_argument("--model_type", type=str, default="Cnn_9layers_FrameAvg")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--hip_graph", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = bench.Workload(args.model_type, args.batch_size, True, 0, 1, dev, hip_graph=3 if args.hip_graph else 0)
    for i in range(6):
        w.step(i)
    w.sync()
    print("host enqueue, un-profiled: %.3f ms/step (B=%d, %s%s)" % (w.host_enqueue_ms(args.steps), args.batch_size, args.model_type,
                                                                   ", HIP graph" if args.hip_graph else ""))
    # per-phase host time of un-throttled steps (drained device, <= 5 steps)
    from sound_event_detection_dcase2017_task4_amd import ops
    from sound_event_detection_dcase2017_task4_amd.pytorch.pytorch_utils import do_mixup
    if w.graphed is None:
        for nsteps in (1, 3, 5):
            w.sync()
            ph = [0.0] * 5
            for i in range(nsteps):
                wave, target = w.pool[i % len(w.pool)]
                t0 = time.perf_counter()
                lam = ops.upload_small(w.mixup.get_lambda(w.B2), w.dev, torch.float32)
                t1 = time.perf_counter()
                out = w.model(wave, lam)
                t2 = time.perf_counter()
                loss = w.loss_func(out, {"target": do_mixup(target, lam)})
                w.opt.zero_grad()
                t3 = time.perf_counter()
                loss.backward()
                t4 = time.perf_counter()
                w.opt.step()
                t5 = time.perf_counter()
                for k, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
                    ph[k] += (b - a) * 1e3 / nsteps
            print("%d un-throttled step(s): lambda upload %.3f, forward %.3f, loss + zero_grad %.3f, backward %.3f, optimizer %.3f ms"
                  % ((nsteps,) + tuple(ph)))
        w.sync()
    dt, _, _ = w.run(40, 2)
    print("step time: %.3f ms (40 steps)" % (dt / 40 * 1e3))
    w.sync()
    pr = cProfile.Profile()
    steps = min(args.steps, 5)
    t0 = time.time()
    pr.enable()
    for i in range(steps):
        w.step(i)
    pr.disable()
    print("profiled: %.3f ms/step" % ((time.time() - t0) / steps * 1e3))
    w.sync()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(args.top)
        print(s.getvalue())


if __name__ == "__main__":
    main()

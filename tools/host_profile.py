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

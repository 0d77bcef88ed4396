#!/usr/bin/env python
"""Does it pay to run the MAIN chain of a step (forward, dgrads, HBM passes) on a high-priority HIP stream, so that the side
stream's weight gradients only fill what the chain leaves idle?   python tools/main_priority_probe.py [--batch_size 256]
(The device offers priorities (0, -1): nothing below the default, so the only way to rank the side stream lower is to lift
the main one.)  GPU box only."""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--model_type", type=str, default="Cnn_9layers_FrameAvg")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = bench.Workload(args.model_type, args.batch_size, True, 0, 1, dev)
    print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
    hp = torch.cuda.Stream(priority=-1)
    for rep in range(3):
        dt, loss, _ = w.run(args.steps, 3)
        print("default stream          : %.3f ms/step  loss %.6f" % (dt / args.steps * 1e3, loss))
        hp.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(hp):
            dt, loss, _ = w.run(args.steps, 3)
        torch.cuda.current_stream().wait_stream(hp)
        print("main chain on priority -1: %.3f ms/step  loss %.6f" % (dt / args.steps * 1e3, loss))


if __name__ == "__main__":
    main()

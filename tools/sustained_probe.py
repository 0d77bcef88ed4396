#!/usr/bin/env python
"""Step time over a long run (chunks of 10 steps) with the GPU's clock / power / temperature read between chunks:
does the step rate hold, and if not, is it the device throttling?   python tools/sustained_probe.py [steps] [model]"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True,
                             text=True, timeout=20).stdout.strip().splitlines()
        return " | ".join(out[-1].split(",")[:14]) if out else "?"
    except Exception as e:       # noqa
        return "rocm-smi unavailable: %r" % (e,)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    mt = sys.argv[2] if len(sys.argv) > 2 else "Cnn_9layers_FrameAvg"
    dev = torch.device("cuda", 0)
    w = bench.Workload(mt, 256, True, 0, 1, dev)
    for i in range(3):
        w.step(i)
    torch.cuda.synchronize()
    print("header:", smi())
    i = 3
    for c in range(steps // 10):
        t0 = time.time()
        for _ in range(10):
            w.step(i)
            i += 1
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 10 * 1e3
        print("steps %4d-%4d: %7.2f ms/step  alloc %.1f GB reserved %.1f GB  %s" % (i - 10, i, dt, torch.cuda.memory_allocated() / 1e9,
                                                                                torch.cuda.memory_reserved() / 1e9, smi() if c % 3 == 0 else ""))


if __name__ == "__main__":
    main()

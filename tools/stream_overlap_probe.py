#!/usr/bin/env python
"""Experiment (round 2): do an MFMA-bound kernel and an HBM-bound / another MFMA-bound kernel overlap usefully when issued on
two HIP streams?  Times, for block-3-sized operands (B=256, 250x16, 256->256):
  A  wgrad                         B  dgrad                C  bn_bwd_apply-like elementwise pass (torch add_ on 1 GB)
sequentially on one stream vs concurrently on two.  Printed to stdout; result recorded in DESIGN.md section 5."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_dcase2017_task4_amd import ops


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def main():
    B, H, W, C = 256, 250, 16, 256
    dev = "cuda"
    x = torch.randn(B, H, W, C, device=dev)
    gy = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.02
    e1 = torch.randn(B, H, W, C, device=dev)
    e2 = torch.randn(B, H, W, C, device=dev)
    side = torch.cuda.Stream()

    def wgrad():
        return ops._wgrad(x, gy, B, H, W, C, C)

    def dgrad():
        return ops._conv_fwd_like(gy, w, B, H, W, C, C, dgrad=True, epi=0)

    def elem():
        e1.add_(e2)

    def both(f, g):
        def run():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                r = f()
            s = g()
            torch.cuda.current_stream().wait_stream(side)
            return r, s
        return run

    def seq(f, g):
        def run():
            return f(), g()
        return run

    coef = torch.randn(3, C, device=dev)

    def bn_apply():
        ops._call("sed_bn_bwd_apply", ops._ptr(e1), ops._ptr(e2), B * H * W, C, ops._ptr(coef), None, ops._stream())

    for name, f, g in (("wgrad + dgrad", wgrad, dgrad), ("wgrad + elementwise", wgrad, elem), ("dgrad + elementwise", dgrad, elem),
                       ("wgrad + bn_bwd_apply", wgrad, bn_apply)):
        a, b = timed(f), timed(g)
        s, c = timed(seq(f, g)), timed(both(f, g))
        print("%-22s alone %.3f + %.3f ms; one stream %.3f ms; two streams %.3f ms (%.1f %% of sequential)"
              % (name, a, b, s, c, 100 * c / s))

    # the same pairs with the SHORT kernel on a high-priority stream, the MFMA kernel on the current stream
    express = torch.cuda.Stream(priority=-1)

    def express_pair(f, g, chain=1):
        def run():
            express.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(express):
                for _ in range(chain):
                    g()
            r = f()
            torch.cuda.current_stream().wait_stream(express)
            return r
        return run

    for name, f, g, chain in (("wgrad | express add_", wgrad, elem, 1), ("wgrad | express bn_bwd_apply", wgrad, bn_apply, 1),
                              ("wgrad | express 3 x bn_bwd_apply (chain)", wgrad, bn_apply, 3)):
        a, b = timed(f), timed(g) * chain
        c = timed(express_pair(f, g, chain))
        print("%-42s alone %.3f + %.3f ms = %.3f; MFMA on current + short kernels on a priority -1 stream %.3f ms (%.1f %%)"
              % (name, a, b, a + b, c, 100 * c / (a + b)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Experiment (round 4): does an HBM-bound streaming kernel make progress BESIDE an MFMA kernel when the MFMA kernel leaves
register-file room for it?  The side-stream schedule gains only 1-1.6 % of the step because the weight-gradient
kernel allocates 2 x 248 of a SIMD's 512 VGPRs and the convolution 3 x 168: a 36-53-VGPR streaming wave cannot co-reside and only
gets the slots retiring workgroups free.  Here the SAME convolution runs as MW = 4 (3 waves / SIMD, 504 VGPRs held) and as MW = 2
(the MW = 2 tile of rounds 2-4, selected by SED_SF16_MW2=1 until that switch was removed: 2 waves / SIMD, ~330 held, 180 free) beside bn_bwd_apply / the pool backward apply on a second stream.

    python tools/coresident_overlap_probe.py            (was run twice: with and without SED_SF16_MW2=1; the switch is gone, the record is profiles/r04)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_dcase2017_task4_amd import ops


def timed(fn, n=6):
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
    gy = torch.randn(B, H, W, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.02
    e1 = torch.randn(B, H, W, C, device=dev)
    e2 = torch.randn(B, H, W, C, device=dev)
    coef = torch.randn(3, C, device=dev)
    side = torch.cuda.Stream()
    packs = ops.sf16_packs(w, True)
    ga = ops.amax_of(gy)

    def conv():
        return ops._conv_fwd_like(gy, w, B, H, W, C, C, dgrad=True, epi=0, x_amax=ga, packs=packs)

    def stream_pass():
        ops._call("sed_bn_bwd_apply", ops._ptr(e1), ops._ptr(e2), B * H * W, C, ops._ptr(coef), None, ops._stream())

    def both(reps):
        def run():
            side.wait_stream(torch.cuda.current_stream())
            ops._STREAM_OVERRIDE = side
            try:
                for _ in range(reps):
                    stream_pass()
            finally:
                ops._STREAM_OVERRIDE = None
            r = conv()
            torch.cuda.current_stream().wait_stream(side)
            return r
        return run

    a, b = timed(conv), timed(stream_pass)
    print("SED_SF16_MW2=%s: conv dgrad 256->256 @ 250x16, B=256 alone %.3f ms; bn_bwd_apply (3.1 GB) alone %.3f ms"
          % (os.environ.get("SED_SF16_MW2", "0"), a, b))
    for reps in (1, 2, 3):
        c = timed(both(reps))
        print("  conv || %d x bn_bwd_apply on a second stream: %.3f ms  (sequential %.3f, perfect overlap %.3f)"
              % (reps, c, a + reps * b, max(a, reps * b)))


if __name__ == "__main__":
    main()

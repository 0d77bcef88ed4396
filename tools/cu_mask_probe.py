#!/usr/bin/env python
"""Experiment (round 3): does a CU-MASKED side stream for the weight-gradient kernels let the HBM-bound BatchNorm / pool passes
of the main stream run beside them on the CUs left free?  (Unmasked, a short kernel beside an MFMA kernel only gets the slots
retiring workgroups free: DESIGN.md section 5.)  Block-3-sized operands (B=256, 250x16, 256->256):
    wgrad on a stream restricted to 256 - R CUs  ||  3 x bn_bwd_apply on the main stream,   R in {0, 16, 32, 64}
against the same kernels one after the other on one stream."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_dcase2017_task4_amd import ops

HIP = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(reserved, pattern):
    """A HIP stream whose kernels may use every CU except `reserved` of the 256 (pattern 'spread': every (256/reserved)-th CU;
    'top': the last ones)."""
    bits = [1] * 256
    if reserved:
        if pattern == "spread":
            step = 256 // reserved
            for i in range(reserved):
                bits[i * step + step - 1] = 0
        else:
            for i in range(256 - reserved, 256):
                bits[i] = 0
    words = (ctypes.c_uint32 * 8)()
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = HIP.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask -> %d" % rc)
    return torch.cuda.ExternalStream(s.value)


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
    x = torch.randn(B, H, W, C, device=dev)
    gy = torch.randn(B, H, W, C, device=dev)
    e1 = torch.randn(B, H, W, C, device=dev)
    e2 = torch.randn(B, H, W, C, device=dev)
    coef = torch.randn(3, C, device=dev)
    gam, xam = ops.amax_of(gy), ops.amax_of(x)

    def wgrad():
        return ops._wgrad_sf16(x, gy, B, H, W, C, C, gy_amax=gam, x_amax=xam)

    def elem3():
        for _ in range(3):
            ops._call("sed_bn_bwd_apply", ops._ptr(e1), ops._ptr(e2), B * H * W, C, ops._ptr(coef), None, ops._stream())

    a, b = timed(wgrad), timed(elem3)
    print("alone: wgrad %.3f ms, 3 x bn_bwd_apply %.3f ms, sum %.3f ms" % (a, b, a + b))
    for reserved in (0, 16, 32, 64):
        for pattern in (("spread", "top") if reserved else ("spread",)):
            side = masked_stream(reserved, pattern)

            def both():
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                ops._STREAM_OVERRIDE = side
                try:
                    r = wgrad()
                finally:
                    ops._STREAM_OVERRIDE = None
                elem3()
                main.wait_stream(side)
                return r

            def masked_alone():
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                ops._STREAM_OVERRIDE = side
                try:
                    r = wgrad()
                finally:
                    ops._STREAM_OVERRIDE = None
                main.wait_stream(side)
                return r

            m, c = timed(masked_alone), timed(both)
            print("reserved %3d CUs (%-6s): wgrad alone on the masked stream %.3f ms; wgrad || 3 x bn_bwd_apply %.3f ms = %.1f %% of the "
                  "sequential %.3f ms" % (reserved, pattern, m, c, 100 * c / (a + b), a + b))


if __name__ == "__main__":
    main()

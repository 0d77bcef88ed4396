#!/usr/bin/env python
"""Which cut of the weight-gradient slice reduce for which slice count?  Times sed_test_wgrad_sf16_reduce (test hook) alone on
partials of the production shapes: 512 workgroups' worth (38 MB) at the metric's batch, i.e. slices = 512 / tiles.
    python tools/wgrad_reduce_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_dcase2017_task4_amd import _lib, ops

LAYERS = [(64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 512), (512, 512)]


def main():
    h = _lib.test_hooks()
    am = ops.amax_of(torch.ones(8, device="cuda"))
    for ci, co in LAYERS:
        tiles = (ci // 32) * (co // 64)
        for wgs in (512, 2048):
            ns = max(1, wgs // tiles)
            part = torch.randn((ns, 9, co, ci), device="cuda")
            dw = torch.empty((co, ci, 3, 3), device="cuda")
            outs, row = {}, []
            for v in (1, 2, 3):
                if v == 3 and (ci % 64 or ns > 64):
                    continue
                fn = lambda: _lib.check(h.sed_test_wgrad_sf16_reduce(ops._ptr(part), ns, co, ci, ops._ptr(am), ops._ptr(am), ops._ptr(dw), v,
                                                                     ops._stream()), "reduce")
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(50):
                    fn()
                b.record()
                torch.cuda.synchronize()
                outs[v] = dw.clone()
                row.append("v%d %6.1f us" % (v, a.elapsed_time(b) * 1000 / 50))
            # partial [slice][tap][co][ci] -> OIHW; the kernel also unscales by the operand scales of the two amax vectors (am = 1 -> 2^13 each)
            ref = part.double().sum(0).permute(1, 2, 0).reshape(co, ci, 3, 3) / (2.0 ** 13) ** 2
            err = max(float((o.double() - ref).abs().max() / ref.abs().max()) for o in outs.values())
            print("%4d->%-4d slices %4d (%5.1f MB)  %s   max rel err %.1e" % (ci, co, ns, part.numel() * 4 / 1e6, "  ".join(row), err))


if __name__ == "__main__":
    main()

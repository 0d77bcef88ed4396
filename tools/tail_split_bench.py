#!/usr/bin/env python
"""Per-layer A/B of the TAIL split of the split-f16 convolutions (csrc/conv_sf16.hip: the tiles of the last, partial round of the
chip split their K range so that their shares fill the round) at the metric's batch size.  One process per mode (ops.CONV_TAIL is
read from SED_CONV_TAIL at import; 0 = the library's rule = un-split):
    for m in 0 2 3 4; do SED_CONV_TAIL=$m python tools/tail_split_bench.py --batch 32; done
Prints, per production launch shape of the training step, the plan (ksplit, nfull, workgroups) and the HIP-event time."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_dcase2017_task4_amd import ops, _lib

BLOCKS = [(64, 64, 1001, 64), (64, 128, 500, 32), (128, 256, 250, 16), (256, 512, 125, 8)]      # (Cin, Cout, H, W) of a ConvBlock


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    B = args.batch
    L = _lib.lib()
    mode = os.environ.get("SED_CONV_TAIL", "0 (library rule: un-split)")
    tot = 0.0
    rows = []
    for bi, (c0, c1, H, W) in enumerate(BLOCKS):
        g = torch.Generator(device="cuda").manual_seed(1 + bi)
        shapes = []
        if bi > 0:
            shapes.append(("fwd conv1 epi1+mm ", c0, c1, "f1"))
        shapes += [("fwd conv2 epi1+inT", c1, c1, "f2"), ("dgrad conv2 epi2  ", c1, c1, "d2")]
        if bi > 0:
            shapes.append(("dgrad conv1 epi0  ", c1, c0, "d1"))
        for name, ci, co, kind in shapes:
            x = torch.randn((B, H, W, ci), device="cuda", generator=g)
            w = torch.randn((co, ci, 3, 3), device="cuda", generator=g) * 0.05
            pack = ops.pack_sf16(w)
            xam = ops.amax_of(x)
            st = ops.BnStats(ci, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.1); st.mean.fill_(0.0); st.invstd.fill_(1.0)
            xamT = ops.act_amax_full(x, st)
            P = int(L.sed_conv_sf16_num_parts(B, H, W, co))
            parts = torch.empty((P * 2 * co + P,), device="cuda")
            mm = torch.empty((P, 2, co), device="cuda")
            yprev = torch.randn((B, H, W, co), device="cuda", generator=g)
            pst = ops.BnStats(co, "cuda"); pst.scale.fill_(1.0); pst.shift.fill_(0.1); pst.mean.fill_(0.0); pst.invstd.fill_(1.0)
            if kind == "f1":
                fn = lambda: ops.conv3x3_sf16(x, pack, B, H, W, ci, co, x_amax=xam, epi=1, partials=parts, minmax=mm)
            elif kind == "f2":
                fn = lambda: ops.conv3x3_sf16(x, pack, B, H, W, ci, co, in_st=st, x_amax=xamT, epi=1, partials=parts, minmax=mm)
            elif kind == "d2":
                fn = lambda: ops.conv3x3_sf16(x, pack, B, H, W, ci, co, x_amax=xam, epi=2, partials=parts, yprev=yprev, p_st=pst)
            else:
                fn = lambda: ops.conv3x3_sf16(x, pack, B, H, W, ci, co, x_amax=xam)
            nfull = ctypes.c_int(0)
            ks = int(L.sed_conv_sf16_split_plan(B, H, W, ci, co, ops.CONV_TAIL, ctypes.byref(nfull))) if ops.CONV_SPLITK else 1
            tr = 256 // W
            tiles = B * ((H + tr - 1) // tr) * (co // 64)
            wgs = tiles if ks == 1 else nfull.value + (tiles - nfull.value) * ks
            ms = timeit(fn, args.reps)
            fl = 2.0 * 9 * B * H * W * ci * co
            tot += ms
            rows.append("%4d->%-4d %4dx%-3d %s tiles %5d  ksplit %d nfull %5d -> %5d wgs  %8.4f ms  %6.1f TFLOP/s"
                        % (ci, co, H, W, name, tiles, ks, nfull.value, wgs, ms, fl / ms / 1e9))
    print("SED_CONV_TAIL=%s  batch %d" % (mode, B))
    print("\n".join(rows))
    print("TOTAL %.4f ms" % tot)


if __name__ == "__main__":
    main()

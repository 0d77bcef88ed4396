#!/usr/bin/env python
"""Micro-benchmark of the MFMA conv kernels per layer shape (HIP-event timed), for kernel iteration on the GPU box.
    python tools/conv_bench.py [--batch 64] [--reps 5] [--only igemm|wgrad]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_dcase2017_task4_amd import ops, _lib

LAYERS = [(64, 64, 1001, 64), (64, 128, 500, 32), (128, 128, 500, 32), (128, 256, 250, 16), (256, 256, 250, 16),
          (256, 512, 125, 8), (512, 512, 125, 8)]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def presplit(t):
    """fp32 [..., C] -> the same bytes holding, per 4 channels, (h0 h1 h2 h3 l0 l1 l2 l3) f16 of t * 2^k, amax -> [2^13, 2^14)"""
    import math
    s = 2.0 ** (14 - math.frexp(float(t.abs().max()))[1])
    v = t * s
    hi = v.half()
    lo = (v - hi.float()).half()
    C = t.shape[-1]
    both = torch.cat([hi.view(-1, C // 4, 4), lo.view(-1, C // 4, 4)], -1).contiguous()
    return both.view(torch.float32).view(t.shape)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--layers", type=str, default="", help="comma-separated indices into LAYERS (default: all)")
    ap.add_argument("--presplit", action="store_true", help="experiment (-DSF_EMU_PRESPLIT builds): hand the split-f16 kernels "
                    "operands that are already (hi, lo) f16 pairs, 16 bytes per 4 channels")
    args = ap.parse_args()
    B = args.batch
    tot_ms, tot_fl = {}, {}
    layers = [LAYERS[int(i)] for i in args.layers.split(",")] if args.layers else LAYERS
    for (ci, co, H, W) in layers:
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn((B, H, W, ci), device="cuda", generator=g)
        gy = torch.randn((B, H, W, co), device="cuda", generator=g)
        w = torch.randn((co, ci, 3, 3), device="cuda", generator=g) * 0.05
        wf, wd = ops._pack(w, True, True)
        st = ops.BnStats(ci, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.1); st.mean.fill_(0.0); st.invstd.fill_(1.0)
        sto = ops.BnStats(ci, "cuda"); sto.scale.fill_(1.0); sto.shift.fill_(0.1); sto.mean.fill_(0.0); sto.invstd.fill_(1.0)
        L = _lib.lib()
        M = B * H * W
        fl = 2.0 * 9 * M * ci * co
        part = torch.empty((L.sed_conv_num_parts(M, co), 2, co), device="cuda")
        partb = torch.empty((L.sed_conv_num_parts(M, ci), 2, ci), device="cuda")
        runs = []
        if args.only in ("", "igemm"):
            runs += [("igemm fwd epi1+inT", lambda: ops._conv_igemm(x, wf, B, H, W, ci, co, in_st=st, epi=1, partials=part)),
                     ("igemm fwd epi0    ", lambda: ops._conv_igemm(x, wf, B, H, W, ci, co)),
                     ("igemm fwd epi0+inT", lambda: ops._conv_igemm(x, wf, B, H, W, ci, co, in_st=st)),
                     ("igemm fwd epi1    ", lambda: ops._conv_igemm(x, wf, B, H, W, ci, co, epi=1, partials=part)),
                     ("igemm dgrad epi2  ", lambda: ops._conv_igemm(gy, wd, B, H, W, co, ci, epi=2, partials=partb, yprev=x, p_st=sto))]
        if args.only in ("", "wino2") and L.sed_conv3x3_wino2_supported(H, W, ci, co):
            uf2, ud2 = ops._pack_wino2(w, True, True)
            pw2, _ = ops._wino2_partials(B, H, W, co, "cuda")
            pwb2, _ = ops._wino2_partials(B, H, W, ci, "cuda")
            runs += [("wino2 fwd epi1+inT", lambda: ops._conv_wino2(x, uf2, B, H, W, ci, co, in_st=st, epi=1, partials=pw2)),
                     ("wino2 fwd epi0    ", lambda: ops._conv_wino2(x, uf2, B, H, W, ci, co)),
                     ("wino2 dgrad epi2  ", lambda: ops._conv_wino2(gy, ud2, B, H, W, co, ci, epi=2, partials=pwb2, yprev=x, p_st=sto))]
        if args.only in ("", "wino2", "sf16") and L.sed_conv3x3_sf16_supported(H, W, ci, co):
            wps, wpsd = ops.pack_sf16(w), ops.pack_sf16(w, dgrad=True)
            xam, xamT, gam0 = ops.amax_of(x), ops.act_amax_full(x, st), ops.amax_of(gy)
            xs, gys = (presplit(x), presplit(gy)) if args.presplit else (x, gy)
            Ps = int(L.sed_conv_sf16_num_parts(B, H, W, co))
            ps1 = torch.empty((Ps * 2 * co + Ps,), device="cuda")
            mm1 = torch.empty((Ps, 2, co), device="cuda")
            psb = torch.empty((int(L.sed_conv_sf16_num_parts(B, H, W, ci)) * 2 * ci,), device="cuda")
            runs += [("sf16  fwd epi0    ", lambda: ops.conv3x3_sf16(xs, wps, B, H, W, ci, co, x_amax=xam)),
                     ("sf16  fwd epi0+inT", lambda: ops.conv3x3_sf16(x, wps, B, H, W, ci, co, in_st=st, x_amax=xamT)),
                     ("sf16  fwd epi1+inT", lambda: ops.conv3x3_sf16(x, wps, B, H, W, ci, co, in_st=st, x_amax=xamT, epi=1, partials=ps1)),
                     ("sf16  fwd epi1+mm ", lambda: ops.conv3x3_sf16(xs, wps, B, H, W, ci, co, x_amax=xam, epi=1, partials=ps1, minmax=mm1)),
                     ("sf16  dgrad epi2  ", lambda: ops.conv3x3_sf16(gys, wpsd, B, H, W, co, ci, x_amax=gam0, epi=2, partials=psb, yprev=x, p_st=sto))]
        if args.only in ("", "wgrad", "sf16w") and L.sed_wgrad_sf16_supported(H, W, ci, co):
            gam, xam2, xamT2 = ops.amax_of(gy), ops.amax_of(x), ops.act_amax_full(x, st)
            xs2, gys2 = (presplit(x), presplit(gy)) if args.presplit else (x, gy)
            runs += [("wgrad sf16 +inT   ", lambda: ops._wgrad_sf16(x, gys2, B, H, W, ci, co, in_st=st, gy_amax=gam, x_amax=xamT2)),
                     ("wgrad sf16        ", lambda: ops._wgrad_sf16(xs2, gys2, B, H, W, ci, co, gy_amax=gam, x_amax=xam2))]
        if args.only in ("", "wgrad"):
            runs += [("wgrad +inT        ", lambda: ops._wgrad_direct(x, gy, B, H, W, ci, co, in_st=st)),
                     ("wgrad             ", lambda: ops._wgrad_direct(x, gy, B, H, W, ci, co)),
                     ("wgrad wino2 +inT  ", lambda: ops._wgrad_wino2(x, gy, B, H, W, ci, co, in_st=st)),
                     ("wgrad wino2       ", lambda: ops._wgrad_wino2(x, gy, B, H, W, ci, co))]
        for name, fn in runs:
            ms = timeit(fn, args.reps)
            print("%4d->%-4d %4dx%-3d %s %8.3f ms  %6.1f TFLOP/s" % (ci, co, H, W, name, ms, fl / ms / 1e9))
            tot_ms[name] = tot_ms.get(name, 0) + ms
            tot_fl[name] = tot_fl.get(name, 0) + fl
    for k in tot_ms:
        print("TOTAL %s %8.3f ms  %6.1f TFLOP/s" % (k, tot_ms[k], tot_fl[k] / tot_ms[k] / 1e9))


if __name__ == "__main__":
    main()

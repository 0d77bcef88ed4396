#!/usr/bin/env python
"""Build, check and time the Winograd F(2x2,3x3) split-f16 convolution EXPERIMENT (conv_wsf16.hip) stand-alone on the GPU box:

    python tools/experiments/conv_wsf16/run.py check            # every shape of check_wsf16.SHAPES against float64
    python tools/experiments/conv_wsf16/run.py bench [--batch 32] [--abl N]   # per-layer times beside the shipped direct kernel

The kernel is NOT part of libsed_hip.so (measured slower than the direct split-f16 kernel).  It is
compiled here with hipcc into /tmp and driven through ctypes on raw device pointers; inputs, amax vectors and the direct kernel
come from the product (`ops`)."""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, REPO)
from sound_event_detection_dcase2017_task4_amd import _lib, build, ops  # noqa: E402

LAYERS = [(64, 128, 500, 32), (128, 128, 500, 32), (128, 256, 250, 16), (256, 256, 250, 16), (256, 512, 125, 8), (512, 512, 125, 8)]


def load():
    so, obj = "/tmp/libwsf16_experiment.so", "/tmp/wsf16_experiment.o"
    src = os.path.join(HERE, "conv_wsf16.hip")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
           "-I", build.INCLUDE, "-I", build.CSRC, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    usage, keep = [], False
    for l in r.stderr.splitlines():                    # the resource blocks of the two un-ablated convolution kernels
        if "Function Name" in l:
            keep = ("conv_wsf16_kernel" in l or "conv_wsf16h_kernel" in l) and "Li0E" in l
        if keep and any(k in l for k in ("Function Name", "VGPRs:", "AGPRs:", "ScratchSize", "VGPRs Spill", "LDS Size", "Occupancy")):
            usage.append(l.split("remark: ")[-1].split(" [-Rpass")[0].strip() if "Function Name" not in l else
                         ("conv_wsf16h_kernel<ROT = %s> (v2, eta halves)" % ("true" if "Lb1E" in l else "false") if "wsf16h" in l else
                          "conv_wsf16_kernel<INT = %s> (v1)" % ("true" if "ILb1" in l else "false")))
    subprocess.run([os.environ.get("CXX", "g++"), "-shared", "-fPIC", "-o", so, obj], check=True)
    _lib.lib()                                          # loads torch's HIP runtime RTLD_GLOBAL first
    h = ctypes.CDLL(so)
    h.sed_conv_wsf16_pack_halfs.restype = ctypes.c_long
    h.sed_conv_wsf16_pack_halfs.argtypes = [ctypes.c_int, ctypes.c_int]
    h.sed_conv3x3_wsf16_supported.argtypes = [ctypes.c_int] * 4
    h.sed_pack_conv_weights_wsf16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    h.sed_conv3x3_wsf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 6
    return h, usage


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Wsf(object):
    def __init__(self):
        self.h, self.usage = load()

    def pack(self, w, dgrad=False):
        Cout, Cin = w.shape[0], w.shape[1]
        up = torch.empty((self.h.sed_conv_wsf16_pack_halfs(Cin, Cout),), dtype=torch.float16, device=w.device)
        ws = torch.zeros((65,), dtype=torch.float32, device=w.device)
        rc = self.h.sed_pack_conv_weights_wsf16(ptr(w.contiguous()), Cout, Cin, 1 if dgrad else 0, ptr(up), ptr(ws), stream())
        assert rc == 0, rc
        return up, ws

    def conv(self, x, pack, B, H, W, Cin, Cout, in_st=None, x_amax=None):
        up, ws = pack
        if x_amax is None:
            x_amax = ops.act_amax_full(x, in_st) if in_st is not None else ops.amax_of(x)
        y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
        rc = self.h.sed_conv3x3_wsf16(ptr(x), ptr(up), ptr(ws), ptr(y), B, H, W, Cin, Cout, ptr(in_st.scale) if in_st is not None else None,
                                      ptr(in_st.shift) if in_st is not None else None, ptr(x_amax), None, None, stream())
        assert rc == 0, rc
        return y


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


def bench(args):
    k = Wsf()
    print("# kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage):")
    for l in k.usage:
        print("#   " + l.strip())
    B = args.batch
    tot = {}
    for (ci, co, H, W) in ([LAYERS[int(i)] for i in args.layers.split(",")] if args.layers else LAYERS):
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn((B, H, W, ci), device="cuda", generator=g)
        gy = torch.randn((B, H, W, co), device="cuda", generator=g)
        w = torch.randn((co, ci, 3, 3), device="cuda", generator=g) * 0.05
        st = ops.BnStats(ci, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.1)
        fl = 2.0 * 9 * B * H * W * ci * co
        wps, wpsd = ops.pack_sf16(w), ops.pack_sf16(w, dgrad=True)
        ups, upd = k.pack(w), k.pack(w, dgrad=True)
        xam, xamT, gam = ops.amax_of(x), ops.act_amax_full(x, st), ops.amax_of(gy)
        runs = [("direct sf16 fwd       ", lambda: ops.conv3x3_sf16(x, wps, B, H, W, ci, co, x_amax=xam)),
                ("direct sf16 fwd +inT  ", lambda: ops.conv3x3_sf16(x, wps, B, H, W, ci, co, in_st=st, x_amax=xamT)),
                ("direct sf16 dgrad     ", lambda: ops.conv3x3_sf16(gy, wpsd, B, H, W, co, ci, x_amax=gam)),
                ("winograd sf16 fwd     ", lambda: k.conv(x, ups, B, H, W, ci, co, x_amax=xam)),
                ("winograd sf16 fwd +inT", lambda: k.conv(x, ups, B, H, W, ci, co, in_st=st, x_amax=xamT)),
                ("winograd sf16 dgrad   ", lambda: k.conv(gy, upd, B, H, W, co, ci, x_amax=gam))]
        for name, fn in runs:
            if args.only_wino and not name.startswith("winograd sf16 dgrad"):
                continue
            ms = timeit(fn, args.reps)
            print("%4d->%-4d %4dx%-3d B=%-3d %s %8.3f ms  %6.1f TFLOP/s (algorithmic)" % (ci, co, H, W, B, name, ms, fl / ms / 1e9))
            t = tot.setdefault(name, [0.0, 0.0]); t[0] += ms; t[1] += fl
    for name, (ms, fl) in tot.items():
        print("TOTAL %s %8.3f ms  %6.1f TFLOP/s" % (name, ms, fl / ms / 1e9))


def check(args):
    import check_wsf16
    k = Wsf()
    worst = 0.0
    for (B, H, W, Cin, Cout, inT) in check_wsf16.SHAPES:
        rel, mx = check_wsf16.forward_case(k, B, H, W, Cin, Cout, inT)
        worst = max(worst, rel)
        print("forward %-34s relative L2 %.2e  max %.2e of the output max" % ((B, H, W, Cin, Cout, inT), rel, mx))
        assert rel < 3e-6 and mx < 3e-5
    for (B, H, W, Cin, Cout) in check_wsf16.DGRAD_SHAPES:
        rel, mx = check_wsf16.dgrad_case(k, B, H, W, Cin, Cout)
        worst = max(worst, rel)
        print("dgrad   %-34s relative L2 %.2e  max %.2e" % ((B, H, W, Cin, Cout), rel, mx))
        assert rel < 3e-6 and mx < 3e-5
    for mag in (1e-6, 1e-3, 1.0, 1e3, 1e6):
        rel, mx = check_wsf16.magnitude_case(k, mag)
        print("magnitude %-8g relative L2 %.2e  max %.2e" % (mag, rel, mx))
        assert rel < 3e-6 and mx < 3e-5
    print("all green; worst relative L2 %.2e (the direct split-f16 kernel is held to 1e-6, the fp32 Winograd kernels measure 5e-7)" % worst)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["check", "bench"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--layers", type=str, default="", help="comma-separated indices into LAYERS")
    ap.add_argument("--only_wino", action="store_true")
    ap.add_argument("--abl", type=int, default=0, help="timing ablation (results wrong): 1 no DMA after the prologue, 2 no operand split, "
                                                      "4 no MFMA; sums allowed (3, 6, 7)")
    a = ap.parse_args()
    if a.abl:
        os.environ["WSF_ABL"] = str(a.abl)
    sys.path.insert(0, HERE)
    (check if a.mode == "check" else bench)(a)

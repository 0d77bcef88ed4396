#!/usr/bin/env python
"""Runner of tools/pk_f32_beside_mfma_probe.hip: the probe kernel on the current stream, an MFMA kernel of this package on a second
stream of the SAME process.   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/pk_f32_beside_mfma_probe.hip -o /tmp/pkprobe.so
python tools/pk_f32_beside_mfma_probe.py [launches]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_dcase2017_task4_amd import ops

FORMS = ["plain fma", "plain add", "add src1 swapped", "add neg_hi", "add src1 swapped + neg_hi", "mul src0.lo broadcast",
         "fma src1.lo/src2.lo broadcast", "mul src1 swapped", "fma src1 swapped", "add src0 swapped", "fma src1.hi broadcast", "fma src0 swapped + src1.hi broadcast + neg",
         "mul src1 = SGPR pair", "fma src0 swapped + SGPR src1.hi broadcast + neg"]


def main(n):
    lib = ctypes.CDLL("/tmp/pkprobe.so")
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    B, H, W, C = 4, 101, 64, 64
    xc = torch.randn((B, H, W, C), device="cuda", generator=g)
    gy = torch.randn((B, H, W, C), device="cuda", generator=g)
    w2 = torch.randn((C, C, 3, 3), device="cuda", generator=g) * 0.05
    pk, xam, gam = ops.pack_sf16(w2), ops.amax_of(xc), ops.amax_of(gy)
    wf, _ = ops._pack(w2, True, True)
    aggrs = {"nothing": None,
             "conv_sf16 (v_mfma_f32_32x32x16_f16)": lambda: ops.conv3x3_sf16(xc, pk, B, H, W, C, C, x_amax=xam),
             "wgrad_sf16 (v_mfma_f32_32x32x16_f16)": lambda: ops._wgrad_sf16(xc, gy, B, H, W, C, C, gy_amax=gam, x_amax=xam),
             "conv_igemm (v_mfma_f32_32x32x2_f32)": lambda: ops._conv_igemm(xc, wf, B, H, W, C, C)}
    sb = torch.cuda.Stream()
    bad = torch.zeros((16,), dtype=torch.int32, device="cuda")
    for name, ag in aggrs.items():
        bad.zero_()
        for it in range(n):
            if ag is not None:
                with torch.cuda.stream(sb):
                    for _ in range(6):
                        ag()
            lib.pk_probe_launch(ctypes.c_void_p(bad.data_ptr()), 512, 100, it * 7919, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        print("beside %-38s (%d evaluations per form): %s" % (name, n * 512 * 256 * 100, ", ".join(
            "%s %d" % (f, int(bad[k])) for k, f in enumerate(FORMS))), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3000)

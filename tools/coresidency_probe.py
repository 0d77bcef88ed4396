#!/usr/bin/env python
"""Is a kernel still bit-reproducible while a split-f16 (f16 MFMA) kernel of this package runs on a SECOND stream of the same
process?  Victims: the package's log-mel kernel, rocFFT through torch.stft, and a few stock PyTorch kernels.
Round-3 finding (MI355X): kernels that use packed-fp32 instructions with op_sel[src1] = 1, op_sel[src0] = 0 -- complex
arithmetic of FFT kernels -- are NOT (tools/pk_f32_beside_mfma_probe.hip isolates the instruction forms); since the operand
order in csrc/logmel.hip was changed the log-mel kernel is.   python tools/coresidency_probe.py [launches]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from sound_event_detection_dcase2017_task4_amd import ops                      # noqa: E402
from sound_event_detection_dcase2017_task4_amd.pytorch import models           # noqa: E402


def main(n):
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    m = models.Cnn_9layers_FrameAvg(32000, 1024, 320, 64, 50, 14000, 17).to("cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.from_numpy((np.random.RandomState(0).randn(8, 32000) * 0.1).astype(np.float32)).cuda()
    B, H, W, C = 4, 101, 64, 64
    xc = torch.randn((B, H, W, C), device="cuda", generator=g)
    gy = torch.randn((B, H, W, C), device="cuda", generator=g)
    w2 = torch.randn((C, C, 3, 3), device="cuda", generator=g) * 0.05
    pk, xam, gam = ops.pack_sf16(w2), ops.amax_of(xc), ops.amax_of(gy)
    wf, _ = ops._pack(w2, True, True)
    a = torch.randn((1 << 22,), device="cuda", generator=g)
    b = torch.randn((1 << 22,), device="cuda", generator=g)
    m1 = torch.randn((1024, 1024), device="cuda", generator=g)
    win = torch.hann_window(1024, device="cuda")
    victims = {
        "log-mel kernel (this package)": lambda: m.extract_logmel(x),
        "torch.stft (rocFFT)": lambda: torch.view_as_real(torch.stft(x, 1024, 320, window=win, return_complex=True)),
        "torch add fp32 (what a reduction does)": lambda: a + b,
        "torch mm fp32": lambda: m1 @ m1,
        "torch softmax": lambda: torch.softmax(m1, 1),
        "conv_sf16 itself": lambda: ops.conv3x3_sf16(xc, pk, B, H, W, C, C, x_amax=xam),
    }
    aggressors = {
        "conv_sf16 (f16 MFMA)": lambda: ops.conv3x3_sf16(xc, pk, B, H, W, C, C, x_amax=xam),
        "wgrad_sf16 (f16 MFMA)": lambda: ops._wgrad_sf16(xc, gy, B, H, W, C, C, gy_amax=gam, x_amax=xam),
        "conv_igemm (fp32 MFMA)": lambda: ops._conv_igemm(xc, wf, B, H, W, C, C),
    }
    sb = torch.cuda.Stream()
    for an, ag in aggressors.items():
        for vn, fn in victims.items():
            ref = fn().clone()
            torch.cuda.synchronize()
            bad = 0
            for _ in range(n):
                with torch.cuda.stream(sb):
                    for _ in range(6):
                        ag()
                if not torch.equal(fn(), ref):
                    bad += 1
            torch.cuda.synchronize()
            print("beside %-24s victim %-40s: %5d of %d launches differ" % (an, vn, bad, n), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1500)

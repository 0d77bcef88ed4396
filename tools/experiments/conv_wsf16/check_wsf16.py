"""Cases of `run.py check`: the Winograd F(2x2,3x3) split-f16 experiment kernel against a float64 convolution -- forward (plain
and with the fused relu(scale*x+shift) operand transform), dgrad (the transposed convolution), ragged image bottoms, every
supported width, magnitudes far from 1.  Winograd carries more rounding noise than a direct sum: gate 3e-6 relative L2."""
import numpy as np
import torch
import torch.nn.functional as F

from sound_event_detection_dcase2017_task4_amd import ops

SHAPES = [(2, 37, 64, 64, 128, False), (3, 21, 32, 128, 128, True), (2, 250, 16, 128, 256, False), (3, 13, 8, 256, 512, True),
          (1, 1, 8, 16, 128, False), (2, 5, 64, 32, 256, True), (2, 9, 64, 64, 64, True), (2, 33, 32, 128, 64, False),
          (1, 125, 8, 512, 512, False), (2, 3, 16, 48, 32, False), (5, 65, 8, 64, 256, True)]
DGRAD_SHAPES = [(2, 37, 64, 64, 128), (3, 21, 32, 128, 128), (2, 50, 16, 128, 256), (3, 13, 8, 256, 512)]


def _ref(x_nhwc, w, scale=None, shift=None):
    x = x_nhwc.double()
    if scale is not None:
        x = torch.relu(x * scale.double() + shift.double())
    return F.conv2d(x.permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1).contiguous()


def _err(y, want):
    d = y.double().cpu() - want
    return float(d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()), float(d.abs().max() / want.abs().max())


def forward_case(k, B, H, W, Cin, Cout, inT):
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn((B, H, W, Cin), generator=g) * 1.5
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * float(np.sqrt(6.0 / (9 * Cin + 9 * Cout)))
    st = scale = shift = None
    if inT:
        scale = torch.rand(Cin, generator=g) + 0.5
        shift = torch.randn(Cin, generator=g) * 0.3
        st = ops.BnStats(Cin, "cuda")
        st.scale.copy_(scale); st.shift.copy_(shift)
    y = k.conv(x.cuda(), k.pack(w.cuda()), B, H, W, Cin, Cout, in_st=st)
    torch.cuda.synchronize()
    return _err(y, _ref(x, w, scale, shift))


def dgrad_case(k, B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(B * 77 + W)
    gy = torch.randn((B, H, W, Cout), generator=g) * 1e-4          # gradient-sized values
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * float(np.sqrt(6.0 / (9 * Cin + 9 * Cout)))
    xr = torch.zeros((B, Cin, H, W), dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, w.double(), padding=1).backward(gy.double().permute(0, 3, 1, 2))
    gx = k.conv(gy.cuda(), k.pack(w.cuda(), dgrad=True), B, H, W, Cout, Cin)
    torch.cuda.synchronize()
    return _err(gx, xr.grad.permute(0, 2, 3, 1).contiguous())


def magnitude_case(k, mag):
    B, H, W, Cin, Cout = 2, 19, 16, 128, 128
    g = torch.Generator().manual_seed(3)
    x = torch.randn((B, H, W, Cin), generator=g) * mag
    x[0, 3, 5, 7] = 40.0 * mag                                       # an outlier: the 4x4 patches around it sum it up to 4 times
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * 0.03 / mag
    y = k.conv(x.cuda(), k.pack(w.cuda()), B, H, W, Cin, Cout)
    torch.cuda.synchronize()
    return _err(y, _ref(x, w))

"""EXPERIMENTAL split-f16 3x3 convolution (csrc/conv_sf16.hip) against a float64 convolution: the three-term split
must be as accurate as the fp32 kernels (tools/split_f16_study.py: 1.5e-7 relative L2)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x_nhwc, w, scale=None, shift=None):
    x = x_nhwc.double()
    if scale is not None:
        x = torch.relu(x * scale.double() + shift.double())
    y = F.conv2d(x.permute(0, 3, 1, 2), w.double(), padding=1)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,W,Cin,Cout,inT", [(2, 37, 64, 64, 128, False), (3, 21, 32, 128, 128, True),
                                               (2, 250, 16, 128, 256, False), (3, 13, 8, 256, 512, True),
                                               (1, 1, 8, 16, 128, False), (2, 5, 64, 32, 256, True)])
def test_sf16_conv_matches_float64(B, H, W, Cin, Cout, inT):
    from sound_event_detection_dcase2017_task4_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn((B, H, W, Cin), generator=g) * 1.5
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * float(np.sqrt(6.0 / (9 * Cin + 9 * Cout)))
    st = None
    scale = shift = None
    if inT:
        scale = torch.rand(Cin, generator=g) + 0.5
        shift = torch.randn(Cin, generator=g) * 0.3
        st = ops.BnStats(Cin, "cuda")
        st.scale.copy_(scale); st.shift.copy_(shift)
    want = _ref(x, w, scale, shift)
    wp = ops.pack_sf16(w.cuda())
    y = ops.conv3x3_sf16(x.cuda(), wp, B, H, W, Cin, Cout, in_st=st)
    torch.cuda.synchronize()
    d = y.double().cpu() - want
    rel = float(d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    mx = float(d.abs().max() / want.abs().max())
    print("sf16 %s: relative L2 %.2e, max %.2e of the output max" % ((B, H, W, Cin, Cout, inT), rel, mx))
    assert rel < 1e-6 and mx < 1e-5      # fp32 accumulation over K = 9*Cin terms: same as a direct fp32 convolution


def test_sf16_dgrad_operand_matches_conv_transpose():
    from sound_event_detection_dcase2017_task4_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 2, 19, 32, 128, 64            # w: (Cout=64, Cin=128); dgrad maps 64 -> 128 channels
    gy = torch.randn((B, H, W, Cout), generator=g)
    w = (torch.rand((Cout, Cin, 3, 3), generator=g) * 2 - 1) * 0.03
    want = F.conv_transpose2d(gy.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1)
    wp = ops.pack_sf16(w.cuda(), dgrad=True)
    gx = ops.conv3x3_sf16(gy.cuda(), wp, B, H, W, Cout, Cin)
    d = gx.double().cpu() - want
    assert float(d.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()) < 1e-6

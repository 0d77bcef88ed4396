"""Precision of ops.ConvBlockFn (forward + all gradients) against the float64 oracle, next to the float32 CPU oracle
(= torch's own fp32 arithmetic).   python tools/convblock_precision.py [USE_WINOGRAD]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model as om
from sound_event_detection_dcase2017_task4_amd import ops
if len(sys.argv) > 1:
    ops.USE_WINOGRAD = int(sys.argv[1])
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
nchw = lambda t: t.permute(0, 3, 1, 2).contiguous()
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()
for (B, Cin, Cout, H, W, ph, pw, gmode) in ((16, 256, 512, 25, 8, 1, 8, "rand"), (16, 256, 512, 25, 8, 1, 8, "const"),
                                            (16, 128, 256, 50, 16, 2, 2, "rand"), (16, 64, 128, 100, 32, 2, 2, "rand")):
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.relu(torch.randn(B, Cin, H, W, generator=g)) * 0.7            # post-ReLU-like, non-zero channel means
    st = {}
    for i, (ci, co) in enumerate(((Cin, Cout), (Cout, Cout)), start=1):
        st["cb.conv%d.weight" % i] = torch.randn(co, ci, 3, 3, generator=g) * (1.5 / np.sqrt(9 * ci))
        st["cb.bn%d.weight" % i] = 1 + 0.2 * torch.randn(co, generator=g)
        st["cb.bn%d.bias" % i] = 0.2 * torch.randn(co, generator=g)
        st["cb.bn%d.running_mean" % i] = 0.1 * torch.randn(co, generator=g)
        st["cb.bn%d.running_var" % i] = 0.5 + torch.rand(co, generator=g)
        st["cb.bn%d.num_batches_tracked" % i] = torch.tensor(0)
    names = ["cb.conv1.weight", "cb.bn1.weight", "cb.bn1.bias", "cb.conv2.weight", "cb.bn2.weight", "cb.bn2.bias"]

    def run(dtype):
        s = {k: (v.to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in st.items()}
        for n in names:
            s[n].requires_grad_(True)
        xr = x.to(dtype).clone().requires_grad_(True)
        if (ph, pw) == (1, 8):
            ref = om.conv_block(xr, s, "cb", (1, 1), True, True).mean(dim=3, keepdim=True)
        else:
            ref = om.conv_block(xr, s, "cb", (ph, pw), True, True)
        return ref, xr, s
    ref64, x64, s64 = run(torch.float64)
    gg = torch.Generator().manual_seed(5)
    if gmode == "rand":
        gout = torch.randn(ref64.shape, generator=gg)
    else:                                            # clip-level gradient: constant over time (FrameAvg head)
        gout = torch.randn(ref64.shape[0], ref64.shape[1], 1, 1, generator=gg).expand(ref64.shape).contiguous()
    ref64.backward(gout.double())
    ref32, x32, s32 = run(torch.float32)
    ref32.backward(gout)
    dev = {k: v.detach().clone().cuda() for k, v in st.items()}
    xg = nhwc(x).cuda().requires_grad_(True)
    params = [dev["cb.conv1.weight"], dev["cb.bn1.weight"], dev["cb.bn1.bias"], dev["cb.bn1.running_mean"],
              dev["cb.bn1.running_var"], dev["cb.conv2.weight"], dev["cb.bn2.weight"], dev["cb.bn2.bias"],
              dev["cb.bn2.running_mean"], dev["cb.bn2.running_var"]]
    for i in (0, 1, 2, 5, 6, 7):
        params[i].requires_grad_(True)
    out, _ = ops.ConvBlockFn.apply(xg, *params, True, ph, pw)
    out.backward(nhwc(gout).cuda())
    print("== B%d %d->%d %dx%d pool(%d,%d) gout=%s  USE_WINOGRAD=%d" % (B, Cin, Cout, H, W, ph, pw, gmode, ops.USE_WINOGRAD))
    print("  %-18s hip %.2e   cpu-fp32 %.2e" % ("out", rel(nchw(out.detach()).cpu(), ref64.detach()), rel(ref32.detach(), ref64.detach())))
    print("  %-18s hip %.2e   cpu-fp32 %.2e" % ("dx", rel(nchw(xg.grad).cpu(), x64.grad), rel(x32.grad, x64.grad)))
    for p, n in zip([params[i] for i in (0, 1, 2, 5, 6, 7)], names):
        print("  %-18s hip %.2e   cpu-fp32 %.2e" % (n, rel(p.grad.cpu(), s64[n].grad), rel(s32[n].grad, s64[n].grad)))

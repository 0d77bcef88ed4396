"""Kernel time of the attention core (sed_mha_fwd / sed_mha_bwd: HIP events around the C-ABI calls), with and without the
attention-dropout mask.    python tools/mha_time.py [B] [T]      # GPU box"""
import sys, torch
sys.path.insert(0, '.')
from sound_event_detection_dcase2017_task4_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 125
torch.manual_seed(0)
M = B * T
q, k, v, go = (torch.randn(M, 512, device="cuda") for _ in range(4))
o = torch.empty_like(q); gq = torch.empty_like(q); gk = torch.empty_like(q); gv = torch.empty_like(q)
stats = torch.empty((B, 8, T, 4), device="cuda")
keep = (torch.rand(8 * B, T, T, device="cuda") >= 0.1)
P, S = ops._ptr, ops._stream
kbits = torch.empty((ops._lib.lib().sed_mha_mask_words(B, T),), dtype=torch.int32, device="cuda")
for name, ka in (("train (mask)", keep), ("eval (no mask)", None)):
    def fwd():
        ops._call("sed_mha_fwd", P(q), P(k), P(v), P(ka), 0.1, B, T, P(o), P(stats), P(kbits), S())
    def bwd():
        ops._call("sed_mha_bwd", P(q), P(k), P(v), P(o), P(go), P(ka), 0.1, B, T, P(stats), P(gq), P(gk), P(gv), P(kbits), S())
    for f, tag in ((fwd, "fwd"), (bwd, "bwd (q side + key side)")):
        for _ in range(3):
            f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(10):
            f()
        b.record()
        torch.cuda.synchronize()
        print("B=%d T=%d %-16s %-24s %.1f us" % (B, T, name, tag, a.elapsed_time(b) * 100))

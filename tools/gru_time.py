"""Kernel time of the two GRU recurrences (HIP events around the C-ABI calls; includes the sentinel memsets and the check kernel).
    python tools/gru_time.py [B ...]      # GPU box"""
import sys, torch
sys.path.insert(0, '.')
from sound_event_detection_dcase2017_task4_amd import ops
torch.manual_seed(0)
import os
if os.environ.get('AGENT') == '1':
    ops._lib.test_hooks().sed_gru_force_agent_scope(1)
T = 125
gru = torch.nn.GRU(512, 256, num_layers=1, bias=True, batch_first=True, bidirectional=True).cuda()
names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
for Bt in [int(a) for a in sys.argv[1:]] or [256, 32]:
    xt = torch.randn(Bt, T, 512, device="cuda")
    gt = torch.randn(Bt, T, 512, device="cuda")
    def once():
        ps = [getattr(gru, n).detach().clone().requires_grad_(True) for n in names]
        xd = xt.clone().requires_grad_(True)
        ops.GruFn.apply(xd, *ps).backward(gt)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    ops.TIMING, ops.TIMING_ONLY = {}, None
    for _ in range(10):
        once()
    torch.cuda.synchronize()
    tm, ops.TIMING = ops.TIMING, None
    for k, v in sorted(tm.items()):
        if "gru" in k:
            ms = [a.elapsed_time(b) for a, b, _ in v]
            print("B=%d  %-28s %.3f ms avg over %d launches (min %.3f)" % (Bt, k, sum(ms) / len(ms), len(ms), min(ms)))

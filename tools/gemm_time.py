"""Time the split-f16 dense GEMMs of the GRU / Transformer heads on their production shapes (HIP events, ops.TIMING).
    python tools/gemm_time.py        # GPU box"""
import sys, torch
sys.path.insert(0, '.')
from sound_event_detection_dcase2017_task4_amd import ops
torch.manual_seed(0)
M = 32000
shapes_tn = [(512, 512), (1536, 512), (768, 256)]          # (N = gy columns, K = x columns): dW of MultiHead, GRU W_ih, GRU W_hh
shapes_nt = [(512, 512), (1536, 512), (512, 1536)]          # (N, K): MultiHead projections, GRU gi, GRU dx
def timed(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for N, K in shapes_tn:
    x = torch.randn(M, K, device="cuda"); gy = torch.randn(M, N, device="cuda")
    xa, ga = ops.amax_of(x), ops.amax_of(gy)
    us = timed(lambda: ops.gemm_tn(x, gy, x_amax=xa, gy_amax=ga))
    print("TN  dw[%4d][%4d] over M=%d: %7.1f us  %6.1f TFLOP/s algorithmic" % (N, K, M, us, 2.0 * M * N * K / us / 1e6))
for N, K in shapes_nt:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; bias = torch.zeros(N, device="cuda")
    pk = ops.gemm_pack_sf16(w)
    xa = ops.amax_of(x)
    us = timed(lambda: ops.gemm_nt_sf16(x, pk, N, bias, x_amax=xa))
    print("NT  y[%d][%4d] = x[.][%4d] w^T: %7.1f us  %6.1f TFLOP/s algorithmic" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))

#!/usr/bin/env python
"""Feasibility study for a SPLIT-f16 convolution path (numpy, CPU; no kernel exists yet -- DESIGN.md section 9).

An fp32 value x is written as hi + lo with hi = f16(s*x), lo = f16(s*x - hi) (s = a power of two per tensor that puts
the tensor's amax near 2^8, so that lo stays in f16's NORMAL range); a product a*b is taken as
hi_a*hi_b + hi_a*lo_b + lo_a*hi_b -- three f16 MFMAs (f16 x f16 is exact in the fp32 accumulator) instead of one fp32
MFMA, at 16x the MFMA rate: 3/16 of the issue time for the same (direct-convolution) MAC count, or 0.42x of the fused
Winograd F(2x2,3x3) kernels' MFMA time.  Question: is the result as good as the fp32 kernels'?  This script measures the
error of a 3x3 convolution layer (im2col GEMM, fp32 accumulation emulated by float32 sums in blocks) against float64 for

    fp32 direct | fp32 Winograd F(2x2,3x3) (the shipped kernels' arithmetic) | f16 plain | f16 weights split (2 terms) |
    f16 both split (3 terms) | f16 both split (4 terms)

on activations / weights / gradients with the statistics of this model (post-ReLU activations, Xavier weights, 1e-6-scale
gradients)."""
import numpy as np


def split(x, terms=2):
    amax = np.abs(x).max()
    s = 2.0 ** np.floor(8 - np.log2(max(amax, 1e-30)))
    xs = (x.astype(np.float64) * s)
    hi = xs.astype(np.float16)
    parts = [hi]
    if terms >= 2:
        lo = (xs - hi.astype(np.float64)).astype(np.float16)
        parts.append(lo)
    return [p.astype(np.float32) for p in parts], s


def gemm32(a, b, kblock=64):
    """fp32 GEMM with fp32 accumulation over K in blocks (the MFMA accumulates a K-step at a time)."""
    out = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(0, a.shape[1], kblock):
        out += a[:, k:k + kblock] @ b[k:k + kblock]
    return out


def study(name, A, Bm):
    truth = A.astype(np.float64) @ Bm.astype(np.float64)
    nrm = np.sqrt((truth ** 2).mean())

    def err(y):
        d = y.astype(np.float64) - truth
        return np.sqrt((d ** 2).mean()) / nrm, np.abs(d).max() / nrm

    res = {"fp32": err(gemm32(A, Bm))}
    (ah, al), sa = split(A)
    (bh, bl), sb = split(Bm)
    inv = np.float32(1.0 / (sa * sb))
    res["f16 plain"] = err(gemm32(ah, bh) * inv)
    res["f16 B split (2 terms)"] = err((gemm32(ah, bh) + gemm32(ah, bl)) * inv)
    res["f16 both split (3 terms)"] = err((gemm32(ah, bh) + gemm32(ah, bl) + gemm32(al, bh)) * inv)
    res["f16 both split (4 terms)"] = err((gemm32(ah, bh) + gemm32(ah, bl) + gemm32(al, bh) + gemm32(al, bl)) * inv)
    print("== %s: A %s, B %s" % (name, A.shape, Bm.shape))
    for k, (l2, mx) in res.items():
        print("   %-26s relative L2 %.2e   max %.2e" % (k, l2, mx))
    return res


def winograd_row(name, x, w):
    """2-D Winograd F(2x2,3x3) in float32 vs direct float32 vs float64, one (H, W, Cin) image, Cout filters."""
    H, W, C = x.shape
    N = w.shape[0]
    xp = np.zeros((H + 2, W + 2, C), x.dtype); xp[1:-1, 1:-1] = x
    truth = np.zeros((H, W, N))
    d32 = np.zeros((H, W, N), np.float32)
    for ky in range(3):
        for kx in range(3):
            truth += xp[ky:ky + H, kx:kx + W].astype(np.float64) @ w[:, :, ky, kx].T.astype(np.float64)
            d32 += xp[ky:ky + H, kx:kx + W] @ w[:, :, ky, kx].T
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
    U = np.einsum("ai,ncij,bj->abnc", G, w.astype(np.float32), G).astype(np.float32)
    y = np.zeros((H, W, N), np.float32)
    for th in range(H // 2):
        for tw in range(W // 2):
            d = xp[2 * th:2 * th + 4, 2 * tw:2 * tw + 4].astype(np.float32)          # (4,4,C)
            V = np.einsum("ai,ijc,bj->abc", Bt, d, Bt).astype(np.float32)
            M = np.einsum("abc,abnc->abn", V, U).astype(np.float32)
            y[2 * th:2 * th + 2, 2 * tw:2 * tw + 2] = np.einsum("ia,abn,jb->ijn", At, M, At)
    nrm = np.sqrt((truth ** 2).mean())
    print("== %s" % name)
    for k, v in (("fp32 direct", d32), ("fp32 Winograd F(2x2,3x3)", y)):
        dd = v.astype(np.float64) - truth
        print("   %-26s relative L2 %.2e   max %.2e" % (k, np.sqrt((dd ** 2).mean()) / nrm, np.abs(dd).max() / nrm))


def main():
    rs = np.random.RandomState(0)
    C, N, P = 128, 128, 4096                      # im2col: K = 9*C
    act = np.maximum(rs.randn(P, 9 * C) * 1.0 + 0.2, 0).astype(np.float32)               # post-BN-ReLU activations
    wgt = (rs.uniform(-1, 1, (9 * C, N)) * np.sqrt(6.0 / (9 * C + 9 * N))).astype(np.float32)   # Xavier uniform
    study("forward  (activations x weights)", act, wgt)
    gy = (rs.randn(P, 9 * N) * 1e-6 * np.exp(rs.randn(P, 1))).astype(np.float32)          # gradients: tiny, heavy-tailed rows
    study("dgrad    (gradients x weights)", gy, wgt.T.copy().reshape(9 * N, C) if False else (rs.uniform(-1, 1, (9 * N, C)) * 0.02).astype(np.float32))
    study("wgrad    (activations^T x gradients, K = pixels)", act[:, :C].T.copy(), (rs.randn(P, N) * 1e-6 * np.exp(rs.randn(P, 1))).astype(np.float32))
    x = np.maximum(rs.randn(16, 16, 64) + 0.2, 0).astype(np.float32)
    w = (rs.uniform(-1, 1, (64, 64, 3, 3)) * np.sqrt(6.0 / (9 * 128))).astype(np.float32)
    winograd_row("reference points: one 16x16x64 image, 64 filters", x, w)


if __name__ == "__main__":
    main()

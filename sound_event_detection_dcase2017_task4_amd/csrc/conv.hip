// 3x3 convolution (stride 1, pad 1, no bias) for the ConvBlocks of reference pytorch/models.py:72-115, NHWC.
//
//  * conv_igemm_kernel  : implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32; exact fp32, needed for the 1e-4
//    parity budget).  M = B*H*W pixels, N = Cout, K = 9*Cin.  Used for forward AND dgrad (dgrad = the same
//    kernel over the tap-flipped, transposed weight pack).  Optional fusions:
//      - input operand transform  a = relu(scale*y + shift)  (the previous BN+ReLU, never materialised);
//      - epilogue 1: per-channel BatchNorm partial statistics (sum, M2) of the raw conv output;
//      - epilogue 2: ReLU mask of the previous BN + BN-backward partial sums (sum dy, sum dy*xhat);
//      - epilogue 3: bias add (NTAPS=1: plain NT GEMM for the GRU / dense heads).
//  * wgrad_kernel       : dW[tap][co][ci] = sum_p gy[p][co] * a[p+tap][ci], fp32 MFMA, split over pixel slices.
//  * conv1 (Cin = 1)    : K = 9 is HBM-bound (AI 4.4 flop/B): direct kernels, no MFMA.
#include "common.h"
#include "sed_hip.h"
#include <stdlib.h>
SED_OBJECT_FLAGS(conv)

namespace {

constexpr int LDS_STRIDE = 36;   // padded layout: 32 floats + 4 pad: ds_read_b128 of 16 distinct rows -> 16 distinct 16-B slots

// LDS tile addressing.  SWZ = false: rows padded to 36 floats (conflict-free b128 reads).  SWZ = true: unpadded 128-B
// rows with the 16-B chunk index XORed by (row & 7): at most 2-way conflicts (irrelevant next to 64-cycle fp32 MFMAs)
// but 11 % less LDS, which is what lets THREE 128x64 workgroups share a CU.
// One LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS bytes [dst, dst + 1024) (dst wave-uniform).
// Inline asm on purpose: with the builtin, hipcc waits vmcnt(0) before EVERY later ds_read of the same __shared__ array
// (it cannot tell the two halves of the double buffer apart), which exposes the whole load latency each K-step.  The
// asm form is invisible to that bookkeeping; the kernel drains it with one explicit s_waitcnt vmcnt(0) before the
// end-of-step barrier.  M0 (the LDS-DMA base) is saved/restored inside the statement.
__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_dst_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_bytes) : "memory");
}

template <bool SWZ> __device__ __forceinline__ int lds_off(int row, int chunk) {
    return SWZ ? row * 32 + ((chunk ^ (row & 7)) << 2) : row * LDS_STRIDE + (chunk << 2);
}
template <bool SWZ> __device__ __forceinline__ float4 lds_frag(const float* base, int row, int chunk) {
    return *reinterpret_cast<const float4*>(base + lds_off<SWZ>(row, chunk));
}

struct ConvP {
    const float* x;          // [M][K] NHWC input (raw previous conv output if in_scale != null)
    const float* w;          // [NTAPS][N][K] packed weights
    float* y;                // [M][N]
    const float* in_scale;   // [K] or null
    const float* in_shift;
    float* partials;         // EPI 1/2: [ceil(M/BM)*WM][2][N]
    const float* yprev;      // EPI 2: raw output of the conv whose BN+ReLU produced this layer's input grad mask
    const float* p_scale;    // EPI 2: [N] ; EPI 3: unused
    const float* p_shift;    // EPI 2: [N] ; EPI 3: bias [N]
    const float* p_mean;     // EPI 2
    const float* p_invstd;   // EPI 2
    int H, W, K, N;
    long M;
    int tune;                // experiment knobs (SED_TUNE env var; 0 = shipped configuration)
    // second operand set, selected by blockIdx.z == 1 (NTAPS == 1 only): two independent GEMMs of the same shape in
    // one launch (the two directions of the BiGRU recurrence)
    const float* x2;
    const float* w2;
    float* y2;
    const float* b2;
};

static int sed_tune() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SED_TUNE"); v = e ? atoi(e) : 0; }
    return v;
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Staging of one K-step of the implicit GEMM is written with NAMED registers (macros over literal row indices), not
// arrays: hipcc left a `float4 breg[4]` array in scratch memory (private segment) with an `s_waitcnt vmcnt(0)` right
// after the loads, which exposed the full global-load latency every K-step (measured: 93 -> see DESIGN.md).
template <int WM, int WN, int TM, int TN, int NTAPS, bool INT, int EPI>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvP p) {
    if (NTAPS == 1 && blockIdx.z == 1) { p.x = p.x2; p.w = p.w2; p.y = p.y2; p.p_shift = p.b2; }
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_LD = BM * 8 / 256, B_LD = BN * 8 / 256;
    static_assert(WM * WN == 4 && A_LD <= 8 && B_LD <= 4, "4 waves; staging macros cover 8 A rows / 4 B rows per thread");
    constexpr bool SWZ = (BN == 64);                         // Cout = 64 layers: unpadded swizzled rows (128x64: 48 KB, 3 WG/CU; 256x64: 80 KB, 2 WG/CU)
    constexpr int ROWF = SWZ ? 32 : LDS_STRIDE;
    // The weight tile needs no masking / transform, so it goes global -> LDS directly (global_load_lds_dwordx4: no
    // staging VGPRs, no ds_write).  An LDS-DMA instruction writes wave-uniform base + lane*16 B, i.e. 8 linear 128-B
    // rows, so the B tile is always the unpadded swizzled layout with the swizzle applied to the SOURCE chunk.
    constexpr bool BDMA = true;
    constexpr int BROWF = BDMA ? 32 : ROWF;
    __shared__ __attribute__((aligned(16))) float As[2][BM * ROWF];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BROWF];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    const int nt_n = p.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(tile / nt_n) * BM;
    const int n0 = (tile % nt_n) * BN;
    const int c4 = tid & 7, lrow = tid >> 3;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const unsigned bs_lds_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) float*)&Bs[0][0]);

    // per-row metadata of the rows this thread stages: A rows lrow + 32*i (i < A_LD), B rows lrow + 32*j (j < B_LD).
    // Rows past M alias row 0 and are always masked.
#define SED_ROW_META(i)                                                                                         \
    const float* aptr##i = p.x + c4 * 4; int rh##i = -(1 << 20), rw##i = -(1 << 20);                            \
    if (i < A_LD) {                                                                                             \
        long pm = m0 + lrow + 32 * i;                                                                           \
        if (pm < p.M) {                                                                                         \
            aptr##i = p.x + pm * p.K + c4 * 4;                                                                  \
            if (NTAPS == 9) {                                                                                   \
                unsigned pu = (unsigned)pm;                                                                     \
                rw##i = (int)(pu % (unsigned)p.W); rh##i = (int)((pu / (unsigned)p.W) % (unsigned)p.H);         \
            } else { rw##i = 0; rh##i = 0; }                                                                    \
        }                                                                                                       \
    }                                                                                                           \
    const int brow##i = (BN / 4) * wvu + 8 * (i < B_LD ? i : 0) + (lane >> 3);     /* B row staged by DMA i */      \
    const float* bptr##i = BDMA ? p.w + (long)(n0 + brow##i) * p.K + (((lane & 7) ^ (brow##i & 7)) << 2)            \
                                : p.w + (long)(n0 + lrow + 32 * (i < B_LD ? i : 0)) * p.K + c4 * 4;                \
    float4 areg##i = make_float4(0.f, 0.f, 0.f, 0.f), breg##i = areg##i;                                        \
    bool vld##i = false;
    SED_ROW_META(0) SED_ROW_META(1) SED_ROW_META(2) SED_ROW_META(3)
    SED_ROW_META(4) SED_ROW_META(5) SED_ROW_META(6) SED_ROW_META(7)
#undef SED_ROW_META

    floatx16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int kchunks = p.K >> 5;
    const int KT = kchunks * NTAPS;
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
    // A: zero outside the image (branch-free: an invalid row reads its own, always-safe pixel and is zeroed by a
    // select), optional BN+ReLU on the fly.  All per-step address arithmetic is wave-uniform (scalar).
#define SED_A_LOAD(i)                                                                                           \
    if (i < A_LD) {                                                                                             \
        vld##i = (NTAPS == 9) ? ((unsigned)(rh##i + dy) < (unsigned)p.H && (unsigned)(rw##i + dx) < (unsigned)p.W) \
                              : (rh##i >= 0);                                                                   \
        areg##i = *reinterpret_cast<const float4*>(aptr##i + (vld##i ? a_off : (long)c0));                      \
    }
// applied at LDS-store time (after the MFMA block), so nothing waits on the loads before the MFMAs start
#define SED_A_FIX(i)                                                                                            \
    if (i < A_LD) {                                                                                             \
        asm volatile("" : "+v"(areg##i.x), "+v"(areg##i.y), "+v"(areg##i.z), "+v"(areg##i.w));                  \
        if (INT) {                                                                                              \
            areg##i.x = bn_relu(areg##i.x, sc.x, sh.x); areg##i.y = bn_relu(areg##i.y, sc.y, sh.y);             \
            areg##i.z = bn_relu(areg##i.z, sc.z, sh.z); areg##i.w = bn_relu(areg##i.w, sc.w, sh.w);             \
        }                                                                                                       \
        areg##i.x = vld##i ? areg##i.x : 0.f; areg##i.y = vld##i ? areg##i.y : 0.f;                             \
        areg##i.z = vld##i ? areg##i.z : 0.f; areg##i.w = vld##i ? areg##i.w : 0.f;                             \
    }
#define SED_B_LOAD(DST, j)                                                                                      \
    if (j < B_LD) {                                                                                             \
        if (BDMA) lds_dma16(bptr##j + b_off, bs_lds_base + (unsigned)(((DST) * BN + (BN / 4) * wvu + 8 * j) * 128));  \
        else breg##j = *reinterpret_cast<const float4*>(bptr##j + b_off);                                       \
    }
#define gload(IT, DST)                                                                                          \
    {                                                                                                           \
        const int it_ = (IT);                                                                                   \
        const int tap = (NTAPS == 9) ? it_ % 9 : 0;                                                             \
        const int c0 = ((NTAPS == 9) ? it_ / 9 : it_) << 5;                                                     \
        const int dy = (NTAPS == 9) ? tap / 3 - 1 : 0, dx = (NTAPS == 9) ? tap % 3 - 1 : 0;                     \
        const long a_off = (long)(dy * p.W + dx) * p.K + c0;                                                    \
        const long b_off = (long)tap * p.N * p.K + c0;                                                          \
        if (INT) {                                                                                              \
            sc = *reinterpret_cast<const float4*>(p.in_scale + c0 + c4 * 4);                                    \
            sh = *reinterpret_cast<const float4*>(p.in_shift + c0 + c4 * 4);                                    \
        }                                                                                                       \
        SED_B_LOAD(DST, 0) SED_B_LOAD(DST, 1) SED_B_LOAD(DST, 2) SED_B_LOAD(DST, 3)                             \
        SED_A_LOAD(0) SED_A_LOAD(1) SED_A_LOAD(2) SED_A_LOAD(3)                                                 \
        SED_A_LOAD(4) SED_A_LOAD(5) SED_A_LOAD(6) SED_A_LOAD(7)                                                 \
    }
#define SED_A_STORE(BUF, i) if (i < A_LD) *reinterpret_cast<float4*>(&As[(BUF)][lds_off<SWZ>(lrow + 32 * i, c4)]) = areg##i;
#define SED_B_STORE(BUF, j) if (!BDMA && j < B_LD) *reinterpret_cast<float4*>(&Bs[(BUF)][lds_off<SWZ>(lrow + 32 * j, c4)]) = breg##j;
#define lstore(BUF)                                                                                             \
    {                                                                                                           \
        SED_A_FIX(0) SED_A_FIX(1) SED_A_FIX(2) SED_A_FIX(3) SED_A_FIX(4) SED_A_FIX(5) SED_A_FIX(6) SED_A_FIX(7) \
        SED_A_STORE(BUF, 0) SED_A_STORE(BUF, 1) SED_A_STORE(BUF, 2) SED_A_STORE(BUF, 3)                         \
        SED_A_STORE(BUF, 4) SED_A_STORE(BUF, 5) SED_A_STORE(BUF, 6) SED_A_STORE(BUF, 7)                         \
        SED_B_STORE(BUF, 0) SED_B_STORE(BUF, 1) SED_B_STORE(BUF, 2) SED_B_STORE(BUF, 3)                         \
    }

    gload(0, 0);
    lstore(0);
    if (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int arow_base = wm * TM * 32 + (lane & 31);
    const int brow_base = wn * TN * 32 + (lane & 31);
    for (int it = 0; it < KT; ++it) {
        const int buf = it & 1;
        // unconditional prefetch (the last iteration re-fetches its own tile into the idle buffer): no branches and
        // no "maybe-uninitialised" staging registers in the loop, which is what keeps them out of scratch memory
        // All staging loads are issued back to back above the MFMA block.  (Interleaving them one per MFMA with
        // sched_group_barrier was measured 5 % SLOWER; ablations in DESIGN.md §5.)
        gload(it + 1 < KT ? it + 1 : it, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);           // +1-2 % (A/B measured with tools/conv_bench.py)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = lds_frag<SWZ>(&As[buf][0], arow_base + a * 32, q * 2 + (lane >> 5));
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = lds_frag<(SWZ || BDMA)>(&Bs[buf][0], brow_base + b * 32, q * 2 + (lane >> 5));
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
                }
            if (q == 2) {
                // stage the next tile into the idle LDS buffer while the MFMA pipe drains the q=2 block: by now the
                // loads have had most of the K-step to land; only the barrier is left at the end
                __builtin_amdgcn_sched_barrier(0);
                lstore(buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (BDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the asm LDS-DMAs of this step have landed
        __syncthreads();
    }
#undef gload
#undef lstore
#undef SED_A_LOAD
#undef SED_A_FIX
#undef SED_B_LOAD
#undef SED_A_STORE
#undef SED_B_STORE

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const long wrow0 = m0 + wm * TM * 32;
    const int half = lane >> 5;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn * TN * 32 + b * 32 + (lane & 31);
        float s1 = 0.f, s2 = 0.f;
        float e_sc = 0.f, e_sh = 0.f, e_mu = 0.f, e_is = 0.f, bias = 0.f;
        if (EPI == 2) { e_sc = p.p_scale[col]; e_sh = p.p_shift[col]; e_mu = p.p_mean[col]; e_is = p.p_invstd[col]; }
        if (EPI == 3) bias = p.p_shift[col];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                long row = wrow0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[a][b][r];
                if (row < p.M) {
                    if (EPI == 2) {
                        float yv = p.yprev[row * p.N + col];
                        v = bn_relu_active(yv, e_sc, e_sh) ? v : 0.f;
                        s1 += v;
                        s2 = fmaf(v, (yv - e_mu) * e_is, s2);
                        acc[a][b][r] = v;
                    }
                    if (EPI == 3) v += bias;
                    if (EPI == 1) s1 += v;
                    p.y[row * p.N + col] = v;
                }
            }
        if (EPI == 1) {
            long cnt_l = p.M - wrow0;
            float cnt = (float)(cnt_l < 0 ? 0 : (cnt_l > TM * 32 ? TM * 32 : cnt_l));
            s1 += __shfl_xor(s1, 32, 64);
            float mean = cnt > 0.f ? s1 / cnt : 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    long row = wrow0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float d = acc[a][b][r] - mean;
                    if (row < p.M) s2 = fmaf(d, d, s2);
                }
            s2 += __shfl_xor(s2, 32, 64);
        }
        if (EPI == 2) { s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64); }
        if ((EPI == 1 || EPI == 2) && half == 0) {
            long part = (m0 / BM) * WM + wm;
            p.partials[(part * 2 + 0) * p.N + col] = s1;
            p.partials[(part * 2 + 1) * p.N + col] = s2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
struct WgradP {
    const float* x;          // [M][K] layer input (raw previous conv output if in_scale != null)
    const float* gy;         // [M][N] gradient wrt this conv's output
    float* partial;          // [nslices][NTAPS][N][K]
    const float* in_scale;
    const float* in_shift;
    int H, W, K, N;
    long M;
    int pix_per_slice;
    int ids_per_slice;       // NTAPS * (N / CO_T) * (K / CI_T)
};

template <int TM, int TN, int NTAPS, bool INT>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradP p) {
    constexpr int CO_T = 64 * TM, CI_T = 64 * TN, BKP = 32;
    constexpr int G_C4 = 16 * TM, X_C4 = 16 * TN;
    constexpr int G_RPP = 256 / G_C4, X_RPP = 256 / X_C4;
    constexpr int G_LD = BKP / G_RPP, X_LD = BKP / X_RPP;       // 2*TM, 2*TN float4 per thread per K-step (<= 4)
    __shared__ __attribute__((aligned(16))) float Gs[2][BKP * CO_T];
    __shared__ __attribute__((aligned(16))) float Xs[2][BKP * CI_T];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int ci_tiles = p.K / CI_T;
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8, so hand each XCD a CONTIGUOUS range of (slice, tile, tap)
    // items -- the 9 taps (and the channel tiles) of one pixel slice then share that XCD's L2 instead of each pulling
    // the same G/X tiles from HBM through a different L2 (PMC: 35 GB fetched for 8.4 GB of operands before this).
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int slice = logical / p.ids_per_slice;
    int id = logical % p.ids_per_slice;
    const int tap = (NTAPS == 9) ? id % 9 : 0;
    id = (NTAPS == 9) ? id / 9 : id;
    const int ci0 = (id % ci_tiles) * CI_T, co0 = (id / ci_tiles) * CO_T;
    const int dy = (NTAPS == 9) ? tap / 3 - 1 : 0, dx = (NTAPS == 9) ? tap % 3 - 1 : 0;
    const long pbeg = (long)slice * p.pix_per_slice;
    long pend = pbeg + p.pix_per_slice;
    if (pend > p.M) pend = p.M;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int g_c4 = tid % G_C4, g_r = tid / G_C4;
    const int x_c4 = tid % X_C4, x_r = tid / X_C4;
    float4 xsc = make_float4(0.f, 0.f, 0.f, 0.f), xsh = xsc;
    if (INT) {
        xsc = *reinterpret_cast<const float4*>(p.in_scale + ci0 + x_c4 * 4);
        xsh = *reinterpret_cast<const float4*>(p.in_shift + ci0 + x_c4 * 4);
    }
    // Named staging registers (see conv_igemm_kernel).  Each staged row advances by BKP = 32 pixels per K-step; its
    // (h, w) is tracked incrementally (one 32-bit divide per row per workgroup instead of per step).
    const int dq = (BKP / p.W) % p.H, dr = BKP % p.W;   // per-step advance of (h, w); dq < H so one wrap suffices
    const long x_tap_off = (long)(dy * p.W + dx) * p.K;
#define SED_W_META(i)                                                                                           \
    const float* gptr##i = p.gy + (pbeg + g_r + G_RPP * (i < G_LD ? i : 0)) * p.N + co0 + g_c4 * 4;             \
    const float* xptr##i = p.x + (pbeg + x_r + X_RPP * (i < X_LD ? i : 0)) * p.K + ci0 + x_c4 * 4;              \
    long gpm##i = pbeg + g_r + G_RPP * i, xpm##i = pbeg + x_r + X_RPP * i;                                      \
    int xh##i = 0, xw##i = 0;                                                                                   \
    if (NTAPS == 9 && i < X_LD) {                                                                               \
        unsigned pu = (unsigned)xpm##i;                                                                         \
        xw##i = (int)(pu % (unsigned)p.W); xh##i = (int)((pu / (unsigned)p.W) % (unsigned)p.H);                 \
    }                                                                                                           \
    float4 greg##i = make_float4(0.f, 0.f, 0.f, 0.f), xreg##i = greg##i;                                        \
    bool gv##i = false, xv##i = false;
    SED_W_META(0) SED_W_META(1) SED_W_META(2) SED_W_META(3)
#undef SED_W_META
    // loads are branch-free: an out-of-range row re-reads the slice's first (always valid) row and is zeroed at store
#define SED_G_LOAD(i)                                                                                           \
    if (i < G_LD) {                                                                                             \
        gv##i = gpm##i < pend;                                                                                  \
        greg##i = *reinterpret_cast<const float4*>(gv##i ? gptr##i : p.gy + pbeg * p.N + co0 + g_c4 * 4);       \
        gptr##i += (long)BKP * p.N; gpm##i += BKP;                                                              \
    }
#define SED_X_LOAD(i)                                                                                           \
    if (i < X_LD) {                                                                                             \
        xv##i = xpm##i < pend;                                                                                  \
        if (NTAPS == 9) xv##i = xv##i && (unsigned)(xh##i + dy) < (unsigned)p.H && (unsigned)(xw##i + dx) < (unsigned)p.W; \
        xreg##i = *reinterpret_cast<const float4*>(xv##i ? xptr##i + x_tap_off : p.x + pbeg * p.K + ci0 + x_c4 * 4); \
        xptr##i += (long)BKP * p.K; xpm##i += BKP;                                                              \
        if (NTAPS == 9) {                                                                                       \
            xw##i += dr; xh##i += dq;                                                                           \
            const bool cw = xw##i >= p.W;                                                                       \
            xw##i = cw ? xw##i - p.W : xw##i; xh##i = cw ? xh##i + 1 : xh##i;                                   \
            xh##i = xh##i >= p.H ? xh##i - p.H : xh##i;                                                         \
        }                                                                                                       \
    }
#define SED_PIN(r) asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
#define SED_G_STORE(BUF, i)                                                                                     \
    if (i < G_LD) {                                                                                             \
        SED_PIN(greg##i)                                                                                        \
        float4 v = greg##i;                                                                                     \
        v.x = gv##i ? v.x : 0.f; v.y = gv##i ? v.y : 0.f; v.z = gv##i ? v.z : 0.f; v.w = gv##i ? v.w : 0.f;     \
        *reinterpret_cast<float4*>(&Gs[(BUF)][(g_r + G_RPP * i) * CO_T + g_c4 * 4]) = v;                        \
    }
#define SED_X_STORE(BUF, i)                                                                                     \
    if (i < X_LD) {                                                                                             \
        SED_PIN(xreg##i)                                                                                        \
        float4 v = xreg##i;                                                                                     \
        if (INT) {                                                                                              \
            v.x = bn_relu(v.x, xsc.x, xsh.x); v.y = bn_relu(v.y, xsc.y, xsh.y);                                 \
            v.z = bn_relu(v.z, xsc.z, xsh.z); v.w = bn_relu(v.w, xsc.w, xsh.w);                                 \
        }                                                                                                       \
        v.x = xv##i ? v.x : 0.f; v.y = xv##i ? v.y : 0.f; v.z = xv##i ? v.z : 0.f; v.w = xv##i ? v.w : 0.f;     \
        *reinterpret_cast<float4*>(&Xs[(BUF)][(x_r + X_RPP * i) * CI_T + x_c4 * 4]) = v;                        \
    }
#define wg_load() { SED_G_LOAD(0) SED_G_LOAD(1) SED_G_LOAD(2) SED_G_LOAD(3) SED_X_LOAD(0) SED_X_LOAD(1) SED_X_LOAD(2) SED_X_LOAD(3) }
#define wg_store(BUF) { SED_G_STORE(BUF, 0) SED_G_STORE(BUF, 1) SED_G_STORE(BUF, 2) SED_G_STORE(BUF, 3) \
                        SED_X_STORE(BUF, 0) SED_X_STORE(BUF, 1) SED_X_STORE(BUF, 2) SED_X_STORE(BUF, 3) }

    const int nsteps = (int)((pend - pbeg + BKP - 1) / BKP);
    if (nsteps > 0) {
        wg_load();
        wg_store(0);
    }
    __syncthreads();
    const int half = lane >> 5;
    const int gcol = wm * 32 * TM + (lane & 31), xcol = wn * 32 * TN + (lane & 31);
    for (int it = 0; it < nsteps; ++it) {
        const int buf = it & 1;
        wg_load();                               // unconditional prefetch (rows past `pend` are masked to zero)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        float af[TM], bf[TN], afn[TM], bfn[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[a] = Gs[buf][half * CO_T + gcol + a * 32];
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[b] = Xs[buf][half * CI_T + xcol + b * 32];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < 15) {                        // fragments of the next pixel pair are in flight during these MFMAs
#pragma unroll
                for (int a = 0; a < TM; ++a) afn[a] = Gs[buf][(2 * j + 2 + half) * CO_T + gcol + a * 32];
#pragma unroll
                for (int b = 0; b < TN; ++b) bfn[b] = Xs[buf][(2 * j + 2 + half) * CI_T + xcol + b * 32];
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
            if (j == 11) {                       // stage the next tile while the MFMA pipe is busy (cf. conv_igemm_kernel)
                __builtin_amdgcn_sched_barrier(0);
                wg_store(buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = afn[a];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = bfn[b];
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
#undef SED_PIN
#undef SED_G_LOAD
#undef SED_X_LOAD
#undef SED_G_STORE
#undef SED_X_STORE
#undef wg_load
#undef wg_store
    float* out = p.partial + ((long)slice * NTAPS + tap) * p.N * p.K;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int co = co0 + wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                int ci = ci0 + wn * 32 * TN + b * 32 + (lane & 31);
                out[(long)co * p.K + ci] = acc[a][b][r];
            }
}

// sum the pixel-slice partials in fp64 and scatter into the reference's OIHW gradient layout
// (NTAPS == 9: out[co][ci][tap]; NTAPS == 1: out[n][k]).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nslices, int ntaps,
                                                           int N, int K, float* __restrict__ out) {
    const long per = (long)ntaps * N * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        double s = 0.0;
        for (int sl = 0; sl < nslices; ++sl) s += (double)partial[(long)sl * per + i];
        int ci = (int)(i % K);
        long q = i / K;
        int co = (int)(q % N);
        int tap = (int)(q / N);
        out[((long)co * K + ci) * ntaps + tap] = (float)s;
    }
}

// OIHW master weights -> forward pack wf[tap][co][ci] and dgrad pack wd[tap'][ci][co] = W[co][ci][8-tap']
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                           float* __restrict__ wf, float* __restrict__ wd) {
    const long total = (long)Cout * Cin * 9;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int tap = (int)(i % 9);
        long q = i / 9;
        int ci = (int)(q % Cin), co = (int)(q / Cin);
        float v = w[i];
        if (wf) wf[((long)tap * Cout + co) * Cin + ci] = v;
        if (wd) wd[((long)(8 - tap) * Cin + ci) * Cout + co] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// conv_block1.conv1: Cin = 1 -> Cout = 64, direct.  x0 [M] (M = B*H*W), w [64][1][3][3] (OIHW), y [M][64].
// Also emits BN partial statistics (sum, M2) per 256-pixel tile via pivot-shifted sums.
constexpr int C1_ROWS = 256;

// the 3x3 neighbourhood of pixel pm = (.., h, w) of the single-channel input, zero outside the image
__device__ __forceinline__ void c1_taps(const float* __restrict__ x0, long pm, int h, int w, int H, int W, float (&xs)[9]) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int dy = t / 3 - 1, dx = t % 3 - 1;
        bool valid = (unsigned)(h + dy) < (unsigned)H && (unsigned)(w + dx) < (unsigned)W;
        xs[t] = valid ? x0[pm + dy * W + dx] : 0.f;
    }
}

// (h, w) of a pixel index walked in fixed steps: one division at the start, then carried along (a 64-bit div/mod per row
// and lane made the Cin = 1 kernels VALU-bound)
struct C1Walk {
    int h, w;
    __device__ __forceinline__ C1Walk(long pm, int H, int W) : h((int)((pm / W) % H)), w((int)(pm % W)) {}
    __device__ __forceinline__ void advance(int step, int H, int W) {
        w += step;
        while (w >= W) { w -= W; if (++h == H) h = 0; }
    }
};

__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ w,
                                                        long M, int H, int W, float* __restrict__ y,
                                                        float* __restrict__ partials, float* __restrict__ minmax) {
    __shared__ float4 red_s[256], red_q[256];
    const int c4 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    float wr[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) wr[t][k] = w[(c4 * 4 + k) * 9 + t];
    const long base = (long)blockIdx.x * C1_ROWS;
    const long nrows = min((long)C1_ROWS, M - base);
    float piv[4] = {0, 0, 0, 0};
    {
        float xs[9];
        const C1Walk p0(base, H, W);
        c1_taps(x0, base, p0.h, p0.w, H, W, xs);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) piv[k] = fmaf(xs[t], wr[t][k], piv[k]);
    }
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    float mx[4], mn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { mx[k] = -__builtin_inff(); mn[k] = __builtin_inff(); }
    C1Walk pos(base + pl, H, W);
    for (int r = pl; r < nrows; r += 16, pos.advance(16, H, W)) {
        long pm = base + r;
        float xs[9], o[4] = {0, 0, 0, 0};
        c1_taps(x0, pm, pos.h, pos.w, H, W, xs);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaf(xs[t], wr[t][k], o[k]);
        if (y) store_nt4(y, pm * 16 + c4, make_float4(o[0], o[1], o[2], o[3]));     // null: statistics / range pass only
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = o[k] - piv[k]; s[k] += d; q[k] = fmaf(d, d, q[k]);
            mx[k] = fmaxf(mx[k], o[k]); mn[k] = fminf(mn[k], o[k]);
        }
    }
    if (minmax) {        // per-channel range of this block's rows: sed_act_amax turns it into the amax of relu(bn1(y))
        red_s[threadIdx.x] = make_float4(mx[0], mx[1], mx[2], mx[3]);
        red_q[threadIdx.x] = make_float4(mn[0], mn[1], mn[2], mn[3]);
        __syncthreads();
        if (threadIdx.x < 16) {
            float4 a = red_s[threadIdx.x], b = red_q[threadIdx.x];
            for (int j = 1; j < 16; ++j) {
                float4 a2 = red_s[threadIdx.x + 16 * j], b2 = red_q[threadIdx.x + 16 * j];
                a.x = fmaxf(a.x, a2.x); a.y = fmaxf(a.y, a2.y); a.z = fmaxf(a.z, a2.z); a.w = fmaxf(a.w, a2.w);
                b.x = fminf(b.x, b2.x); b.y = fminf(b.y, b2.y); b.z = fminf(b.z, b2.z); b.w = fminf(b.w, b2.w);
            }
            float4* po = reinterpret_cast<float4*>(minmax + (long)blockIdx.x * 128);
            po[threadIdx.x] = a;
            po[16 + threadIdx.x] = b;
        }
        __syncthreads();
    }
    if (partials) {
        red_s[threadIdx.x] = make_float4(s[0], s[1], s[2], s[3]);
        red_q[threadIdx.x] = make_float4(q[0], q[1], q[2], q[3]);
        __syncthreads();
        if (threadIdx.x < 16) {
            float4 a = red_s[threadIdx.x], b = red_q[threadIdx.x];
            for (int j = 1; j < 16; ++j) {
                float4 a2 = red_s[threadIdx.x + 16 * j], b2 = red_q[threadIdx.x + 16 * j];
                a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
                b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
            }
            float n = (float)nrows, inv = 1.0f / n;
            float4* po = reinterpret_cast<float4*>(partials + (long)blockIdx.x * 128);
            po[threadIdx.x] = make_float4(a.x + n * piv[0], a.y + n * piv[1], a.z + n * piv[2], a.w + n * piv[3]);
            po[16 + threadIdx.x] = make_float4(fmaxf(b.x - a.x * a.x * inv, 0.f), fmaxf(b.y - a.y * a.y * inv, 0.f),
                                               fmaxf(b.z - a.z * a.z * inv, 0.f), fmaxf(b.w - a.w * a.w * inv, 0.f));
        }
    }
}

// Second pass of block 1's first layer in training (round 4): a1 = relu(scale * conv1(x0) + shift) written ONCE, already in
// the operand format of the split-f16 kernels -- per channel PAIR two dwords {hi0 | hi1 << 16, lo0 | lo1 << 16} with
// hi = f16(sa * a), lo = f16(sa * a - hi), sa = the power-of-two scale of a_amax (what the fused-input staging of
// conv_sf16 / wgrad_sf16 computes from the raw y1 on every K-step and in every tile that reads it).  Same 4 bytes per
// element as the raw fp32 y1 this replaces; y1 itself is never materialised (BatchNorm's statistics come from a
// statistics-only pass of conv1_fwd_kernel; the backward kernels recompute it from the one-channel input where they
// need it).  The consumers' staging becomes a plain copy, and the MFMA operands are bit-identical to the fused path's.
__device__ __forceinline__ void c1_split2(float a, float b, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(a), "v"(b));
}
__device__ __forceinline__ float c1_scale_of(float amax) {      // = sf_scale_of of conv_sf16.hip
    if (!(amax > 0.f) || !(amax < __builtin_inff())) return 1.f;
    int e;
    frexpf(amax, &e);
    e = 14 - e;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return ldexpf(1.f, e);
}

__global__ __launch_bounds__(256) void conv1_act_sf16_kernel(const float* __restrict__ x0, const float* __restrict__ w,
                                                             long M, int H, int W, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ a_amax,
                                                             unsigned* __restrict__ out, int* __restrict__ err_host,
                                                             int* __restrict__ err_dev) {
    const int c4 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    float wr[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) wr[t][k] = w[(c4 * 4 + k) * 9 + t];
    const float sa = c1_scale_of(amax_read(a_amax));
    float4 sc = reinterpret_cast<const float4*>(scale)[c4], sh = reinterpret_cast<const float4*>(shift)[c4];
    // the power-of-two operand scale rides on the affine: relu(sa*sc*y + sa*sh) == sa*relu(sc*y + sh) bit for bit
    sc.x *= sa; sc.y *= sa; sc.z *= sa; sc.w *= sa; sh.x *= sa; sh.y *= sa; sh.z *= sa; sh.w *= sa;
    const long base = (long)blockIdx.x * C1_ROWS;
    const long nrows = min((long)C1_ROWS, M - base);
    bool bad = false;
    C1Walk pos(base + pl, H, W);
    for (int r = pl; r < nrows; r += 16, pos.advance(16, H, W)) {
        const long pm = base + r;
        float xs[9], o[4] = {0, 0, 0, 0};
        c1_taps(x0, pm, pos.h, pos.w, H, W, xs);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = fmaf(xs[t], wr[t][k], o[k]);       // the very sequence of conv1_fwd_kernel
        bad |= !((fabsf(o[0]) + fabsf(o[1])) + (fabsf(o[2]) + fabsf(o[3])) < __builtin_inff());   // (ReLU's fmaxf would swallow a NaN)
        unsigned h01, l01, h23, l23;
        c1_split2(bn_relu(o[0], sc.x, sh.x), bn_relu(o[1], sc.y, sh.y), h01, l01);
        c1_split2(bn_relu(o[2], sc.z, sh.z), bn_relu(o[3], sc.w, sh.w), h23, l23);
        const floatx4 ov = {__uint_as_float(h01), __uint_as_float(l01), __uint_as_float(h23), __uint_as_float(l23)};
        __builtin_nontemporal_store(ov, reinterpret_cast<floatx4*>(out) + (pm * 16 + c4));
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) {
        if (err_host) __hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (err_dev) __hip_atomic_store(err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// sum over the 16 lanes of a DPP row, result in every lane: four v_add_f32 with DPP operands (xor 1, xor 2 inside the
// quads, then half-mirror and mirror), no LDS crossbar traffic (__shfl_xor = ds_bpermute made this kernel LDS-issue-bound)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// backward of the Cin=1 conv: one pass over gy produces (i) per-block dW partials [nblk][9][64] and
// (ii) t[p][tap] = <gy[p][:], w[:, tap]> (the scatter form of dgrad); conv1_dgrad_gather then sums 9 neighbours.
constexpr int C1B_ROWS = 4096;      // rows per workgroup at most
// Rows per workgroup of a launch: 4096 while that gives at least three rounds of the 768 resident workgroups (three per CU), else
// what gives ~two rounds, never fewer than 256 (one row per thread and trip).  At the metric's batch size 4096 rows meant 501
// workgroups on 768 slots; at 4 clips per GPU the floor of 1024 rows (until round 6: the callers' fixed scratch) meant 250
// workgroups on 256 CUs -- one latency-bound wave set per CU, 67 us for a pass that takes 31 us at the rate of bs=32.  The scratch
// is sized by sed_conv1_bwd_partial_floats() now.
static int c1b_rows_for(long M) {
    if (M / C1B_ROWS >= 3 * 768) return C1B_ROWS;
    long rows = (M + 2 * 768 - 1) / (2 * 768);
    rows = (rows + 15) / 16 * 16;
    if (rows < 256) rows = 256;
    if (rows > C1B_ROWS) rows = C1B_ROWS;
    return (int)rows;
}
// AFF: gy is the masked dgrad output dz of the NEXT conv and the BatchNorm backward g = a*dz + b*y + c (coef [3][64])
// is applied on load, which saves the separate sed_bn_bwd_apply pass over the two largest tensors of the model.
// yraw null (round 4): y = conv1(x0) is RECOMPUTED from the nine taps the kernel loads anyway (36 FMAs per lane and row, the
// fma sequence of conv1_fwd_kernel: bit-identical) instead of read back -- 4.2 GB less traffic per step at batch 256
template <bool AFF>
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ w,
                                                        const float* __restrict__ gy, const float* __restrict__ yraw,
                                                        const float* __restrict__ coef, long M, int H, int W,
                                                        float* __restrict__ dw_partials, float* __restrict__ tbuf, int rows_per_wg) {
    __shared__ float red[16][9 * 64 + 4];
    const int c4 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    float wr[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) wr[t][k] = w[(c4 * 4 + k) * 9 + t];
    float dw[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) dw[t][k] = 0.f;
    float4 ca = make_float4(0.f, 0.f, 0.f, 0.f), cb = ca, cc = ca;
    if (AFF) {
        ca = reinterpret_cast<const float4*>(coef)[c4]; cb = reinterpret_cast<const float4*>(coef)[16 + c4];
        cc = reinterpret_cast<const float4*>(coef)[32 + c4];
    }
    const long base = (long)blockIdx.x * rows_per_wg;
    const long nrows = min((long)rows_per_wg, M - base);
    C1Walk pos(base + pl, H, W);
    for (int r = pl; r < nrows; r += 16, pos.advance(16, H, W)) {
        long pm = base + r;
        float4 g = load_nt4(gy, pm * 16 + c4);
        float xs[9];
        c1_taps(x0, pm, pos.h, pos.w, H, W, xs);
        if (AFF) {
            float4 v;
            if (yraw) {
                v = load_nt4(yraw, pm * 16 + c4);
            } else {
                float o[4] = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = fmaf(xs[t], wr[t][k], o[k]);
                v = make_float4(o[0], o[1], o[2], o[3]);
            }
            g.x = fmaf(ca.x, g.x, fmaf(cb.x, v.x, cc.x)); g.y = fmaf(ca.y, g.y, fmaf(cb.y, v.y, cc.y));
            g.z = fmaf(ca.z, g.z, fmaf(cb.z, v.z, cc.z)); g.w = fmaf(ca.w, g.w, fmaf(cb.w, v.w, cc.w));
        }
        float tp[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            dw[t][0] = fmaf(g.x, xs[t], dw[t][0]); dw[t][1] = fmaf(g.y, xs[t], dw[t][1]);
            dw[t][2] = fmaf(g.z, xs[t], dw[t][2]); dw[t][3] = fmaf(g.w, xs[t], dw[t][3]);
            tp[t] = g.x * wr[t][0] + g.y * wr[t][1] + g.z * wr[t][2] + g.w * wr[t][3];
        }
        if (tbuf) {
#pragma unroll
            for (int t = 0; t < 9; ++t) tp[t] = row16_sum(tp[t]);     // all 16 lanes of a pixel end up with the channel sum
            if (c4 < 9) {
                float v = tp[0];
#pragma unroll
                for (int t = 1; t < 9; ++t) v = (c4 == t) ? tp[t] : v;
                tbuf[pm * 9 + c4] = v;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[pl][t * 64 + c4 * 4 + k] = dw[t][k];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64; i += 256) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) v += red[j][i];
        dw_partials[(long)blockIdx.x * 576 + i] = v;
    }
}

__global__ __launch_bounds__(256) void conv1_dgrad_gather_kernel(const float* __restrict__ tbuf, long M, int H, int W,
                                                                 float* __restrict__ gx0) {
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < M; q += (long)gridDim.x * 256) {
        int w = (int)(q % W), h = (int)((q / W) % H);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int dy = t / 3 - 1, dx = t % 3 - 1;        // forward: y[p] += x[p + (dy,dx)] * w[t]  =>  p = q - (dy,dx)
            bool valid = (unsigned)(h - dy) < (unsigned)H && (unsigned)(w - dx) < (unsigned)W;
            if (valid) s += tbuf[(q - dy * W - dx) * 9 + t];
        }
        gx0[q] = s;
    }
}

// dW[co][0][tap] = sum over blocks of dw_partials[blk][tap][co]; one workgroup per tap (64 outputs) x 16 row slices,
// fixed-order tree in LDS (deterministic: no atomics on the value path)
__global__ __launch_bounds__(1024) void conv1_wgrad_reduce_kernel(const float* __restrict__ parts, int nblk,
                                                                  float* __restrict__ dw) {
    __shared__ double red[1024];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    double s = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;        // four loads in flight: one per dependent iteration cost a round trip each
    int b = sl;
    for (; b + 48 < nblk; b += 64) {
        const float v0 = parts[(long)b * 576 + i], v1 = parts[(long)(b + 16) * 576 + i];
        const float v2 = parts[(long)(b + 32) * 576 + i], v3 = parts[(long)(b + 48) * 576 + i];
        s += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
    }
    for (; b < nblk; b += 16) s += (double)parts[(long)b * 576 + i];
    s = (s + s1) + (s2 + s3);
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0) {
        for (int j = 1; j < 16; ++j) s += red[threadIdx.x + 64 * j];
        int tap = i / 64, co = i % 64;
        dw[co * 9 + tap] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int NTAPS>
int launch_igemm(const ConvP& p, bool in_transform, int epi, hipStream_t stream, int nz = 1) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    if (p.N % BN != 0 || p.K % 32 != 0) return SED_EINVAL;
    dim3 grid((unsigned)(sed_cdiv(p.M, BM) * (p.N / BN)), 1, (unsigned)nz), block(256);
#define SED_LAUNCH(INT_, EPI_) \
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, TM, TN, NTAPS, INT_, EPI_>), grid, block, 0, stream, p)
    if (in_transform) {
        if (epi == 0) SED_LAUNCH(true, 0); else if (epi == 1) SED_LAUNCH(true, 1); else return SED_EINVAL;
    } else {
        if (epi == 0) SED_LAUNCH(false, 0); else if (epi == 1) SED_LAUNCH(false, 1);
        else if (epi == 2) SED_LAUNCH(false, 2); else if (epi == 3) SED_LAUNCH(false, 3); else return SED_EINVAL;
    }
#undef SED_LAUNCH
    SED_LAUNCH_CHECK();
    return 0;
}

template <int TM, int TN, int NTAPS>
int launch_wgrad(const WgradP& p, int nslices, bool in_transform, hipStream_t stream) {
    if (p.N % (64 * TM) != 0 || p.K % (64 * TN) != 0) return SED_EINVAL;
    WgradP q = p;
    q.ids_per_slice = NTAPS * (p.N / (64 * TM)) * (p.K / (64 * TN));
    dim3 grid((unsigned)((long)q.ids_per_slice * nslices)), block(256);
    if (in_transform) hipLaunchKernelGGL((wgrad_kernel<TM, TN, NTAPS, true>), grid, block, 0, stream, q);
    else hipLaunchKernelGGL((wgrad_kernel<TM, TN, NTAPS, false>), grid, block, 0, stream, q);
    SED_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------------

// rows covered by one statistics partial of sed_conv3x3_igemm for a given Cout (depends on the tile config)
// Statistics partials cover 64 rows each for the 128x128 and 256x64 tiles and 32 rows for the small-M 128x64 tile.
// The 256x64 tile (80 KB, 2 WG/CU) measured +4 % on epilogue-0 launches but -11 % with the dgrad epilogue 2 (64 extra
// yprev loads per lane at 2 WG/CU): net zero on the step, so it stays an experiment knob (SED_TUNE bit 2).
static bool sed_big64(long M) { return M >= 65536 && (sed_tune() & 4); }
SED_API int sed_conv_rows_per_part(long M, int Cout) { return (Cout >= 128 || sed_big64(M)) ? 64 : 32; }
SED_API int sed_conv_num_parts(long M, int Cout) {
    if (Cout >= 128) return sed_cdiv(M, 128) * 2;
    return sed_big64(M) ? sed_cdiv(M, 256) * 4 : sed_cdiv(M, 128) * 4;
}

// Forward or dgrad 3x3 conv on fp32 MFMA.  x [B*H*W][Cin], w_packed [9][Cout][Cin], y [B*H*W][Cout].
//   epi 0: plain store.   epi 1: + BN statistics partials (sum, M2) of y.
//   epi 2: y <- y * [relu mask of (p_scale*yprev + p_shift)], + partials (sum dy, sum dy*xhat)   (dgrad side)
//   in_scale/in_shift != null: the input operand is relu(in_scale*x + in_shift) computed on the fly.
SED_API int sed_conv3x3_igemm(const float* x, const float* w_packed, float* y, int B, int H, int W, int Cin, int Cout,
                              const float* in_scale, const float* in_shift, int epi, float* partials,
                              const float* yprev, const float* p_scale, const float* p_shift, const float* p_mean,
                              const float* p_invstd, hipStream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin % 32 != 0 || Cout % 64 != 0) return SED_EINVAL;
    if ((long)B * H * W * (long)(Cin > Cout ? Cin : Cout) >= (1L << 31) * 16L) return SED_EINVAL;
    if ((long)B * H * W >= (1L << 31)) return SED_EINVAL;
    ConvP p{x, w_packed, y, in_scale, in_shift, partials, yprev, p_scale, p_shift, p_mean, p_invstd, H, W, Cin, Cout,
            (long)B * H * W, sed_tune(), nullptr, nullptr, nullptr, nullptr};
    bool in_t = in_scale != nullptr;
    if (Cout >= 128) return launch_igemm<2, 2, 2, 2, 9>(p, in_t, epi, stream);
    if (sed_big64(p.M)) return launch_igemm<4, 1, 2, 2, 9>(p, in_t, epi, stream);   // 256 x 64 tile (experiment)
    return launch_igemm<4, 1, 1, 2, 9>(p, in_t, epi, stream);
}

// Plain fp32-MFMA GEMM, "NT": y[M][N] = x[M][K] * w[N][K]^T (+ bias[N]).  K % 32 == 0, N % 64 == 0.
SED_API int sed_gemm_nt(const float* x, const float* w, const float* bias, float* y, long M, int N, int K,
                        hipStream_t stream) {
    if (M <= 0 || K % 32 != 0 || N % 64 != 0 || M >= (1L << 31)) return SED_EINVAL;
    ConvP p{x, w, y, nullptr, nullptr, nullptr, nullptr, nullptr, bias, nullptr, nullptr, 1, 1, K, N, M, 0, nullptr, nullptr, nullptr, nullptr};
    int epi = bias ? 3 : 0;
    if (N % 128 == 0 && M >= 4096) return launch_igemm<2, 2, 2, 2, 1>(p, false, epi, stream);
    return launch_igemm<2, 2, 1, 1, 1>(p, false, epi, stream);
}

// Two independent NT GEMMs of one shape in ONE launch (blockIdx.z picks the operand set): the forward and backward
// directions of the BiGRU recurrence step.  bias0/bias1 may both be null.
SED_API int sed_gemm_nt_pair(const float* x0, const float* x1, const float* w0, const float* w1, const float* bias0,
                             const float* bias1, float* y0, float* y1, long M, int N, int K, hipStream_t stream) {
    if (M <= 0 || K % 32 != 0 || N % 64 != 0 || M >= (1L << 31) || ((bias0 == nullptr) != (bias1 == nullptr))) return SED_EINVAL;
    ConvP p{x0, w0, y0, nullptr, nullptr, nullptr, nullptr, nullptr, bias0, nullptr, nullptr, 1, 1, K, N, M, 0, x1, w1, y1, bias1};
    int epi = bias0 ? 3 : 0;
    if (N % 128 == 0 && M >= 4096) return launch_igemm<2, 2, 2, 2, 1>(p, false, epi, stream, 2);
    return launch_igemm<2, 2, 1, 1, 1>(p, false, epi, stream, 2);
}

SED_API long sed_wgrad_partial_floats(long M, int Cin, int Cout, int ntaps, int* nslices_out, int* pix_per_slice_out) {
    // Pixel slices: (i) bound the fp32 accumulation chain (<= 16384 pixels per slice), (ii) fill the chip, and
    // (iii) make the workgroup count land just under a whole number of "rounds" of the chip's concurrent capacity
    // (256 CUs x workgroups/CU for the tile config) -- a 4.4-round grid wastes the tail of its 5th round.
    const int tm = Cout >= 128 ? 2 : 1, tn = Cin >= 128 ? 2 : 1;
    const long tiles = (long)ntaps * (Cout / (64 * tm)) * (Cin / (64 * tn));
    const long capacity = 256L * (tm * tn == 4 ? 2 : (tm * tn == 2 ? 3 : 4));
    long ns_min = (M + 16383) / 16384;
    long fill = (2 * capacity + tiles - 1) / tiles;
    if (fill > ns_min) ns_min = fill;
    long rounds = (tiles * ns_min + capacity - 1) / capacity;
    long ns = rounds * capacity / tiles;
    if (ns < ns_min) ns = ns_min;
    long pps = ((M + ns - 1) / ns + 31) / 32 * 32;
    if (pps < 256) pps = 256;                    // tiny problems: do not explode the partial buffer
    ns = (M + pps - 1) / pps;
    if (nslices_out) *nslices_out = (int)ns;
    if (pix_per_slice_out) *pix_per_slice_out = (int)pps;
    return ns * ntaps * (long)Cin * Cout;
}

// dW (OIHW, [Cout][Cin][3][3]) = sum_p gy[p][co] * a[p+tap][ci];  a = relu(in_scale*x+in_shift) if in_scale.
// partial: scratch of sed_wgrad_partial_floats(...) floats.
SED_API int sed_conv3x3_wgrad(const float* x, const float* gy, float* dw_oihw, float* partial, int B, int H, int W,
                              int Cin, int Cout, const float* in_scale, const float* in_shift, hipStream_t stream) {
    if (B <= 0 || Cin % 64 != 0 || Cout % 64 != 0 || (long)B * H * W >= (1L << 31)) return SED_EINVAL;
    long M = (long)B * H * W;
    int ns, pps;
    sed_wgrad_partial_floats(M, Cin, Cout, 9, &ns, &pps);
    WgradP p{x, gy, partial, in_scale, in_shift, H, W, Cin, Cout, M, pps, 0};
    bool in_t = in_scale != nullptr;
    int rc;
    if (Cout >= 128 && Cin >= 128) rc = launch_wgrad<2, 2, 9>(p, ns, in_t, stream);
    else if (Cout >= 128) rc = launch_wgrad<2, 1, 9>(p, ns, in_t, stream);
    else rc = launch_wgrad<1, 1, 9>(p, ns, in_t, stream);
    if (rc) return rc;
    long per = 9L * Cin * Cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(sed_cdiv(per, 256) > 4096 ? 4096 : sed_cdiv(per, 256)), dim3(256), 0, stream,
                       partial, ns, 9, Cout, Cin, dw_oihw);
    SED_LAUNCH_CHECK();
    return 0;
}

// "TN" GEMM: dw[N][K] = sum_m gy[m][n] * x[m][k]   (weight gradients of the dense layers).  N,K % 64 == 0.
SED_API int sed_gemm_tn(const float* x, const float* gy, float* dw, float* partial, long M, int N, int K,
                        hipStream_t stream) {
    if (M <= 0 || N % 64 != 0 || K % 64 != 0 || M >= (1L << 31)) return SED_EINVAL;
    int ns, pps;
    sed_wgrad_partial_floats(M, K, N, 1, &ns, &pps);
    WgradP p{x, gy, partial, nullptr, nullptr, 1, 1, K, N, M, pps, 0};
    int rc;
    if (N >= 128 && K >= 128) rc = launch_wgrad<2, 2, 1>(p, ns, false, stream);
    else if (N >= 128) rc = launch_wgrad<2, 1, 1>(p, ns, false, stream);
    else rc = launch_wgrad<1, 1, 1>(p, ns, false, stream);
    if (rc) return rc;
    long per = (long)N * K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(sed_cdiv(per, 256) > 4096 ? 4096 : sed_cdiv(per, 256)), dim3(256), 0, stream,
                       partial, ns, 1, N, K, dw);
    SED_LAUNCH_CHECK();
    return 0;
}

// OIHW -> wf [9][Cout][Cin] (forward) and wd [9][Cin][Cout] tap-flipped (dgrad).  Either output may be null.
SED_API int sed_pack_conv_weights(const float* w_oihw, int Cout, int Cin, float* wf, float* wd, hipStream_t stream) {
    long total = (long)Cout * Cin * 9;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(sed_cdiv(total, 256) > 2048 ? 2048 : sed_cdiv(total, 256)), dim3(256), 0, stream,
                       w_oihw, Cout, Cin, wf, wd);
    SED_LAUNCH_CHECK();
    return 0;
}

// conv_block1.conv1 (Cin=1, Cout=64).  partials (nullable): ceil(M/256)*128 floats, rows per part = 256.
SED_API int sed_conv1_fwd(const float* x0, const float* w_oihw, float* y, int B, int H, int W, float* partials, float* minmax,
                          hipStream_t stream) {
    long M = (long)B * H * W;
    if (M <= 0 || M >= (1L << 31)) return SED_EINVAL;
    hipLaunchKernelGGL(conv1_fwd_kernel, dim3(sed_cdiv(M, C1_ROWS)), dim3(256), 0, stream, x0, w_oihw, M, H, W, y, partials, minmax);
    SED_LAUNCH_CHECK();
    return 0;
}
SED_API int sed_conv1_rows_per_part(void) { return C1_ROWS; }

// a1 = relu(scale * conv1(x0) + shift) as split-f16 operand pairs (see conv1_act_sf16_kernel); out: B*H*W*64 dwords.
SED_API int sed_conv1_act_sf16(const float* x0, const float* w_oihw, int B, int H, int W, const float* scale, const float* shift,
                               const float* a_amax, void* out, int* err_host, int* err_dev, hipStream_t stream) {
    long M = (long)B * H * W;
    if (!x0 || !w_oihw || !scale || !shift || !a_amax || !out || M <= 0 || M >= (1L << 31)) return SED_EINVAL;
    hipLaunchKernelGGL(conv1_act_sf16_kernel, dim3(sed_cdiv(M, C1_ROWS)), dim3(256), 0, stream, x0, w_oihw, M, H, W, scale, shift,
                       a_amax, (unsigned*)out, err_host, err_dev);
    SED_LAUNCH_CHECK();
    return 0;
}

// backward of conv_block1.conv1: dw [64][1][3][3]; gx0 [M] (nullable: skip the input gradient).
// scratch: dw_partials sed_conv1_bwd_partial_floats(B, H, W) floats; tbuf M*9 floats (only if gx0).
// bn_coef (nullable): gy is then the masked dgrad output dz and g = a*dz + b*y1 + c is formed on load (coef [3][64] from
// sed_bn_bwd_finalize), replacing a sed_bn_bwd_apply pass; y1 = bn_y, or recomputed from x0 when bn_y is null.
SED_API long sed_conv1_bwd_partial_floats(int B, int H, int W) {
    const long M = (long)B * H * W;
    if (M <= 0) return 0;
    return (long)sed_cdiv(M, c1b_rows_for(M)) * 576;
}

SED_API int sed_conv1_bwd(const float* x0, const float* w_oihw, const float* gy, const float* bn_y, const float* bn_coef,
                          int B, int H, int W, float* dw, float* gx0, float* dw_partials, float* tbuf, hipStream_t stream) {
    long M = (long)B * H * W;
    if (M <= 0 || M >= (1L << 31) / 9 || (bn_y != nullptr && bn_coef == nullptr)) return SED_EINVAL;
    const int rows = c1b_rows_for(M);
    int nblk = sed_cdiv(M, rows);
    if (bn_coef)
        hipLaunchKernelGGL(conv1_bwd_kernel<true>, dim3(nblk), dim3(256), 0, stream, x0, w_oihw, gy, bn_y, bn_coef, M, H, W,
                           dw_partials, gx0 ? tbuf : (float*)nullptr, rows);
    else
        hipLaunchKernelGGL(conv1_bwd_kernel<false>, dim3(nblk), dim3(256), 0, stream, x0, w_oihw, gy, bn_y, bn_coef, M, H, W,
                           dw_partials, gx0 ? tbuf : (float*)nullptr, rows);
    hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(9), dim3(1024), 0, stream, dw_partials, nblk, dw);
    if (gx0) {
        int g = sed_cdiv(M, 256);
        hipLaunchKernelGGL(conv1_dgrad_gather_kernel, dim3(g > 8192 ? 8192 : g), dim3(256), 0, stream, tbuf, M, H, W, gx0);
    }
    SED_LAUNCH_CHECK();
    return 0;
}

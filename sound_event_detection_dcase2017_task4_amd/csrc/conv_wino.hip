// 3x3 convolution as implicit GEMM with a FUSED 1-D Winograd F(2,3) along the mel (W) axis, fp32 MFMA.
//
// For one output row and a pair of adjacent outputs (w0, w0+1) the three horizontal taps g0,g1,g2 over the four inputs
// d0..d3 = x[w0-1 .. w0+2] are evaluated with 4 products instead of 6:
//     m0 = (d0-d2) g0,  m1 = (d1+d2) (g0+g1+g2)/2,  m2 = (d2-d1) (g0-g1+g2)/2,  m3 = (d1-d3) g2
//     y0 = m0+m1+m2,    y1 = m1-m2-m3
// Each m_xi is itself a contraction over (ky, ci), i.e. four GEMMs [64 pairs x 3*Cin] x [3*Cin x Cout] replace one
// [128 pixels x 9*Cin] x [9*Cin x Cout]: 12*Cin MACs per output pair instead of 18*Cin (1.5x fewer MFMA flops; the
// coefficients are +-1 and 1/2, so the arithmetic stays fp32-exact-class).  Nothing is materialised in HBM: the raw
// input rows (with a zero halo column on each side) are staged in LDS ONCE per (ky, channel chunk) and serve all three
// horizontal taps; the four transformed operands are formed in registers from four LDS rows when the MFMA fragment is
// read; the inverse transform is applied to the four accumulators in the epilogue.
//
// Same fusions as conv_igemm_kernel (conv.hip): input relu(scale*x+shift), epilogue 1 BN statistics, epilogue 2
// ReLU mask + BN-backward sums.  Requirements: W even, W | 128, Cin % 16 == 0, Cout % 64 == 0.
#include "common.h"
#include "sed_hip.h"
SED_OBJECT_FLAGS(conv_wino)

namespace {

constexpr int WBK = 16;            // channels per K-step
constexpr int WAS = 16;            // A row stride in floats: unpadded 64-B rows, 16-B chunk index XOR ((row >> 1) & 3)
__device__ __forceinline__ int wa_off(int row, int chunk) { return row * WAS + ((chunk ^ ((row >> 1) & 3)) << 2); }
constexpr int WMAXROWS = 160;      // RPT*(W+2) <= 16*10

struct WinoP {
    const float* x;          // [M][K]
    const float* wu;         // [3 ky][4 xi][N][K] transformed weights
    float* y;                // [M][N]
    const float* in_scale;
    const float* in_shift;
    float* partials;         // EPI 1/2: [ceil(M/128)*2][2][N]
    const float* yprev;
    const float* p_scale;
    const float* p_shift;
    const float* p_mean;
    const float* p_invstd;
    int H, W, K, N;
    long M;
};

__device__ __forceinline__ int xcd_remap_w(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ void lds_dma16_w(const float* gsrc, unsigned lds_dst_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_bytes) : "memory");
}

// 128 output pixels (= 64 horizontal pairs) x 64 output channels per workgroup; 4 waves as 2 (pairs) x 2 (channels),
// each wave 32 pairs x 32 channels x 4 Winograd positions = 4 MFMA accumulators.
template <bool INT, int EPI>
__global__ __launch_bounds__(256, 3) void conv_wino_kernel(WinoP p) {
    __shared__ __attribute__((aligned(16))) float As[2][WMAXROWS * WAS];   // raw rows incl. halo columns
    __shared__ __attribute__((aligned(16))) float Bs[2][4 * 64 * WBK];     // [xi][n][k], unpadded, source-swizzled

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int wm = wvu >> 1, wn = wvu & 1;
    const int nt_n = p.N / 64;
    const int tile = xcd_remap_w(blockIdx.x, gridDim.x);
    const long m0 = (long)(tile / nt_n) * 128;
    const int n0 = (tile % nt_n) * 64;
    const int W = p.W, W2 = W + 2, halfW = W >> 1;

    // zero both A buffers once: the halo columns (w = -1, w = W) stay zero for the whole kernel
    for (int i = tid; i < 2 * WMAXROWS * WAS / 4; i += 256) reinterpret_cast<float4*>(&As[0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // staging: pixel (tid >> 2) + 64*i, 16-B chunk (tid & 3)
    const int c4 = tid & 3;
#define SED_WMETA(i)                                                                                            \
    const long pm##i = m0 + (tid >> 2) + 64 * i;                                                                \
    const bool pv##i = pm##i < p.M;                                                                             \
    int ph##i = 0, lrow##i = 0;                                                                                 \
    {                                                                                                           \
        unsigned pu = (unsigned)(pv##i ? pm##i : 0);                                                            \
        int w_ = (int)(pu % (unsigned)W);                                                                       \
        ph##i = (int)((pu / (unsigned)W) % (unsigned)p.H);                                                      \
        lrow##i = (((tid >> 2) + 64 * i) / W) * W2 + w_ + 1;                                                    \
    }                                                                                                           \
    const float* aptr##i = p.x + (pv##i ? pm##i : 0) * p.K + c4 * 4;                                            \
    float4 areg##i = make_float4(0.f, 0.f, 0.f, 0.f);                                                           \
    bool vld##i = false;
    SED_WMETA(0) SED_WMETA(1)
#undef SED_WMETA
    // B DMA: wave wv stages Winograd position xi = wv; instruction j covers rows 16j .. 16j+15 (64-B rows)
#define SED_WBMETA(j)                                                                                           \
    const int brow##j = 16 * j + (lane >> 2);                                                                   \
    const float* bptr##j = p.wu + ((long)wvu * p.N + n0 + brow##j) * p.K + (((lane & 3) ^ ((brow##j >> 2) & 3)) << 2);
    SED_WBMETA(0) SED_WBMETA(1) SED_WBMETA(2) SED_WBMETA(3)
#undef SED_WBMETA
    const unsigned bs_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)&Bs[0][0]);

    floatx16 acc[4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    const int kchunks = p.K / WBK;
    const int KT = kchunks * 3;
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;

#define SED_WA_LOAD(i)                                                                                          \
    {                                                                                                           \
        vld##i = pv##i && (unsigned)(ph##i + dy) < (unsigned)p.H;                                               \
        areg##i = *reinterpret_cast<const float4*>(aptr##i + (vld##i ? a_off : (long)c0));                      \
    }
#define SED_WB_LOAD(DST, j) lds_dma16_w(bptr##j + b_off, bs_base + (unsigned)(((DST) * 4 * 64 * WBK + (wvu * 64 + 16 * j) * WBK) * 4));
#define wgload(IT, DST)                                                                                         \
    {                                                                                                           \
        const int it_ = (IT);                                                                                   \
        const int ky = it_ % 3;                                                                                 \
        const int c0 = (it_ / 3) * WBK;                                                                         \
        const int dy = ky - 1;                                                                                  \
        const long a_off = (long)dy * W * p.K + c0;                                                             \
        const long b_off = (long)ky * 4 * p.N * p.K + c0;                                                       \
        if (INT) {                                                                                              \
            sc = *reinterpret_cast<const float4*>(p.in_scale + c0 + c4 * 4);                                    \
            sh = *reinterpret_cast<const float4*>(p.in_shift + c0 + c4 * 4);                                    \
        }                                                                                                       \
        SED_WB_LOAD(DST, 0) SED_WB_LOAD(DST, 1) SED_WB_LOAD(DST, 2) SED_WB_LOAD(DST, 3)                         \
        SED_WA_LOAD(0) SED_WA_LOAD(1)                                                                           \
    }
#define SED_WA_STORE(BUF, i)                                                                                    \
    {                                                                                                           \
        asm volatile("" : "+v"(areg##i.x), "+v"(areg##i.y), "+v"(areg##i.z), "+v"(areg##i.w));                  \
        float4 v = areg##i;                                                                                     \
        if (INT) {                                                                                              \
            v.x = bn_relu(v.x, sc.x, sh.x); v.y = bn_relu(v.y, sc.y, sh.y);                                     \
            v.z = bn_relu(v.z, sc.z, sh.z); v.w = bn_relu(v.w, sc.w, sh.w);                                     \
        }                                                                                                       \
        v.x = vld##i ? v.x : 0.f; v.y = vld##i ? v.y : 0.f; v.z = vld##i ? v.z : 0.f; v.w = vld##i ? v.w : 0.f; \
        *reinterpret_cast<float4*>(&As[(BUF)][wa_off(lrow##i, c4)]) = v;                                        \
    }
#define wlstore(BUF) { SED_WA_STORE(BUF, 0) SED_WA_STORE(BUF, 1) }

    __syncthreads();                                   // zero fill visible before the first stores
    wgload(0, 0);
    wlstore(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment addressing: lane (i = lane & 31) owns pair t = wm*32 + i of the tile -> LDS rows base .. base+3
    const int t = wm * 32 + (lane & 31);
    const int arow0 = (t / halfW) * W2 + 2 * (t % halfW);
    const int brow = wn * 32 + (lane & 31);

    for (int it = 0; it < KT; ++it) {
        const int buf = it & 1;
        wgload(it + 1 < KT ? it + 1 : it, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int chunk = q * 2 + (lane >> 5);
            float4 d0 = *reinterpret_cast<const float4*>(&As[buf][wa_off(arow0, chunk)]);
            float4 d1 = *reinterpret_cast<const float4*>(&As[buf][wa_off(arow0 + 1, chunk)]);
            float4 d2 = *reinterpret_cast<const float4*>(&As[buf][wa_off(arow0 + 2, chunk)]);
            float4 d3 = *reinterpret_cast<const float4*>(&As[buf][wa_off(arow0 + 3, chunk)]);
            float4 bf[4];
#pragma unroll
            for (int x = 0; x < 4; ++x)
                bf[x] = *reinterpret_cast<const float4*>(&Bs[buf][(x * 64 + brow) * WBK + ((chunk ^ ((brow >> 2) & 3)) << 2)]);
            float4 v0 = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
            float4 v1 = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
            float4 v2 = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
            float4 v3 = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
#define SED_WMMA(XI, VV)                                                                                        \
    acc[XI] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.x, bf[XI].x, acc[XI], 0, 0, 0);                            \
    acc[XI] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.y, bf[XI].y, acc[XI], 0, 0, 0);                            \
    acc[XI] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.z, bf[XI].z, acc[XI], 0, 0, 0);                            \
    acc[XI] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.w, bf[XI].w, acc[XI], 0, 0, 0);
            SED_WMMA(0, v0) SED_WMMA(1, v1) SED_WMMA(2, v2) SED_WMMA(3, v3)
#undef SED_WMMA
        }
        __builtin_amdgcn_sched_barrier(0);
        wlstore(buf ^ 1);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef wgload
#undef wlstore
#undef SED_WA_LOAD
#undef SED_WB_LOAD
#undef SED_WA_STORE

    // ---- epilogue: inverse transform y0 = m0+m1+m2, y1 = m1-m2-m3; pair t <-> pixels m0 + 2t, m0 + 2t + 1
    const int half = lane >> 5;
    const int col = n0 + wn * 32 + (lane & 31);
    const long wrow0 = m0 + wm * 64;                   // first pixel of this wave's 64-pixel span
    float s1 = 0.f, s2 = 0.f;
    float e_sc = 0.f, e_sh = 0.f, e_mu = 0.f, e_is = 0.f;
    if (EPI == 2) { e_sc = p.p_scale[col]; e_sh = p.p_shift[col]; e_mu = p.p_mean[col]; e_is = p.p_invstd[col]; }
    float yv[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int tl = (r & 3) + 8 * (r >> 2) + 4 * half;    // pair index within the wave's 32 pairs
        float y0 = acc[0][r] + acc[1][r] + acc[2][r];
        float y1 = acc[1][r] - acc[2][r] - acc[3][r];
        long row = wrow0 + 2 * tl;
        if (EPI == 2) {
            if (row < p.M) {
                float a0 = p.yprev[row * p.N + col], a1 = p.yprev[(row + 1) * p.N + col];
                y0 = bn_relu_active(a0, e_sc, e_sh) ? y0 : 0.f;
                y1 = bn_relu_active(a1, e_sc, e_sh) ? y1 : 0.f;
                s1 += y0 + y1;
                s2 = fmaf(y0, (a0 - e_mu) * e_is, s2);
                s2 = fmaf(y1, (a1 - e_mu) * e_is, s2);
            }
        }
        if (row < p.M) {                               // M is a multiple of W (even): both pixels of a pair are valid together
            p.y[row * p.N + col] = y0;
            p.y[(row + 1) * p.N + col] = y1;
            if (EPI == 1) s1 += y0 + y1;
        }
        yv[2 * r] = y0; yv[2 * r + 1] = y1;
    }
    if (EPI == 1) {
        long cnt_l = p.M - wrow0;
        float cnt = (float)(cnt_l < 0 ? 0 : (cnt_l > 64 ? 64 : cnt_l));
        s1 += __shfl_xor(s1, 32, 64);
        float mean = cnt > 0.f ? s1 / cnt : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int tl = (r & 3) + 8 * (r >> 2) + 4 * half;
            long row = wrow0 + 2 * tl;
            if (row < p.M) {
                float da = yv[2 * r] - mean, db = yv[2 * r + 1] - mean;
                s2 = fmaf(da, da, s2);
                s2 = fmaf(db, db, s2);
            }
        }
        s2 += __shfl_xor(s2, 32, 64);
    }
    if (EPI == 2) { s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64); }
    if ((EPI == 1 || EPI == 2) && half == 0) {
        long part = (m0 / 128) * 2 + wm;
        p.partials[(part * 2 + 0) * p.N + col] = s1;
        p.partials[(part * 2 + 1) * p.N + col] = s2;
    }
}

// OIHW -> Winograd-domain packs.  Forward operand uf[ky][xi][co][ci] from g(kx) = W[co][ci][ky][kx]; dgrad operand
// ud[ky][xi][ci][co] from g(kx) = W[co][ci][2-ky][2-kx].  u0 = g0, u1 = (g0+g1+g2)/2, u2 = (g0-g1+g2)/2, u3 = g2.
__global__ __launch_bounds__(256) void pack_wino_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                        float* __restrict__ uf, float* __restrict__ ud) {
    const long total = (long)Cout * Cin * 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int ky = (int)(i % 3);
        long q = i / 3;
        int ci = (int)(q % Cin), co = (int)(q / Cin);
        const float* g = w + ((long)co * Cin + ci) * 9 + ky * 3;
        float g0 = g[0], g1 = g[1], g2 = g[2];
        if (uf) {
            float u[4] = {g0, 0.5f * (g0 + g1 + g2), 0.5f * (g0 - g1 + g2), g2};
#pragma unroll
            for (int x = 0; x < 4; ++x) uf[(((long)ky * 4 + x) * Cout + co) * Cin + ci] = u[x];
        }
        if (ud) {                                      // flipped taps: g'(kx') = g(2 - kx'), ky' = 2 - ky
            float u[4] = {g2, 0.5f * (g0 + g1 + g2), 0.5f * (g2 - g1 + g0), g0};
#pragma unroll
            for (int x = 0; x < 4; ++x) ud[(((long)(2 - ky) * 4 + x) * Cin + ci) * Cout + co] = u[x];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient in the Winograd domain:  dU_xi[ky][co][ci] = sum over output pairs t of dM_xi[t][co] * V_xi[t@ky][ci]
// with dM = (g0, g0+g1, g0-g1, -g1) from the output-gradient pair and V = (d0-d2, d1+d2, d2-d1, d1-d3) from the input
// row h+ky-1; the reduce kernel maps dU back to the three horizontal taps (dg0 = dU0 + (dU1+dU2)/2,
// dg1 = (dU1-dU2)/2, dg2 = (dU1+dU2)/2 + dU3).  12 MACs per pair instead of 18.  One workgroup = one ky, 64 co x 64 ci,
// all four positions (4 accumulators per wave), K = the pairs of a pixel slice, 64 pixels per K-step.
struct WWinoP {
    const float* x;
    const float* gy;
    float* partial;          // [nslices][3 ky][4 xi][N][K]
    const float* in_scale;
    const float* in_shift;
    int H, W, logW, K, N;
    long M;
    int pix_per_slice, ids_per_slice;
};
constexpr int WXROWS = 80;   // (64 / W) * (W + 2) <= 8 * 10

template <bool INT>
__global__ __launch_bounds__(256, 2) void wgrad_wino_kernel(WWinoP p) {
    __shared__ __attribute__((aligned(16))) float Gs[2][64 * 64];
    __shared__ __attribute__((aligned(16))) float Xs[2][WXROWS * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int W = p.W, W2 = W + 2;
    const int ci_tiles = p.K / 64;
    const int logical = xcd_remap_w(blockIdx.x, gridDim.x);
    const int slice = logical / p.ids_per_slice;
    int id = logical % p.ids_per_slice;
    const int ky = id % 3;
    id /= 3;
    const int ci0 = (id % ci_tiles) * 64, co0 = (id / ci_tiles) * 64;
    const int dy = ky - 1;
    const long pbeg = (long)slice * p.pix_per_slice;
    long pend = pbeg + p.pix_per_slice;
    if (pend > p.M) pend = p.M;

    for (int i = tid; i < 2 * WXROWS * 64 / 4; i += 256) reinterpret_cast<float4*>(&Xs[0][0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    floatx16 acc[4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    const int c4 = tid & 15, r0 = tid >> 4;            // staged pixel r0 + 16*i (i < 4), 16-B chunk c4
    float4 xsc = make_float4(0.f, 0.f, 0.f, 0.f), xsh = xsc;
    if (INT) {
        xsc = *reinterpret_cast<const float4*>(p.in_scale + ci0 + c4 * 4);
        xsh = *reinterpret_cast<const float4*>(p.in_shift + ci0 + c4 * 4);
    }
    const int dq = (64 >> p.logW) % p.H;               // image rows per K-step (mod H); 64 % W == 0 so w never changes
    const long x_off = (long)dy * W * p.K;
#define SED_WW_META(i)                                                                                          \
    long pm##i = pbeg + r0 + 16 * i;                                                                            \
    const float* gptr##i = p.gy + pm##i * p.N + co0 + c4 * 4;                                                   \
    const float* xptr##i = p.x + pm##i * p.K + ci0 + c4 * 4;                                                    \
    int xh##i = (int)(((unsigned)(pm##i < p.M ? pm##i : 0) >> p.logW) % (unsigned)p.H);                         \
    const int xslot##i = ((r0 + 16 * i) >> p.logW) * W2 + ((r0 + 16 * i) & (W - 1)) + 1;                        \
    float4 greg##i = make_float4(0.f, 0.f, 0.f, 0.f), xreg##i = greg##i;                                        \
    bool gv##i = false, xv##i = false;
    SED_WW_META(0) SED_WW_META(1) SED_WW_META(2) SED_WW_META(3)
#undef SED_WW_META
    const float* g_safe = p.gy + pbeg * p.N + co0 + c4 * 4;
    const float* x_safe = p.x + pbeg * p.K + ci0 + c4 * 4;
#define SED_WW_LOAD(i)                                                                                          \
    {                                                                                                           \
        gv##i = pm##i < pend;                                                                                   \
        xv##i = gv##i && (unsigned)(xh##i + dy) < (unsigned)p.H;                                                \
        greg##i = *reinterpret_cast<const float4*>(gv##i ? gptr##i : g_safe);                                   \
        xreg##i = *reinterpret_cast<const float4*>(xv##i ? xptr##i + x_off : x_safe);                           \
        gptr##i += 64L * p.N; xptr##i += 64L * p.K; pm##i += 64;                                                \
        xh##i += dq; xh##i = xh##i >= p.H ? xh##i - p.H : xh##i;                                                \
    }
#define SED_WW_PIN(r) asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
#define SED_WW_STORE(BUF, i)                                                                                    \
    {                                                                                                           \
        SED_WW_PIN(greg##i) SED_WW_PIN(xreg##i)                                                                 \
        float4 g = greg##i, v = xreg##i;                                                                        \
        g.x = gv##i ? g.x : 0.f; g.y = gv##i ? g.y : 0.f; g.z = gv##i ? g.z : 0.f; g.w = gv##i ? g.w : 0.f;     \
        if (INT) {                                                                                              \
            v.x = bn_relu(v.x, xsc.x, xsh.x); v.y = bn_relu(v.y, xsc.y, xsh.y);                                 \
            v.z = bn_relu(v.z, xsc.z, xsh.z); v.w = bn_relu(v.w, xsc.w, xsh.w);                                 \
        }                                                                                                       \
        v.x = xv##i ? v.x : 0.f; v.y = xv##i ? v.y : 0.f; v.z = xv##i ? v.z : 0.f; v.w = xv##i ? v.w : 0.f;     \
        *reinterpret_cast<float4*>(&Gs[(BUF)][(r0 + 16 * i) * 64 + c4 * 4]) = g;                                \
        *reinterpret_cast<float4*>(&Xs[(BUF)][xslot##i * 64 + c4 * 4]) = v;                                     \
    }
#define ww_load() { SED_WW_LOAD(0) SED_WW_LOAD(1) SED_WW_LOAD(2) SED_WW_LOAD(3) }
#define ww_store(BUF) { SED_WW_STORE(BUF, 0) SED_WW_STORE(BUF, 1) SED_WW_STORE(BUF, 2) SED_WW_STORE(BUF, 3) }

    const int nsteps = (int)((pend - pbeg + 63) / 64);
    __syncthreads();                                   // zero fill (halo columns) visible
    if (nsteps > 0) {
        ww_load();
        ww_store(0);
    }
    __syncthreads();
    const int half = lane >> 5;
    const int gcol = wm * 32 + (lane & 31), xcol = wn * 32 + (lane & 31);
    for (int it = 0; it < nsteps; ++it) {
        const int buf = it & 1;
        ww_load();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int px = 4 * j + 2 * half;           // first pixel of this lane-half's pair (within the 64 of the step)
            const int slot = (px >> p.logW) * W2 + (px & (W - 1));
            const float g0 = Gs[buf][px * 64 + gcol], g1 = Gs[buf][(px + 1) * 64 + gcol];
            const float* xp = &Xs[buf][slot * 64 + xcol];
            const float d0 = xp[0], d1 = xp[64], d2 = xp[128], d3 = xp[192];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, d0 - d2, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0 + g1, d1 + d2, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(g0 - g1, d2 - d1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(-g1, d1 - d3, acc[3], 0, 0, 0);
            if (j == 11) {
                __builtin_amdgcn_sched_barrier(0);
                ww_store(buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
#undef SED_WW_LOAD
#undef SED_WW_PIN
#undef SED_WW_STORE
#undef ww_load
#undef ww_store
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        float* out = p.partial + (((long)slice * 3 + ky) * 4 + x) * p.N * p.K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            int ci = ci0 + wn * 32 + (lane & 31);
            out[(long)co * p.K + ci] = acc[x][r];
        }
    }
}

// sum the slices in fp64, map the four Winograd positions back to the three horizontal taps, scatter to OIHW
__global__ __launch_bounds__(256) void wgrad_wino_reduce_kernel(const float* __restrict__ partial, int nslices, int N,
                                                                int K, float* __restrict__ out) {
    const long nk = (long)N * K;
    const long total = 3 * nk;
    const long per_slice = 12 * nk;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int ky = (int)(i / nk);
        long e = i % nk;                               // co * K + ci
        double u[4] = {0.0, 0.0, 0.0, 0.0};
        for (int sl = 0; sl < nslices; ++sl) {
            const float* q = partial + (long)sl * per_slice + ((long)ky * 4) * nk + e;
            u[0] += (double)q[0]; u[1] += (double)q[nk]; u[2] += (double)q[2 * nk]; u[3] += (double)q[3 * nk];
        }
        float* o = out + e * 9 + ky * 3;
        o[0] = (float)(u[0] + 0.5 * (u[1] + u[2]));
        o[1] = (float)(0.5 * (u[1] - u[2]));
        o[2] = (float)(0.5 * (u[1] + u[2]) + u[3]);
    }
}

}  // namespace

// 1 if the Winograd kernel supports this layer shape (W even and dividing 128, channel multiples), else 0.
SED_API int sed_conv3x3_wino_supported(int H, int W, int Cin, int Cout) {
    return (H > 0 && W >= 2 && W <= 64 && (W & 1) == 0 && 128 % W == 0 && (128 / W) * (W + 2) <= WMAXROWS && Cin % WBK == 0 &&
            Cout % 64 == 0) ? 1 : 0;
}

SED_API int sed_pack_conv_weights_wino(const float* w_oihw, int Cout, int Cin, float* uf, float* ud, hipStream_t stream) {
    long total = (long)Cout * Cin * 3;
    hipLaunchKernelGGL(pack_wino_kernel, dim3(sed_cdiv(total, 256) > 2048 ? 2048 : sed_cdiv(total, 256)), dim3(256), 0, stream,
                       w_oihw, Cout, Cin, uf, ud);
    SED_LAUNCH_CHECK();
    return 0;
}

// Same contract as sed_conv3x3_igemm with w_packed = the Winograd pack [3][4][Cout][Cin]; statistics partials cover
// 64 rows each: [ceil(M/128)*2][2][Cout].
SED_API int sed_conv3x3_wino(const float* x, const float* w_wino, float* y, int B, int H, int W, int Cin, int Cout,
                             const float* in_scale, const float* in_shift, int epi, float* partials, const float* yprev,
                             const float* p_scale, const float* p_shift, const float* p_mean, const float* p_invstd,
                             hipStream_t stream) {
    if (B <= 0 || !sed_conv3x3_wino_supported(H, W, Cin, Cout) || (long)B * H * W >= (1L << 31)) return SED_EINVAL;
    WinoP p{x, w_wino, y, in_scale, in_shift, partials, yprev, p_scale, p_shift, p_mean, p_invstd, H, W, Cin, Cout, (long)B * H * W};
    dim3 grid((unsigned)(sed_cdiv(p.M, 128) * (Cout / 64))), block(256);
    bool in_t = in_scale != nullptr;
#define SED_WL(INT_, EPI_) hipLaunchKernelGGL((conv_wino_kernel<INT_, EPI_>), grid, block, 0, stream, p)
    if (in_t) { if (epi == 0) SED_WL(true, 0); else if (epi == 1) SED_WL(true, 1); else return SED_EINVAL; }
    else { if (epi == 0) SED_WL(false, 0); else if (epi == 1) SED_WL(false, 1); else if (epi == 2) SED_WL(false, 2); else return SED_EINVAL; }
#undef SED_WL
    SED_LAUNCH_CHECK();
    return 0;
}

// Pixel slices of the Winograd wgrad (multiples of 64 pixels, <= 16384 per slice, grid = whole rounds of 512 workgroups).
SED_API long sed_wgrad_wino_partial_floats(long M, int Cin, int Cout, int* nslices_out, int* pix_per_slice_out) {
    const long tiles = 3L * (Cout / 64) * (Cin / 64);
    const long capacity = 512;
    long ns_min = (M + 16383) / 16384;
    long fill = (2 * capacity + tiles - 1) / tiles;
    if (fill > ns_min) ns_min = fill;
    long rounds = (tiles * ns_min + capacity - 1) / capacity;
    long ns = rounds * capacity / tiles;
    if (ns < ns_min) ns = ns_min;
    long pps = ((M + ns - 1) / ns + 63) / 64 * 64;
    if (pps < 256) pps = 256;
    ns = (M + pps - 1) / pps;
    if (nslices_out) *nslices_out = (int)ns;
    if (pix_per_slice_out) *pix_per_slice_out = (int)pps;
    return ns * 12L * Cin * Cout;
}

// dW (OIHW) via the Winograd domain; same contract as sed_conv3x3_wgrad.  partial: sed_wgrad_wino_partial_floats floats.
// Additionally needs W to be a power of two dividing 64 and Cin % 64 == 0.
SED_API int sed_conv3x3_wgrad_wino(const float* x, const float* gy, float* dw_oihw, float* partial, int B, int H, int W,
                                   int Cin, int Cout, const float* in_scale, const float* in_shift, hipStream_t stream) {
    int logW = 0;
    while ((1 << logW) < W) ++logW;
    if (B <= 0 || (1 << logW) != W || W < 2 || W > 64 || (64 / W) * (W + 2) > WXROWS || Cin % 64 != 0 || Cout % 64 != 0 ||
        (long)B * H * W >= (1L << 31))
        return SED_EINVAL;
    long M = (long)B * H * W;
    int ns, pps;
    sed_wgrad_wino_partial_floats(M, Cin, Cout, &ns, &pps);
    WWinoP p{x, gy, partial, in_scale, in_shift, H, W, logW, Cin, Cout, M, pps, 3 * (Cout / 64) * (Cin / 64)};
    dim3 grid((unsigned)((long)p.ids_per_slice * ns)), block(256);
    if (in_scale) hipLaunchKernelGGL((wgrad_wino_kernel<true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((wgrad_wino_kernel<false>), grid, block, 0, stream, p);
    long total = 3L * Cin * Cout;
    hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3(sed_cdiv(total, 256) > 4096 ? 4096 : sed_cdiv(total, 256)), dim3(256), 0,
                       stream, partial, ns, Cout, Cin, dw_oihw);
    SED_LAUNCH_CHECK();
    return 0;
}

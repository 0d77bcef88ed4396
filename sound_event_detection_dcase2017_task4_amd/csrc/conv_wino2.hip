// 3x3 convolution with a FUSED 2-D Winograd F(2x2,3x3), fp32 MFMA: 16 multiplies per 2x2 output tile and (ci, co)
// pair instead of 36 (the 1-D kernel in conv_wino.hip needs 24).
//
//     U = G g G^T (4x4 per (co, ci), packed once per step),   V = B^T d B (4x4 input patch, formed in registers),
//     M[eta][xi] = sum_ci V[eta][xi] * U[eta][xi],            Y = A^T M A (2x2 outputs)
//     B^T rows: (d0-d2, d1+d2, d2-d1, d1-d3);  G rows: (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2);  A^T = [1 1 1 0; 0 1 -1 -1]
//
// Work split.  A workgroup owns 64 tiles (RP consecutive row pairs of ONE image x all W/2 tile columns) x 32 output
// channels.  The 16 Winograd coordinates of a tile would need 16 accumulators (256 registers at a 32x32 MFMA tile), so
// the four waves are 2 tile blocks x 2 ETA HALVES: a wave accumulates eta in {0,1} or {2,3} for its 32 tiles (8
// accumulators = 128 registers, 2 waves/SIMD), and the halves are combined once, in the epilogue, through LDS: the
// eta-half-0 wave finishes output row 2*th, the other one row 2*th+1.
//
// LDS.  K-step = 8 input channels.  A stage: the (2*RP+2) raw input rows of the block, one 32-byte LDS row per pixel,
// columns split into an even and an odd plane (so the tiles of a row pair are CONSECUTIVE LDS rows for each of the four
// patch columns) with a zero halo entry at either end; 16-byte chunk index XOR ((row >> 3) & 1) makes every
// ds_read_b128 quarter-wave conflict-free.  B stage: U[16][32 co][8 ci] by LDS-DMA from the k-step-major pack
// (16 KB contiguous per (k-step, co block)), same swizzle applied on the source side.
//
// Fusions as in conv_wino.hip: input relu(scale*x+shift); epilogue 1 = BN statistics (sum, M2) per wave (64 pixels)
// + the per-part pixel count (tiles past the image edge are not counted); epilogue 2 = ReLU mask + BN-backward sums.
#include "common.h"
#include "sed_hip.h"

namespace {

constexpr int W2_AROWS = 400;                 // >= (2*RP+2) * 2 * S for every supported W (max 396 at W = 64)
constexpr int W2_ASTAGE = W2_AROWS * 8;       // floats
constexpr int W2_BSTAGE = 16 * 32 * 8;        // floats

struct Wino2P {
    const float* x;          // [B][H][W][K]
    const float* wu;         // [K/8][16][N][8]
    float* y;                // [B][H][W][N]
    const float* in_scale;
    const float* in_shift;
    float* partials;         // EPI 1/2: [nparts][2][N] (+ [nparts] counts for EPI 1), nparts = tile blocks * 4
    const float* yprev;
    const float* p_scale;
    const float* p_shift;
    const float* p_mean;
    const float* p_invstd;
    int B, H, W, K, N;
    int logW, RP, nrb, S;
    long nparts;
};

__device__ __forceinline__ int xcd_remap2(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ void lds_dma16_2(const float* gsrc, unsigned lds_dst_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_bytes) : "memory");
}

// float offset of 16-byte chunk `chunk` (0/1) of 32-byte LDS row `row`
__device__ __forceinline__ int sw_off(int row, int chunk) { return row * 8 + ((chunk ^ ((row >> 3) & 1)) << 2); }

__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 b) {
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}

template <bool INT, int EPI>
__global__ __launch_bounds__(256, 2) void conv_wino2_kernel(Wino2P p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * W2_ASTAGE + 2 * W2_BSTAGE];
    float* const As = smem;                          // [2][W2_ASTAGE]
    float* const Bs = smem + 2 * W2_ASTAGE;          // [2][W2_BSTAGE]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int eh = wvu & 1, tb = wvu >> 1;
    const int nb = p.N >> 5;
    const int logical = xcd_remap2(blockIdx.x, gridDim.x);
    const int n0 = (logical % nb) * 32;
    const int tblk = logical / nb;
    const int b = tblk / p.nrb, rb = tblk % p.nrb;
    const int W = p.W, logW = p.logW, logTW = logW - 1, TW = W >> 1, S = p.S, RP = p.RP;
    const int th0 = rb * RP;
    const int hbase = 2 * th0 - 1;                   // image row of block row 0
    const int nrows = 2 * RP + 2;

    // ---- A staging: item e = tid + 256*i -> pixel e >> 1 of the block's rows, 16-byte chunk e & 1
    const int c2 = tid & 1;
#define SED_W2META(i)                                                                                           \
    bool sok##i;                                                                                                \
    int lso##i;                                                                                                 \
    const float* aptr##i;                                                                                       \
    float4 areg##i = make_float4(0.f, 0.f, 0.f, 0.f);                                                           \
    {                                                                                                           \
        const int pix = (tid + 256 * i) >> 1;                                                                   \
        const int r = pix >> logW, w = pix & (W - 1);                                                           \
        const int h = hbase + r;                                                                                \
        sok##i = r < nrows && (unsigned)h < (unsigned)p.H;                                                      \
        lso##i = sw_off((r * 2 + ((w + 1) & 1)) * S + ((w + 1) >> 1), c2);                                      \
        aptr##i = p.x + (sok##i ? (((long)b * p.H + h) * W + w) * p.K : 0L) + c2 * 4;                           \
    }
    SED_W2META(0) SED_W2META(1) SED_W2META(2)
#undef SED_W2META
    // ---- B DMA: wave wv stages coordinates 4*wv .. 4*wv+3, one instruction (64 lanes x 16 B = 32 rows) each
    const int brow_in = lane >> 1;
    const float* bptr = p.wu + ((long)(wvu * 4) * p.N + n0 + brow_in) * 8 + (((lane & 1) ^ ((brow_in >> 3) & 1)) << 2);
    const long b_xi_stride = (long)p.N * 8;          // next Winograd coordinate
    const long b_k_stride = 16L * p.N * 8;           // next k-step
    const unsigned bs_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)Bs);

    floatx16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int KT = p.K >> 3;
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;

#define SED_W2A_LOAD(i) areg##i = *reinterpret_cast<const float4*>(aptr##i + (sok##i ? a_off : 0));
#define SED_W2B_LOAD(DST, j) lds_dma16_2(bptr + b_off + (j) * b_xi_stride, bs_base + (unsigned)(((DST) * W2_BSTAGE + (wvu * 4 + (j)) * 256) * 4));
#define w2gload(IT, DST)                                                                                        \
    {                                                                                                           \
        const int a_off = (IT) * 8;                                                                             \
        const long b_off = (long)(IT) * b_k_stride;                                                             \
        if (INT) {                                                                                              \
            sc = *reinterpret_cast<const float4*>(p.in_scale + a_off + c2 * 4);                                 \
            sh = *reinterpret_cast<const float4*>(p.in_shift + a_off + c2 * 4);                                 \
        }                                                                                                       \
        SED_W2B_LOAD(DST, 0) SED_W2B_LOAD(DST, 1) SED_W2B_LOAD(DST, 2) SED_W2B_LOAD(DST, 3)                     \
        SED_W2A_LOAD(0) SED_W2A_LOAD(1) SED_W2A_LOAD(2)                                                         \
    }
#define SED_W2A_STORE(BUF, i)                                                                                   \
    {                                                                                                           \
        asm volatile("" : "+v"(areg##i.x), "+v"(areg##i.y), "+v"(areg##i.z), "+v"(areg##i.w));                  \
        float4 v = areg##i;                                                                                     \
        if (INT) {                                                                                              \
            v.x = bn_relu(v.x, sc.x, sh.x); v.y = bn_relu(v.y, sc.y, sh.y);                                     \
            v.z = bn_relu(v.z, sc.z, sh.z); v.w = bn_relu(v.w, sc.w, sh.w);                                     \
        }                                                                                                       \
        if (sok##i) *reinterpret_cast<float4*>(&As[(BUF) * W2_ASTAGE + lso##i]) = v;                            \
    }
#define w2lstore(BUF) { SED_W2A_STORE(BUF, 0) SED_W2A_STORE(BUF, 1) SED_W2A_STORE(BUF, 2) }

    // halo entries and rows outside the image are never stored: zero both A stages once
    for (int i = tid; i < 2 * W2_ASTAGE / 4; i += 256) reinterpret_cast<float4*>(As)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();                                   // zero fill visible before the first stores
    w2gload(0, 0);
    w2lstore(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- fragment addressing.  Lane (i = lane & 31) owns tile t = tb*32 + i of the block; k half = lane >> 5.
    const int chunk = lane >> 5;
    const int t = tb * 32 + (lane & 31);
    const int rp_l = t >> logTW, tw_l = t & (TW - 1);
    // eta half 0 combines rows (0,2) and (1,2) of the patch; half 1 rows (2,1) and (1,3):  eta_a = ra - rc,
    // eta_b = rc + sgn*rb with sgn = +1 / -1.
    const int ra = 2 * rp_l + (eh ? 2 : 0), rbw = 2 * rp_l + (eh ? 3 : 1), rc = 2 * rp_l + (eh ? 1 : 2);
    const float sgn = eh ? -1.f : 1.f;
    int oa[4], ob[4], oc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        oa[j] = sw_off((ra * 2 + (j & 1)) * S + tw_l + (j >> 1), chunk);
        ob[j] = sw_off((rbw * 2 + (j & 1)) * S + tw_l + (j >> 1), chunk);
        oc[j] = sw_off((rc * 2 + (j & 1)) * S + tw_l + (j >> 1), chunk);
    }
    const int bo = eh * 8 * 256 + sw_off(lane & 31, chunk);          // + (e*4 + xi) * 256

    for (int it = 0; it < KT; ++it) {
        const int buf = it & 1;
        const float* Ab = As + buf * W2_ASTAGE;
        const float* Bb = Bs + buf * W2_BSTAGE + bo;
        w2gload(it + 1 < KT ? it + 1 : it, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        float4 xa[4], xb[4], xc[4], bf[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xa[j] = *reinterpret_cast<const float4*>(Ab + oa[j]);
            xc[j] = *reinterpret_cast<const float4*>(Ab + oc[j]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) bf[a] = *reinterpret_cast<const float4*>(Bb + a * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) xb[j] = *reinterpret_cast<const float4*>(Ab + ob[j]);
#pragma unroll
        for (int a = 4; a < 8; ++a) bf[a] = *reinterpret_cast<const float4*>(Bb + a * 256);
#define SED_W2MMA(A_, VV)                                                                                       \
    acc[A_] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.x, bf[A_].x, acc[A_], 0, 0, 0);                            \
    acc[A_] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.y, bf[A_].y, acc[A_], 0, 0, 0);                            \
    acc[A_] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.z, bf[A_].z, acc[A_], 0, 0, 0);                            \
    acc[A_] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.w, bf[A_].w, acc[A_], 0, 0, 0);
        {
            float4 c0 = f4sub(xa[0], xc[0]), c1 = f4sub(xa[1], xc[1]), c2_ = f4sub(xa[2], xc[2]), c3 = f4sub(xa[3], xc[3]);
            float4 v0 = f4sub(c0, c2_), v1 = f4add(c1, c2_), v2 = f4sub(c2_, c1), v3 = f4sub(c1, c3);
            SED_W2MMA(0, v0) SED_W2MMA(1, v1) SED_W2MMA(2, v2) SED_W2MMA(3, v3)
        }
        {
            float4 c0 = f4fma(sgn, xb[0], xc[0]), c1 = f4fma(sgn, xb[1], xc[1]), c2_ = f4fma(sgn, xb[2], xc[2]),
                   c3 = f4fma(sgn, xb[3], xc[3]);
            float4 v0 = f4sub(c0, c2_), v1 = f4add(c1, c2_), v2 = f4sub(c2_, c1), v3 = f4sub(c1, c3);
            SED_W2MMA(4, v0) SED_W2MMA(5, v1) SED_W2MMA(6, v2) SED_W2MMA(7, v3)
        }
#undef SED_W2MMA
        __builtin_amdgcn_sched_barrier(0);
        w2lstore(buf ^ 1);
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef w2gload
#undef w2lstore
#undef SED_W2A_LOAD
#undef SED_W2B_LOAD
#undef SED_W2A_STORE

    // ---- epilogue.  z[e][q] = column inverse transform of this wave's two eta rows; the wave keeps
    // mine[q] = +-(z[0][q] + z[1][q]) and hands z[1] (half 0) / z[0] (half 1) to its partner wave through LDS
    // (the staging buffers are free after the loop's last barrier).
    float* xch = smem;                                 // [4 waves][32][64 lanes]
    float mine[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float z00 = acc[0][r] + acc[1][r] + acc[2][r], z01 = acc[1][r] - acc[2][r] - acc[3][r];
        float z10 = acc[4][r] + acc[5][r] + acc[6][r], z11 = acc[5][r] - acc[6][r] - acc[7][r];
        float g0 = eh ? z00 : z10, g1 = eh ? z01 : z11;
        xch[(wvu * 32 + 2 * r) * 64 + lane] = g0;
        xch[(wvu * 32 + 2 * r + 1) * 64 + lane] = g1;
        mine[2 * r] = eh ? -(z00 + z10) : (z00 + z10);
        mine[2 * r + 1] = eh ? -(z01 + z11) : (z01 + z11);
    }
    const int half = lane >> 5;
    const int col = n0 + (lane & 31);
    float s1 = 0.f, s2 = 0.f, cnt = 0.f;
    float e_sc = 0.f, e_sh = 0.f, e_mu = 0.f, e_is = 0.f;
    float yp[EPI == 2 ? 32 : 1];
    if (EPI == 2) {                                    // previous-layer activations: loads overlap the LDS exchange
        e_sc = p.p_scale[col]; e_sh = p.p_shift[col]; e_mu = p.p_mean[col]; e_is = p.p_invstd[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tl = tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int h = 2 * (th0 + (tl >> logTW)) + eh;
            const long pix = ((long)b * p.H + (h < p.H ? h : 0)) * W + 2 * (tl & (TW - 1));
            yp[2 * r] = p.yprev[pix * p.N + col];
            yp[2 * r + 1] = p.yprev[(pix + 1) * p.N + col];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tl = tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int h = 2 * (th0 + (tl >> logTW)) + eh;
        const bool ok = h < p.H;
        const long pix = ((long)b * p.H + h) * W + 2 * (tl & (TW - 1));
        float y0 = mine[2 * r] + xch[((wvu ^ 1) * 32 + 2 * r) * 64 + lane];
        float y1 = mine[2 * r + 1] + xch[((wvu ^ 1) * 32 + 2 * r + 1) * 64 + lane];
        if (EPI == 2 && ok) {
            float a0 = yp[(EPI == 2 ? 2 * r : 0)], a1 = yp[(EPI == 2 ? 2 * r + 1 : 0)];
            y0 = bn_relu_active(a0, e_sc, e_sh) ? y0 : 0.f;
            y1 = bn_relu_active(a1, e_sc, e_sh) ? y1 : 0.f;
            s1 += y0 + y1;
            s2 = fmaf(y0, (a0 - e_mu) * e_is, s2);
            s2 = fmaf(y1, (a1 - e_mu) * e_is, s2);
        }
        if (ok) {
            p.y[pix * p.N + col] = y0;
            p.y[(pix + 1) * p.N + col] = y1;
            if (EPI == 1) { s1 += y0 + y1; cnt += 2.f; }
        }
        mine[2 * r] = y0; mine[2 * r + 1] = y1;
    }
    if (EPI == 1) {
        s1 += __shfl_xor(s1, 32, 64);
        cnt += __shfl_xor(cnt, 32, 64);
        const float mean = cnt > 0.f ? s1 / cnt : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tl = tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (2 * (th0 + (tl >> logTW)) + eh < p.H) {
                float da = mine[2 * r] - mean, db = mine[2 * r + 1] - mean;
                s2 = fmaf(da, da, s2);
                s2 = fmaf(db, db, s2);
            }
        }
        s2 += __shfl_xor(s2, 32, 64);
    }
    if (EPI == 2) { s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64); }
    if ((EPI == 1 || EPI == 2) && half == 0) {
        const long part = (long)tblk * 4 + wvu;
        p.partials[(part * 2 + 0) * p.N + col] = s1;
        p.partials[(part * 2 + 1) * p.N + col] = s2;
        if (EPI == 1 && n0 == 0 && lane == 0) p.partials[p.nparts * 2 * p.N + part] = cnt;
    }
}

// OIHW -> k-step-major 2-D Winograd packs.  Forward operand uf[ci/8][eta*4+xi][co][ci%8] from g = W[co][ci][.][.];
// dgrad operand ud[co/8][eta*4+xi][ci][co%8] from the tap-flipped g'[ky][kx] = W[co][ci][2-ky][2-kx].
__global__ __launch_bounds__(256) void pack_wino2_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                         float* __restrict__ uf, float* __restrict__ ud) {
    const long total = (long)Cout * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ci = (int)(i % Cin), co = (int)(i / Cin);
        const float* g = w + i * 9;
#pragma unroll
        for (int flip = 0; flip < 2; ++flip) {
            float* dst = flip ? ud : uf;
            if (!dst) continue;
            float t[4][3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float g0 = g[flip ? (2 - 0) * 3 + (2 - kx) : 0 * 3 + kx];
                float g1 = g[flip ? (2 - 1) * 3 + (2 - kx) : 1 * 3 + kx];
                float g2 = g[flip ? (2 - 2) * 3 + (2 - kx) : 2 * 3 + kx];
                t[0][kx] = g0; t[1][kx] = 0.5f * (g0 + g1 + g2); t[2][kx] = 0.5f * (g0 - g1 + g2); t[3][kx] = g2;
            }
#pragma unroll
            for (int eta = 0; eta < 4; ++eta) {
                float u[4] = {t[eta][0], 0.5f * (t[eta][0] + t[eta][1] + t[eta][2]), 0.5f * (t[eta][0] - t[eta][1] + t[eta][2]),
                              t[eta][2]};
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    if (!flip) uf[(((long)(ci >> 3) * 16 + eta * 4 + xi) * Cout + co) * 8 + (ci & 7)] = u[xi];
                    else ud[(((long)(co >> 3) * 16 + eta * 4 + xi) * Cin + ci) * 8 + (co & 7)] = u[xi];
                }
            }
        }
    }
}

__host__ bool wino2_geometry(int H, int W, int* logW, int* RP, int* nrb, int* S) {
    int lw = 0;
    while ((1 << lw) < W) ++lw;
    if ((1 << lw) != W || W < 8 || W > 64 || H < 1) return false;
    const int TW = W / 2, rp = 64 / TW, TH = (H + 1) / 2;
    *logW = lw; *RP = rp; *nrb = (TH + rp - 1) / rp; *S = (TW == 8) ? 10 : TW + 1;
    return (2 * rp + 2) * 2 * (*S) <= W2_AROWS;
}

}  // namespace

// 1 if the 2-D Winograd kernel supports this layer shape (W in {8,16,32,64}, Cin % 8 == 0, Cout % 32 == 0), else 0.
SED_API int sed_conv3x3_wino2_supported(int H, int W, int Cin, int Cout) {
    int a, b, c, d;
    return (wino2_geometry(H, W, &a, &b, &c, &d) && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 32 == 0) ? 1 : 0;
}

// Statistics parts written by epilogues 1/2: one per wave = 4 per (image, row-pair block); 0 if unsupported.
SED_API long sed_conv_wino2_num_parts(int B, int H, int W) {
    int lw, rp, nrb, s;
    if (B <= 0 || !wino2_geometry(H, W, &lw, &rp, &nrb, &s)) return 0;
    return (long)B * nrb * 4;
}

SED_API int sed_pack_conv_weights_wino2(const float* w_oihw, int Cout, int Cin, float* uf, float* ud, hipStream_t stream) {
    if (Cout <= 0 || Cin <= 0 || (uf && Cin % 8) || (ud && Cout % 8)) return SED_EINVAL;
    long total = (long)Cout * Cin;
    hipLaunchKernelGGL(pack_wino2_kernel, dim3(sed_cdiv(total, 256) > 2048 ? 2048 : sed_cdiv(total, 256)), dim3(256), 0, stream,
                       w_oihw, Cout, Cin, uf, ud);
    SED_LAUNCH_CHECK();
    return 0;
}

// Same contract as sed_conv3x3_igemm with w_packed = the 2-D Winograd pack [Cin/8][16][Cout][8].  partials (epi 1/2):
// [P][2][Cout] floats with P = sed_conv_wino2_num_parts(B,H,W); epi 1 appends P per-part pixel counts after them
// (P*2*Cout + P floats in total) because parts at the bottom edge of an image hold fewer pixels.
SED_API int sed_conv3x3_wino2(const float* x, const float* w_wino2, float* y, int B, int H, int W, int Cin, int Cout,
                              const float* in_scale, const float* in_shift, int epi, float* partials, const float* yprev,
                              const float* p_scale, const float* p_shift, const float* p_mean, const float* p_invstd,
                              hipStream_t stream) {
    int lw, rp, nrb, s;
    if (B <= 0 || !sed_conv3x3_wino2_supported(H, W, Cin, Cout) || !wino2_geometry(H, W, &lw, &rp, &nrb, &s) ||
        (long)B * H * W >= (1L << 31) || (long)B * nrb * (Cout / 32) >= (1L << 31))
        return SED_EINVAL;
    Wino2P p{x, w_wino2, y, in_scale, in_shift, partials, yprev, p_scale, p_shift, p_mean, p_invstd,
             B, H, W, Cin, Cout, lw, rp, nrb, s, (long)B * nrb * 4};
    dim3 grid((unsigned)((long)B * nrb * (Cout / 32))), block(256);
    bool in_t = in_scale != nullptr;
#define SED_W2L(INT_, EPI_) hipLaunchKernelGGL((conv_wino2_kernel<INT_, EPI_>), grid, block, 0, stream, p)
    if (in_t) { if (epi == 0) SED_W2L(true, 0); else if (epi == 1) SED_W2L(true, 1); else return SED_EINVAL; }
    else { if (epi == 0) SED_W2L(false, 0); else if (epi == 1) SED_W2L(false, 1); else if (epi == 2) SED_W2L(false, 2); else return SED_EINVAL; }
#undef SED_W2L
    SED_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient in the 2-D Winograd domain:
//     dU[eta][xi][co][ci] = sum over tiles of dM[eta][xi][tile][co] * V[eta][xi][tile][ci],
//     dM = A dY A^T (4x4 from the 2x2 output-gradient tile, A rows (1,0),(1,1),(1,-1),(0,-1)),  V = B^T d B as above,
// and the reduce kernel maps dU back to the taps, dg = G^T dU G.  16 MACs per tile instead of 36.
// Workgroup = 64 co x 32 ci, four waves = 2 co blocks x 2 eta halves (8 accumulators each); K = tiles, streamed in
// units of 16 tiles (RPK row pairs x TCK tile columns = 64 output pixels): the raw gy pixels [64][64 co] and the raw
// x patch region [(2*RPK+2) x (2*TCK+2)][32 ci] are staged per unit, both operands are transformed in registers from
// scalar LDS reads (lane = channel).  The two lane halves of an MFMA read adjacent tile columns, i.e. addresses two
// pixels apart; a 32-float swizzle keyed on the tile-column parity keeps them on different banks.
namespace {

struct WWino2P {
    const float* x;          // [B][H][W][K]
    const float* gy;         // [B][H][W][N]
    float* partial;          // [nslices][16][N][K]
    const float* in_scale;
    const float* in_shift;
    int B, H, W, K, N;
    int nrg, nsg;            // unit grid per image: row-pair groups x tile-column segments
    long U;                  // units in total
    int units_per_slice, ids_per_slice;
};

template <bool INT, bool W8>
__global__ __launch_bounds__(256, 2) void wgrad_wino2_kernel(WWino2P p) {
    constexpr int TCK = W8 ? 4 : 8, RPK = 16 / TCK, CW = 2 * TCK + 2, XR = 2 * RPK + 2, GW = 2 * TCK;
    constexpr int XPIX = XR * CW;                      // 100 / 108 patch pixels
    constexpr int XSTAGE = ((XPIX + 1) / 2) * 64;      // floats: two pixels per 64-float line
    constexpr int GSTAGE = 64 * 64;
    __shared__ __attribute__((aligned(16))) float Gs[2][GSTAGE];
    __shared__ __attribute__((aligned(16))) float Xs[2][XSTAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int eh = wvu & 1, cob = wvu >> 1;
    const int logical = xcd_remap2(blockIdx.x, gridDim.x);
    const int slice = logical / p.ids_per_slice;
    const int id = logical % p.ids_per_slice;
    const int ci_tiles = p.K >> 5;
    const int ci0 = (id % ci_tiles) * 32, co0 = (id / ci_tiles) * 64;
    const int W = p.W;

    long u = (long)slice * p.units_per_slice;
    long uend = u + p.units_per_slice;
    if (uend > p.U) uend = p.U;
    const int nsteps = (int)(uend > u ? uend - u : 0);
    int sg = (int)(u % p.nsg);
    long tq = u / p.nsg;
    int rg = (int)(tq % p.nrg);
    int b = (int)(tq / p.nrg);

    floatx16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // ---- staging maps (fixed per thread): x item e = tid + 256*i -> patch pixel e >> 3, chunk e & 7;
    //      gy item e -> pixel e >> 4 of the 64, chunk e & 15
    const int xch = tid & 7, gch = tid & 15;
    float4 xsc = make_float4(0.f, 0.f, 0.f, 0.f), xsh = xsc;
    if (INT) {
        xsc = *reinterpret_cast<const float4*>(p.in_scale + ci0 + xch * 4);
        xsh = *reinterpret_cast<const float4*>(p.in_shift + ci0 + xch * 4);
    }
#define SED_WW2_META(i)                                                                                         \
    const int xpx##i = (tid + 256 * i) >> 3;                                                                    \
    const bool xit##i = xpx##i < XPIX;                                                                          \
    const int xr##i = xpx##i / CW, xc##i = xpx##i % CW;                                                         \
    const int xls##i = (xpx##i >> 1) * 64 + 32 * ((xpx##i & 1) ^ ((xc##i >> 1) & 1)) + xch * 4;                 \
    const int gpx##i = (tid + 256 * i) >> 4;                                                                    \
    const int gr##i = gpx##i / GW, gc##i = gpx##i % GW;                                                         \
    const int gls##i = gpx##i * 64 + ((gch * 4) ^ (32 * ((gc##i >> 1) & 1)));                                   \
    const int xoff##i = (xr##i * W + xc##i) * p.K + ci0 + xch * 4;                                              \
    const int goff##i = (gr##i * W + gc##i) * p.N + co0 + gch * 4;                                              \
    float4 xreg##i = make_float4(0.f, 0.f, 0.f, 0.f), greg##i = xreg##i;                                        \
    bool xv##i = false, gv##i = false;
    SED_WW2_META(0) SED_WW2_META(1) SED_WW2_META(2) SED_WW2_META(3)
#undef SED_WW2_META
    const float* x_safe = p.x + ci0 + xch * 4;
    const float* g_safe = p.gy + co0 + gch * 4;

#define SED_WW2_LOAD(i)                                                                                         \
    {                                                                                                           \
        const int h = h0 + xr##i, w = w0 + xc##i;                                                               \
        xv##i = live && xit##i && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)W;                     \
        xreg##i = *reinterpret_cast<const float4*>(xv##i ? x_unit + xoff##i : x_safe);                         \
        const int hg = h0 + 1 + gr##i;                                                                          \
        gv##i = live && hg < p.H;                                                                               \
        greg##i = *reinterpret_cast<const float4*>(gv##i ? g_unit + goff##i : g_safe);                         \
    }
#define ww2_load(LIVE)                                                                                          \
    {                                                                                                           \
        const bool live = (LIVE);                                                                               \
        const int h0 = 2 * rg * RPK - 1, w0 = 2 * sg * TCK - 1;                                                 \
        const float* x_unit = p.x + (((long)b * p.H + h0) * W + w0) * p.K;          /* uniform; only valid offsets used */ \
        const float* g_unit = p.gy + (((long)b * p.H + h0 + 1) * W + w0 + 1) * p.N;                             \
        SED_WW2_LOAD(0) SED_WW2_LOAD(1) SED_WW2_LOAD(2) SED_WW2_LOAD(3)                                         \
        if (++sg == p.nsg) { sg = 0; if (++rg == p.nrg) { rg = 0; ++b; } }                                      \
    }
#define SED_WW2_PIN(r) asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
#define SED_WW2_STORE(BUF, i)                                                                                   \
    {                                                                                                           \
        SED_WW2_PIN(greg##i) SED_WW2_PIN(xreg##i)                                                               \
        float4 g = greg##i, v = xreg##i;                                                                        \
        g.x = gv##i ? g.x : 0.f; g.y = gv##i ? g.y : 0.f; g.z = gv##i ? g.z : 0.f; g.w = gv##i ? g.w : 0.f;     \
        if (INT) {                                                                                              \
            v.x = bn_relu(v.x, xsc.x, xsh.x); v.y = bn_relu(v.y, xsc.y, xsh.y);                                 \
            v.z = bn_relu(v.z, xsc.z, xsh.z); v.w = bn_relu(v.w, xsc.w, xsh.w);                                 \
        }                                                                                                       \
        v.x = xv##i ? v.x : 0.f; v.y = xv##i ? v.y : 0.f; v.z = xv##i ? v.z : 0.f; v.w = xv##i ? v.w : 0.f;     \
        *reinterpret_cast<float4*>(&Gs[(BUF)][gls##i]) = g;                                                     \
        if (xit##i) *reinterpret_cast<float4*>(&Xs[(BUF)][xls##i]) = v;                                         \
    }
#define ww2_store(BUF) { SED_WW2_STORE(BUF, 0) SED_WW2_STORE(BUF, 1) SED_WW2_STORE(BUF, 2) SED_WW2_STORE(BUF, 3) }

    ww2_load(nsteps > 0);
    ww2_store(0);
    __syncthreads();

    // ---- fragment addressing: lane = channel (lane & 31), lane half = second tile of the k pair (next tile column)
    const int half = lane >> 5, cl = lane & 31;
    const int gbase = half * 128 + ((cob * 32 + cl) ^ (32 * half));
    const int xbase0 = half * 64 + 32 * half + cl;          // patch columns with (jj & 1) ^ (jj >> 1) == 0
    const int xbase1 = half * 64 + 32 * (1 ^ half) + cl;    // ... == 1
    // eta half 0: dM rows (d0, d0 + d1), V rows (r0 - r2, r1 + r2); half 1: (d0 - d1, -d1), (r2 - r1, r1 - r3)
    const float al = eh ? -1.f : 0.f, be = eh ? 0.f : 1.f, ga = eh ? -1.f : 1.f, sgn = eh ? -1.f : 1.f;
    const int ra = (eh ? 2 : 0) * (CW / 2) * 64, rbw = (eh ? 3 : 1) * (CW / 2) * 64, rc = (eh ? 1 : 2) * (CW / 2) * 64;

    for (int it = 0; it < nsteps; ++it) {
        const int buf = it & 1;
        ww2_load(it + 1 < nsteps);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        const float* Gb = &Gs[buf][gbase];
        const float* Xa0 = &Xs[buf][xbase0 + ra], *Xa1 = &Xs[buf][xbase1 + ra];
        const float* Xq0 = &Xs[buf][xbase0 + rbw], *Xq1 = &Xs[buf][xbase1 + rbw];
        const float* Xc0 = &Xs[buf][xbase0 + rc], *Xc1 = &Xs[buf][xbase1 + rc];
        // raw LDS values of k pair J: 2x2 output-gradient tile (dq) and the three patch rows this eta half needs
#define SED_WW2_RAW(J, DQ, XA, XB, XC)                                                                          \
    {                                                                                                           \
        constexpr int rp_k = (2 * (J)) / TCK, tcb = (2 * (J)) % TCK;                                            \
        const float* gp = Gb + ((2 * rp_k) * GW + 2 * tcb) * 64;                                                \
        DQ[0] = gp[0]; DQ[1] = gp[64]; DQ[2] = gp[GW * 64]; DQ[3] = gp[GW * 64 + 64];                           \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                      \
            const bool s1_ = ((jj & 1) ^ (jj >> 1)) != 0;                                                       \
            const int eo = (rp_k * CW + tcb + (jj >> 1)) * 64;   /* line of patch row 2*rp_k, column 2*tcb+jj */ \
            XA[jj] = (s1_ ? Xa1 : Xa0)[eo];                                                                     \
            XB[jj] = (s1_ ? Xq1 : Xq0)[eo];                                                                     \
            XC[jj] = (s1_ ? Xc1 : Xc0)[eo];                                                                     \
        }                                                                                                       \
    }
        float dq[4], xa[4], xb[4], xc[4], dqn[4], xan[4], xbn[4], xcn[4];
        SED_WW2_RAW(0, dq, xa, xb, xc)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // reads of the NEXT k pair are in flight while this pair's MFMAs run
            if (j == 0) SED_WW2_RAW(1, dqn, xan, xbn, xcn)
            if (j == 1) SED_WW2_RAW(2, dqn, xan, xbn, xcn)
            if (j == 2) SED_WW2_RAW(3, dqn, xan, xbn, xcn)
            if (j == 3) SED_WW2_RAW(4, dqn, xan, xbn, xcn)
            if (j == 4) SED_WW2_RAW(5, dqn, xan, xbn, xcn)
            if (j == 5) SED_WW2_RAW(6, dqn, xan, xbn, xcn)
            if (j == 6) SED_WW2_RAW(7, dqn, xan, xbn, xcn)
            __builtin_amdgcn_sched_barrier(0);
            // output-gradient tile -> this wave's two eta rows of dM, then the four xi columns
            const float ea0 = fmaf(al, dq[2], dq[0]), ea1 = fmaf(al, dq[3], dq[1]);
            const float eb0 = fmaf(ga, dq[2], be * dq[0]), eb1 = fmaf(ga, dq[3], be * dq[1]);
            // input patch -> this wave's two eta rows of V
            float ca[4], cb[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                ca[jj] = xa[jj] - xc[jj];
                cb[jj] = fmaf(sgn, xb[jj], xc[jj]);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ea0, ca[0] - ca[2], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ea0 + ea1, ca[1] + ca[2], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ea0 - ea1, ca[2] - ca[1], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(-ea1, ca[1] - ca[3], acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(eb0, cb[0] - cb[2], acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(eb0 + eb1, cb[1] + cb[2], acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(eb0 - eb1, cb[2] - cb[1], acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(-eb1, cb[1] - cb[3], acc[7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j == 5) {
                ww2_store(buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { dq[jj] = dqn[jj]; xa[jj] = xan[jj]; xb[jj] = xbn[jj]; xc[jj] = xcn[jj]; }
        }
#undef SED_WW2_RAW
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
#undef SED_WW2_LOAD
#undef SED_WW2_PIN
#undef SED_WW2_STORE
#undef ww2_load
#undef ww2_store
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        float* out = p.partial + ((long)slice * 16 + eh * 8 + a) * p.N * p.K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            out[(long)co * p.K + ci0 + cl] = acc[a][r];
        }
    }
}

// Slice reduction, two stages so that layers with few (co, ci) pairs but hundreds of slices still fill the chip:
// stage A sums a chunk of slices in fp64 (block = 64 consecutive (co,ci) elements x 4 groups of 4 coordinates);
// stage B sums the chunks, applies dg = G^T dU G and scatters to OIHW.
__global__ __launch_bounds__(256) void wgrad_wino2_reduce_a_kernel(const float* __restrict__ partial, int nslices,
                                                                   int slices_per_chunk, long nk, double* __restrict__ ws) {
    const long e = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int ag = threadIdx.x >> 6;
    if (e >= nk) return;
    const int s0 = blockIdx.y * slices_per_chunk;
    const int s1 = min(nslices, s0 + slices_per_chunk);
    double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
    for (int sl = s0; sl < s1; ++sl) {
        const float* q = partial + ((long)sl * 16 + ag * 4) * nk + e;
        u0 += (double)q[0]; u1 += (double)q[nk]; u2 += (double)q[2 * nk]; u3 += (double)q[3 * nk];
    }
    double* o = ws + ((long)blockIdx.y * 16 + ag * 4) * nk + e;
    o[0] = u0; o[nk] = u1; o[2 * nk] = u2; o[3 * nk] = u3;
}

__global__ __launch_bounds__(256) void wgrad_wino2_reduce_b_kernel(const double* __restrict__ ws, int nchunks, long nk,
                                                                   float* __restrict__ out) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nk; e += (long)gridDim.x * 256) {
        double u[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) u[a] = 0.0;
        for (int c = 0; c < nchunks; ++c) {
            const double* q = ws + (long)c * 16 * nk + e;
#pragma unroll
            for (int a = 0; a < 16; ++a) u[a] += q[a * nk];
        }
        double t[3][4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const double u0 = u[xi], u1 = u[4 + xi], u2 = u[8 + xi], u3 = u[12 + xi];
            t[0][xi] = u0 + 0.5 * (u1 + u2);
            t[1][xi] = 0.5 * (u1 - u2);
            t[2][xi] = 0.5 * (u1 + u2) + u3;
        }
        float* o = out + e * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            o[ky * 3 + 0] = (float)(t[ky][0] + 0.5 * (t[ky][1] + t[ky][2]));
            o[ky * 3 + 1] = (float)(0.5 * (t[ky][1] - t[ky][2]));
            o[ky * 3 + 2] = (float)(0.5 * (t[ky][1] + t[ky][2]) + t[ky][3]);
        }
    }
}

__host__ int wwino2_chunks(long ns, int* slices_per_chunk) {
    long spc = (ns + 15) / 16;            // <= 16 chunks ...
    if (spc < 32) spc = 32;               // ... of >= 32 slices (keeps the fp64 workspace small when slices are few)
    if (spc > ns) spc = ns;
    *slices_per_chunk = (int)spc;
    return (int)((ns + spc - 1) / spc);
}

__host__ bool wwino2_geometry(int B, int H, int W, int Cin, int Cout, int* nrg, int* nsg, long* U) {
    if (B <= 0 || H < 1 || !(W == 8 || W == 16 || W == 32 || W == 64) || Cin <= 0 || Cin % 32 || Cout <= 0 || Cout % 64) return false;
    const int TCK = W == 8 ? 4 : 8, RPK = 16 / TCK, TH = (H + 1) / 2;
    *nrg = (TH + RPK - 1) / RPK;
    *nsg = (W / 2) / TCK;
    *U = (long)B * (*nrg) * (*nsg);
    return true;
}

}  // namespace

// Unit slices of the 2-D Winograd wgrad (a unit = 16 tiles = 64 output pixels; <= 512 units per slice bounds the fp32
// accumulation chains; the slice count fills whole rounds of 512 resident workgroups).  Returns the scratch size in floats.
SED_API long sed_wgrad_wino2_partial_floats(int B, int H, int W, int Cin, int Cout, int* nslices_out, int* units_per_slice_out) {
    int nrg, nsg;
    long U;
    if (!wwino2_geometry(B, H, W, Cin, Cout, &nrg, &nsg, &U)) return 0;
    const long ids = (long)(Cout / 64) * (Cin / 32);
    const long capacity = 512;
    long ns_min = (U + 511) / 512;
    long fill = (2 * capacity + ids - 1) / ids;
    if (fill > ns_min) ns_min = fill;
    long rounds = (ids * ns_min + capacity - 1) / capacity;
    long ns = rounds * capacity / ids;
    if (ns < ns_min) ns = ns_min;
    long ups = (U + ns - 1) / ns;
    if (ups < 4) ups = 4;
    ns = (U + ups - 1) / ups;
    if (nslices_out) *nslices_out = (int)ns;
    if (units_per_slice_out) *units_per_slice_out = (int)ups;
    int spc;
    const int nchunks = wwino2_chunks(ns, &spc);
    return ns * 16L * Cin * Cout + 2L * nchunks * 16L * Cin * Cout;      // fp32 slices + fp64 chunk sums
}

// dW (OIHW) via the 2-D Winograd domain; same contract as sed_conv3x3_wgrad.  partial: sed_wgrad_wino2_partial_floats
// floats.  Needs W in {8,16,32,64}, Cin % 32 == 0, Cout % 64 == 0.
SED_API int sed_conv3x3_wgrad_wino2(const float* x, const float* gy, float* dw_oihw, float* partial, int B, int H, int W,
                                    int Cin, int Cout, const float* in_scale, const float* in_shift, hipStream_t stream) {
    int nrg, nsg, ns, ups;
    long U;
    if (!wwino2_geometry(B, H, W, Cin, Cout, &nrg, &nsg, &U) || (long)B * H * W >= (1L << 31)) return SED_EINVAL;
    sed_wgrad_wino2_partial_floats(B, H, W, Cin, Cout, &ns, &ups);
    WWino2P p{x, gy, partial, in_scale, in_shift, B, H, W, Cin, Cout, nrg, nsg, U, ups, (Cout / 64) * (Cin / 32)};
    dim3 grid((unsigned)((long)p.ids_per_slice * ns)), block(256);
    const bool in_t = in_scale != nullptr, w8 = W == 8;
    if (in_t) { if (w8) hipLaunchKernelGGL((wgrad_wino2_kernel<true, true>), grid, block, 0, stream, p);
                else hipLaunchKernelGGL((wgrad_wino2_kernel<true, false>), grid, block, 0, stream, p); }
    else { if (w8) hipLaunchKernelGGL((wgrad_wino2_kernel<false, true>), grid, block, 0, stream, p);
           else hipLaunchKernelGGL((wgrad_wino2_kernel<false, false>), grid, block, 0, stream, p); }
    const long nk = (long)Cin * Cout;
    int spc;
    const int nchunks = wwino2_chunks(ns, &spc);
    double* ws = reinterpret_cast<double*>(partial + (long)ns * 16 * nk);      // 8-byte aligned: 16*nk floats per slice
    hipLaunchKernelGGL(wgrad_wino2_reduce_a_kernel, dim3((unsigned)sed_cdiv(nk, 64), nchunks), dim3(256), 0, stream, partial, ns,
                       spc, nk, ws);
    hipLaunchKernelGGL(wgrad_wino2_reduce_b_kernel, dim3(sed_cdiv(nk, 256) > 4096 ? 4096 : sed_cdiv(nk, 256)), dim3(256), 0,
                       stream, ws, nchunks, nk, dw_oihw);
    SED_LAUNCH_CHECK();
    return 0;
}

// 3x3 convolution with a FUSED 2-D Winograd F(2x2,3x3), fp32 MFMA: 16 multiplies per 2x2 output tile and (ci, co)
// pair instead of 36.
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
// Fusions as in the direct kernel (conv.hip): input relu(scale*x+shift); epilogue 1 = BN statistics (sum, M2) per wave (64 pixels)
// + the per-part pixel count (tiles past the image edge are not counted); epilogue 2 = ReLU mask + BN-backward sums.
#include "common.h"
#include "sed_hip.h"
SED_OBJECT_FLAGS(conv_wino2)

// Round-2 measurements with timing switches that have since been removed from this file (no output stores / no yprev loads /
// no eta exchange / no statistics / one K-step only; tools/experiments/experiment_kernel_ablations.patch) (64->64 @ 1001x64, B = 128): prologue + ONE K-step + epilogue
// = 0.94 ms of the 2.69 ms the 8-step kernel takes, i.e. the per-workgroup fixed cost equals 2.8 K-steps and is NOT hidden
// by the co-resident workgroup; of it the output stores are 0.17 ms, the eta exchange 0.03 ms, the statistics 0, the
// previous-activation loads of the dgrad epilogue 0.33 ms.  Delaying half of the first generation of workgroups by 7-14 us
// (so that the two workgroups of a CU stay out of phase) changes nothing: the fixed cost is memory-system time (first-touch
// loads of 32-byte pieces out of 128-byte lines + 4-byte-per-lane stores), not idle issue slots.  Touching the dgrad
// epilogue's previous-activation lines at kernel start (one 4-byte load per thread, so that they wait in L2) makes the
// kernel 1-4 % SLOWER: the extra cold requests queue in front of the first A tile.
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

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4fma(float s, float4 a, float4 b) {
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}

// Input transform of one eta row for the forward kernel, on 4-channel fragments held as float pairs: column values
// c_j = a_j - r_j (SUB) or s*a_j + r_j (FMA), then the xi columns v0 = c0-c2, v1 = c1+c2, v2 = c2-c1, v3 = c1-c3.
// 16 packed-fp32 instructions in ONE asm block: hipcc scalarises the equivalent vector code (2x the VALU instructions,
// plus register-pair moves), and every VALU instruction issued costs MFMA time here.  The trailing s_nop covers the
// VALU-write -> MFMA-read wait states the compiler cannot see through the asm.
#define SED_XF_TAIL                                                                                             \
        "v_pk_add_f32 %8, %4, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"   /* t = c2 - c1 */                            \
        "v_pk_add_f32 %9, %5, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"                                                \
        "v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"   /* v0 = c0 - c2 */                           \
        "v_pk_add_f32 %1, %1, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"                                                \
        "v_pk_add_f32 %6, %2, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"   /* v3 = c1 - c3 */                           \
        "v_pk_add_f32 %7, %3, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"                                                \
        "v_pk_add_f32 %2, %2, %4\n\t"                             /* v1 = c1 + c2 */                           \
        "v_pk_add_f32 %3, %3, %5\n\t"                                                                          \
        "s_nop 1"
#define SED_XF_OUTS "=&v"(v0l), "=&v"(v0h), "=&v"(v1l), "=&v"(v1h), "=&v"(c2l), "=&v"(c2h), "=&v"(v3l), "=&v"(v3h), "=&v"(v2l), "=&v"(v2h)
#define SED_XF_INS "v"(f2{a[0].x, a[0].y}), "v"(f2{a[0].z, a[0].w}), "v"(f2{a[1].x, a[1].y}), "v"(f2{a[1].z, a[1].w}),     \
                   "v"(f2{a[2].x, a[2].y}), "v"(f2{a[2].z, a[2].w}), "v"(f2{a[3].x, a[3].y}), "v"(f2{a[3].z, a[3].w}),     \
                   "v"(f2{r[0].x, r[0].y}), "v"(f2{r[0].z, r[0].w}), "v"(f2{r[1].x, r[1].y}), "v"(f2{r[1].z, r[1].w}),     \
                   "v"(f2{r[2].x, r[2].y}), "v"(f2{r[2].z, r[2].w}), "v"(f2{r[3].x, r[3].y}), "v"(f2{r[3].z, r[3].w})
#define SED_XF_PACK                                                                                             \
    v[0] = f4v{v0l.x, v0l.y, v0h.x, v0h.y}; v[1] = f4v{v1l.x, v1l.y, v1h.x, v1h.y};                             \
    v[2] = f4v{v2l.x, v2l.y, v2h.x, v2h.y}; v[3] = f4v{v3l.x, v3l.y, v3h.x, v3h.y};
__device__ __forceinline__ void wino2_xf_sub(const f4v (&a)[4], const f4v (&r)[4], f4v (&v)[4]) {
    f2 v0l, v0h, v1l, v1h, c2l, c2h, v3l, v3h, v2l, v2h;
    asm("v_pk_add_f32 %0, %10, %18 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %11, %19 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %2, %12, %20 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %3, %13, %21 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %4, %14, %22 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %5, %15, %23 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %6, %16, %24 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %7, %17, %25 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        SED_XF_TAIL
        : SED_XF_OUTS : SED_XF_INS);
    SED_XF_PACK
}
__device__ __forceinline__ void wino2_xf_fma(f2 sg, const f4v (&a)[4], const f4v (&r)[4], f4v (&v)[4]) {
    f2 v0l, v0h, v1l, v1h, c2l, c2h, v3l, v3h, v2l, v2h;
    asm("v_pk_fma_f32 %0, %26, %10, %18\n\t"
        "v_pk_fma_f32 %1, %26, %11, %19\n\t"
        "v_pk_fma_f32 %2, %26, %12, %20\n\t"
        "v_pk_fma_f32 %3, %26, %13, %21\n\t"
        "v_pk_fma_f32 %4, %26, %14, %22\n\t"
        "v_pk_fma_f32 %5, %26, %15, %23\n\t"
        "v_pk_fma_f32 %6, %26, %16, %24\n\t"
        "v_pk_fma_f32 %7, %26, %17, %25\n\t"
        SED_XF_TAIL
        : SED_XF_OUTS : SED_XF_INS, "s"(sg));
    SED_XF_PACK
}
#undef SED_XF_TAIL
#undef SED_XF_OUTS
#undef SED_XF_INS
#undef SED_XF_PACK

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

    // ---- A staging: item e = tid + 256*i -> pixel e >> 1 of the block's rows, 16-byte chunk e & 1.
    // Raw buffer loads over this workgroup's image: thread-constant byte offset + a scalar k offset, no per-step
    // address arithmetic on the vector unit (every VALU instruction issued costs MFMA time on this chip).
    const int c2 = tid & 1;
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (long)b * p.H * W * p.K, 0, (int)((unsigned)p.H * W * p.K * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(INT ? p.in_scale : p.x), 0, p.K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(INT ? p.in_shift : p.x), 0, p.K * 4, 0x00020000);
#define SED_W2META(i)                                                                                           \
    bool sok##i;                                                                                                \
    int lso##i, aoff##i;                                                                                        \
    float4 areg##i = make_float4(0.f, 0.f, 0.f, 0.f);                                                           \
    {                                                                                                           \
        const int pix = (tid + 256 * i) >> 1;                                                                   \
        const int r = pix >> logW, w = pix & (W - 1);                                                           \
        const int h = hbase + r;                                                                                \
        sok##i = r < nrows && (unsigned)h < (unsigned)p.H;                                                      \
        lso##i = sw_off((r * 2 + ((w + 1) & 1)) * S + ((w + 1) >> 1), c2);                                      \
        aoff##i = sok##i ? ((h * W + w) * p.K + c2 * 4) * 4 : OOB;                                              \
    }
    SED_W2META(0) SED_W2META(1) SED_W2META(2)
#undef SED_W2META
    // ---- B DMA: wave wv stages coordinates 4*wv .. 4*wv+3, one instruction (64 lanes x 16 B = 32 rows) each;
    // scalar base (k-step, coordinate) + thread-constant 32-bit offset
    const int brow_in = lane >> 1;
    const int boff = ((n0 + brow_in) * 8 + (((lane & 1) ^ ((brow_in >> 3) & 1)) << 2)) * 4;
    const float* const bw = p.wu + (long)(wvu * 4) * p.N * 8;
    const long b_xi_stride = (long)p.N * 8;          // next Winograd coordinate
    const long b_k_stride = 16L * p.N * 8;           // next k-step
    const unsigned bs_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)Bs);

    floatx16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int KT = p.K >> 3;
    f2 sc01 = {0.f, 0.f}, sc23 = sc01, sh01 = sc01, sh23 = sc01;

#define SED_W2A_LOAD(i) areg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff##i, k_off, 0));
#define SED_W2B_LOAD(DST, j)                                                                                    \
    {                                                                                                           \
        unsigned keep_;                                                                                         \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_)                                                                             \
                     : "v"(boff), "s"(bs_base + (unsigned)(((DST) * W2_BSTAGE + (wvu * 4 + (j)) * 256) * 4)),   \
                       "s"(bsrc + (j) * b_xi_stride)                                                            \
                     : "memory");                                                                               \
    }
#define w2gload(IT, DST)                                                                                        \
    {                                                                                                           \
        const int k_off = (IT) * 32;                                   /* bytes */                              \
        const float* bsrc = bw + (long)(IT) * b_k_stride;                                                       \
        if (INT) {                                                                                              \
            const float4 s4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srs, c2 * 16, k_off, 0)); \
            const float4 h4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hrs, c2 * 16, k_off, 0)); \
            sc01 = f2{s4.x, s4.y}; sc23 = f2{s4.z, s4.w}; sh01 = f2{h4.x, h4.y}; sh23 = f2{h4.z, h4.w};         \
        }                                                                                                       \
        SED_W2B_LOAD(DST, 0) SED_W2B_LOAD(DST, 1) SED_W2B_LOAD(DST, 2) SED_W2B_LOAD(DST, 3)                     \
        SED_W2A_LOAD(0) SED_W2A_LOAD(1) SED_W2A_LOAD(2)                                                         \
    }
#define SED_W2A_STORE(BUF, i)                                                                                   \
    {                                                                                                           \
        asm volatile("" : "+v"(areg##i.x), "+v"(areg##i.y), "+v"(areg##i.z), "+v"(areg##i.w));                  \
        float4 v = areg##i;                                                                                     \
        if (INT) {                                                                                              \
            f2 v01 = f2{v.x, v.y} * sc01 + sh01, v23 = f2{v.z, v.w} * sc23 + sh23;                              \
            v01 = __builtin_elementwise_max(v01, f2{0.f, 0.f}); v23 = __builtin_elementwise_max(v23, f2{0.f, 0.f}); \
            v = make_float4(v01.x, v01.y, v23.x, v23.y);                                                        \
        }                                                                                                       \
        if (sok##i) *reinterpret_cast<float4*>(&As[(BUF) * W2_ASTAGE + lso##i]) = v;                            \
    }
#define w2lstore(BUF) { SED_W2A_STORE(BUF, 0) SED_W2A_STORE(BUF, 1) SED_W2A_STORE(BUF, 2) }

    // The K loop never stores the halo entries (plane 0 position 0 = column -1, plane 1 position TW = column W of every
    // block row) nor the pixels of rows outside the image: zero exactly those, once, in both stages (~2 LDS stores per
    // thread instead of a 25-store fill of both stages + a barrier; the barrier behind the first tile store publishes them)
    for (int i = tid; i < nrows * 8; i += 256) {       // (34 block rows at W = 8: 272 entries)
        const int r = i >> 3, e = (i >> 2) & 1;
        const int row = (r * 2 + e) * S + (e ? TW : 0);
        *reinterpret_cast<float4*>(&As[(i & 1) * W2_ASTAGE + sw_off(row, (i >> 1) & 1)]) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#define SED_W2ZERO(i)                                                                                           \
    if (!sok##i && (((tid + 256 * i) >> 1) >> logW) < nrows) {                                                  \
        *reinterpret_cast<float4*>(&As[lso##i]) = make_float4(0.f, 0.f, 0.f, 0.f);                              \
        *reinterpret_cast<float4*>(&As[W2_ASTAGE + lso##i]) = make_float4(0.f, 0.f, 0.f, 0.f);                  \
    }
    SED_W2ZERO(0) SED_W2ZERO(1) SED_W2ZERO(2)
#undef SED_W2ZERO
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
    const f2 sgn2 = eh ? f2{-1.f, -1.f} : f2{1.f, 1.f};
    const float* pa[4], *pb[4], *pc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pa[j] = As + sw_off((ra * 2 + (j & 1)) * S + tw_l + (j >> 1), chunk);
        pb[j] = As + sw_off((rbw * 2 + (j & 1)) * S + tw_l + (j >> 1), chunk);
        pc[j] = As + sw_off((rc * 2 + (j & 1)) * S + tw_l + (j >> 1), chunk);
    }
    const float* const pbf = Bs + eh * 8 * 256 + sw_off(lane & 31, chunk);          // + (e*4 + xi) * 256

    // one K-step on stage BUF (compile-time, so every LDS access is base register + immediate) while step ITN is
    // fetched into the other stage
#define SED_W2MMA(A_, VV)                                                                                       \
    acc[A_] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.x, bf[A_].x, acc[A_], 0, 0, 0);                            \
    acc[A_] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.y, bf[A_].y, acc[A_], 0, 0, 0);                            \
    acc[A_] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.z, bf[A_].z, acc[A_], 0, 0, 0);                            \
    acc[A_] = __builtin_amdgcn_mfma_f32_32x32x2f32(VV.w, bf[A_].w, acc[A_], 0, 0, 0);
#define W2_STEP(BUF, ITN)                                                                                       \
    {                                                                                                           \
        w2gload(ITN, (BUF) ^ 1);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        __builtin_amdgcn_s_setprio(1);                                                                          \
        f4v xa[4], xb[4], xc[4], bf[8];                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            xa[j] = *reinterpret_cast<const f4v*>(pa[j] + (BUF) * W2_ASTAGE);                                   \
            xc[j] = *reinterpret_cast<const f4v*>(pc[j] + (BUF) * W2_ASTAGE);                                   \
        }                                                                                                       \
        _Pragma("unroll") for (int a = 0; a < 4; ++a) bf[a] = *reinterpret_cast<const f4v*>(pbf + (BUF) * W2_BSTAGE + a * 256); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) xb[j] = *reinterpret_cast<const f4v*>(pb[j] + (BUF) * W2_ASTAGE); \
        _Pragma("unroll") for (int a = 4; a < 8; ++a) bf[a] = *reinterpret_cast<const f4v*>(pbf + (BUF) * W2_BSTAGE + a * 256); \
        {                                                                                                       \
            f4v v[4];                                                                                           \
            wino2_xf_sub(xa, xc, v);                                                                            \
            SED_W2MMA(0, v[0]) SED_W2MMA(1, v[1]) SED_W2MMA(2, v[2]) SED_W2MMA(3, v[3])                         \
        }                                                                                                       \
        /* the next step's A tile goes to LDS mid-step, behind 16 queued MFMAs: in front of the barrier its   */ \
        /* BN-affine + ReLU instructions sat on every wave's critical path (3-5 % on the fused-input layers)  */ \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        w2lstore((BUF) ^ 1);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        {                                                                                                       \
            f4v v[4];                                                                                           \
            wino2_xf_fma(sgn2, xb, xc, v);                                                                      \
            SED_W2MMA(4, v[0]) SED_W2MMA(5, v[1]) SED_W2MMA(6, v[2]) SED_W2MMA(7, v[3])                         \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        __builtin_amdgcn_s_setprio(0);                                                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        __syncthreads();                                                                                        \
    }
    {
        // step 0 is peeled: straight after acc = 0 the compiler folds the zero accumulators into the MFMAs' C operand
        // (no 128 v_mov per wave, which matters for the 8-step 64-channel layers)
        W2_STEP(0, KT > 1 ? 1 : 0)
        int it = 1;
        for (; it + 1 < KT; it += 2) {
            W2_STEP(1, it + 1)
            W2_STEP(0, it + 2 < KT ? it + 2 : it + 1)
        }
        if (it < KT) W2_STEP(1, it)
    }
#undef W2_STEP
#undef SED_W2MMA
#undef w2gload
#undef w2lstore
#undef SED_W2A_LOAD
#undef SED_W2B_LOAD
#undef SED_W2A_STORE

    // ---- epilogue.  z[e][q] = column inverse transform of this wave's two eta rows; the wave keeps
    // mine[q] = +-(z[0][q] + z[1][q]) and hands z[1] (half 0) / z[0] (half 1) to its partner wave through LDS
    // (the staging buffers are free after the loop's last barrier).  Kept lean on the vector unit (float pairs over
    // adjacent accumulator rows, buffer-descriptor addressing with out-of-range rows dropped by the hardware): for the
    // 64-channel layers the K loop is only 8 steps long and the epilogue is a third of the workgroup's instructions.
    float* xch = smem;                                 // [4 waves][32][64 lanes]
    const int half = lane >> 5;
    const int col = n0 + (lane & 31);
    // byte offset (within the image) of output pixel (2*th+eh, 2*tw) of block tile tl, channel col:
    //   pixel = (2*th0+eh)*W + 2*(tl + (tl & ~(TW-1)));  rows past the image land beyond num_records
    const unsigned y_img_bytes = (unsigned)p.H * W * p.N * 4u;
    const __amdgpu_buffer_rsrc_t yrs =
        __builtin_amdgcn_make_buffer_rsrc(p.y + (long)b * p.H * W * p.N, 0, (int)y_img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(EPI == 2 ? p.yprev + (long)b * p.H * W * p.N : p.x), 0, (int)y_img_bytes, 0x00020000);
    const int n4 = p.N * 4;
    const unsigned n8 = (unsigned)p.N * 8u;
    const int t0 = tb * 32 + 4 * half;
    const unsigned lane_off = (unsigned)(((2 * th0 + eh) * W * p.N + col) * 4);
    unsigned yoff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned tl = (unsigned)(t0 + (r & 3) + 8 * (r >> 2));
        yoff[r] = __umul24(tl + (tl & ~(unsigned)(TW - 1)), n8) + lane_off;
    }
    float s1 = 0.f, s2 = 0.f;
    f2 s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
    float e_sc = 0.f, e_sh = 0.f, e_mu = 0.f, e_is = 0.f;
    // everything below works on pairs over the accumulator rows (2rp, 2rp+1), separately for q = 0 and q = 1
    f2 yp0[EPI == 2 ? 8 : 1], yp1[EPI == 2 ? 8 : 1];   // previous-layer activations
#define SED_YPREV_LOADS \
    if (EPI == 2) { \
        e_sc = p.p_scale[col]; e_sh = p.p_shift[col]; e_mu = p.p_mean[col]; e_is = p.p_invstd[col]; \
_Pragma("unroll") \
        for (int rp = 0; rp < 8; ++rp) { \
            yp0[EPI == 2 ? rp : 0] = f2{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, (int)yoff[2 * rp], 0, 0)), \
                                        __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, (int)yoff[2 * rp + 1], 0, 0))}; \
            yp1[EPI == 2 ? rp : 0] = f2{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, (int)yoff[2 * rp], n4, 0)), \
                                        __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, (int)yoff[2 * rp + 1], n4, 0))}; \
        } \
    }
    SED_YPREV_LOADS         // requested first, needed behind the output transform + the eta exchange (placing them
                            // after the exchange writes instead measures the same: the compiler hoists them anyway)
#undef SED_YPREV_LOADS
    f2 mine[16];                                       // [row pair rp][q]: accumulator rows 2rp, 2rp+1
#define SED_AP(a) f2{acc[a][2 * rp], acc[a][2 * rp + 1]}
#define SED_GIVE(G0, G1)                                                                                        \
    {                                                                                                           \
    xch[(wvu * 32 + 4 * rp + 0) * 64 + lane] = G0.x; xch[(wvu * 32 + 4 * rp + 1) * 64 + lane] = G1.x;           \
    xch[(wvu * 32 + 4 * rp + 2) * 64 + lane] = G0.y; xch[(wvu * 32 + 4 * rp + 3) * 64 + lane] = G1.y; }
    if (eh) {
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const f2 z00 = SED_AP(0) + SED_AP(1) + SED_AP(2), z01 = SED_AP(1) - SED_AP(2) - SED_AP(3);
            const f2 z10 = SED_AP(4) + SED_AP(5) + SED_AP(6), z11 = SED_AP(5) - SED_AP(6) - SED_AP(7);
            SED_GIVE(z00, z01)
            mine[2 * rp] = -z00 - z10; mine[2 * rp + 1] = -z01 - z11;
        }
    } else {
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const f2 z00 = SED_AP(0) + SED_AP(1) + SED_AP(2), z01 = SED_AP(1) - SED_AP(2) - SED_AP(3);
            const f2 z10 = SED_AP(4) + SED_AP(5) + SED_AP(6), z11 = SED_AP(5) - SED_AP(6) - SED_AP(7);
            SED_GIVE(z10, z11)
            mine[2 * rp] = z00 + z10; mine[2 * rp + 1] = z01 + z11;
        }
    }
#undef SED_AP
#undef SED_GIVE
    __syncthreads();
    const float* rx = xch + ((wvu ^ 1) * 32) * 64 + lane;
    f2 yq0[8], yq1[8];                                 // outputs
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
        f2 y0 = mine[2 * rp], y1 = mine[2 * rp + 1];
        {
            y0 += f2{rx[(4 * rp + 0) * 64], rx[(4 * rp + 2) * 64]};      // q = 0 of rows 2rp, 2rp+1
            y1 += f2{rx[(4 * rp + 1) * 64], rx[(4 * rp + 3) * 64]};      // q = 1
        }
        if (EPI == 1 || EPI == 2) {
            const bool oka = yoff[2 * rp] < y_img_bytes, okb = yoff[2 * rp + 1] < y_img_bytes;
            if (EPI == 2) {
                const f2 a0 = yp0[EPI == 2 ? rp : 0], a1 = yp1[EPI == 2 ? rp : 0];
                const f2 t0_ = a0 * f2{e_sc, e_sc} + f2{e_sh, e_sh}, t1_ = a1 * f2{e_sc, e_sc} + f2{e_sh, e_sh};
                y0.x = (oka && t0_.x > 0.f) ? y0.x : 0.f; y0.y = (okb && t0_.y > 0.f) ? y0.y : 0.f;
                y1.x = (oka && t1_.x > 0.f) ? y1.x : 0.f; y1.y = (okb && t1_.y > 0.f) ? y1.y : 0.f;
                s2p = y0 * ((a0 - f2{e_mu, e_mu}) * f2{e_is, e_is}) + s2p;
                s2p = y1 * ((a1 - f2{e_mu, e_mu}) * f2{e_is, e_is}) + s2p;
            } else {
                y0.x = oka ? y0.x : 0.f; y0.y = okb ? y0.y : 0.f;
                y1.x = oka ? y1.x : 0.f; y1.y = okb ? y1.y : 0.f;
            }
            s1p += y0 + y1;
        }
        yq0[rp] = y0; yq1[rp] = y1;
        {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y0.x), yrs, (int)yoff[2 * rp], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y1.x), yrs, (int)yoff[2 * rp], n4, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y0.y), yrs, (int)yoff[2 * rp + 1], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y1.y), yrs, (int)yoff[2 * rp + 1], n4, 0);
        }
    }
    float cnt = 0.f;
    if (EPI == 1) {
        // valid pixels of this wave's 32 tiles (scalar): row pairs th < ceil((H - eh) / 2)
        const int rp_cnt = TW >= 32 ? 1 : (32 >> logTW), per_rp = TW >= 32 ? 32 : TW;
        int nv = (p.H - eh + 1) / 2 - (th0 + ((tb * 32) >> logTW));
        nv = nv < 0 ? 0 : (nv > rp_cnt ? rp_cnt : nv);
        cnt = (float)(2 * nv * per_rp);
        s1 = s1p.x + s1p.y;
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = cnt > 0.f ? s1 / cnt : 0.f;
        const f2 mean2 = {mean, mean};
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const bool oka = yoff[2 * rp] < y_img_bytes, okb = yoff[2 * rp + 1] < y_img_bytes;
            f2 d0 = yq0[rp] - mean2, d1 = yq1[rp] - mean2;
            d0.x = oka ? d0.x : 0.f; d0.y = okb ? d0.y : 0.f;
            d1.x = oka ? d1.x : 0.f; d1.y = okb ? d1.y : 0.f;
            s2p = d0 * d0 + s2p;
            s2p = d1 * d1 + s2p;
        }
        s2 = s2p.x + s2p.y;
        s2 += __shfl_xor(s2, 32, 64);
    }
    if (EPI == 2) {
        s1 = s1p.x + s1p.y; s2 = s2p.x + s2p.y;
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    }
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
        (long)B * H * W >= (1L << 31) || (long)B * nrb * (Cout / 32) >= (1L << 31) ||
        (long)H * W * (Cin > Cout ? Cin : Cout) * 4 >= (1L << 31))   // one image must fit a 31-bit buffer-descriptor range
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

// The 12 packed-fp32 instructions that turn the raw LDS values of one k pair into the 8 + 8 MFMA operands of an eta half.
// Written as ONE asm block because hipcc scalarises most v2f32 expressions with swizzles (28 VALU instead of 12), and on
// this chip every VALU instruction issued costs MFMA time (measured: kernel time ~ 64*MFMAs + 4*VALUs cycles per SIMD).
// The trailing s_nop covers the VALU-write -> MFMA-read wait states the compiler cannot see through the asm.
__device__ __forceinline__ void wgrad_wino2_transforms(f2 d0, f2 d1, f2 a03, f2 a12, f2 b03, f2 b12, f2 c03, f2 c12, f2 al,
                                                       f2 be, f2 ga, f2& ea, f2& eb, f2& ma, f2& mb, f2& va03, f2& va12,
                                                       f2& vb03, f2& vb12) {
    asm("v_pk_fma_f32 %0, %16, %9, %8\n\t"                                                  // ea = al*d1 + d0
        "v_pk_fma_f32 %1, %17, %8, %9\n\t"                                                  // eb = be*d0 + d1 (sign of half 1 moved into cb)
        "v_pk_add_f32 %4, %10, %14 neg_lo:[0,1] neg_hi:[0,1]\n\t"                           // ca03 = a03 - c03
        "v_pk_add_f32 %5, %11, %15 neg_lo:[0,1] neg_hi:[0,1]\n\t"                           // ca12 = a12 - c12
        "v_pk_fma_f32 %6, %18, %14, %12\n\t"                                                // cb03 = ga*c03 + b03
        "v_pk_fma_f32 %7, %18, %15, %13\n\t"                                                // cb12 = ga*c12 + b12
        // operand order: wherever the LOW lane reads a HIGH half, that read sits on src0 (op_sel[src0] = 1, op_sel[src1] = 0).
        // The mirrored form (op_sel = [0,1]) returns wrong values now and then beside f16 MFMAs of another kernel on the CU
        // (tools/pk_f32_beside_mfma_probe.hip); additions commute, so the results are bit-identical.
        "v_pk_add_f32 %2, %0, %0 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"             // ma = (ea.y+ea.x, -ea.y+ea.x)
        "v_pk_add_f32 %3, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"             // mb
        "v_pk_add_f32 %4, %5, %4 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]\n\t" // va03 = (-c2+c0, -c1+c3)
        "v_pk_add_f32 %5, %5, %5 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"             // va12 = (c2+c1, c2-c1)
        "v_pk_add_f32 %6, %7, %6 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]\n\t" // vb03
        "v_pk_add_f32 %7, %7, %7 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"             // vb12
        "s_nop 1"
        : "=&v"(ea), "=&v"(eb), "=&v"(ma), "=&v"(mb), "=&v"(va03), "=&v"(va12), "=&v"(vb03), "=&v"(vb12)
        : "v"(d0), "v"(d1), "v"(a03), "v"(a12), "v"(b03), "v"(b12), "v"(c03), "v"(c12), "s"(al), "s"(be), "s"(ga));
}

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
    //      gy item e -> pixel e >> 4 of the 64, chunk e & 15.
    // Loads are raw BUFFER loads over one image (descriptor rebuilt per unit on the scalar unit): rows above / below the
    // image fall outside [0, num_records) and return 0 without any per-lane masking; the only explicit test is the
    // patch's halo column at the left / right image edge (offset forced out of range).
    const int xch = tid & 7, gch = tid & 15;
    f2 xsc01 = {0.f, 0.f}, xsc23 = xsc01, xsh01 = xsc01, xsh23 = xsc01;
    if (INT) {
        const float4 sc4 = *reinterpret_cast<const float4*>(p.in_scale + ci0 + xch * 4);
        const float4 sh4 = *reinterpret_cast<const float4*>(p.in_shift + ci0 + xch * 4);
        xsc01 = f2{sc4.x, sc4.y}; xsc23 = f2{sc4.z, sc4.w}; xsh01 = f2{sh4.x, sh4.y}; xsh23 = f2{sh4.z, sh4.w};
    }
    constexpr int OOB = (int)0x80000000;
    const unsigned x_img_bytes = (unsigned)p.H * W * p.K * 4u, g_img_bytes = (unsigned)p.H * W * p.N * 4u;
#define SED_WW2_META(i)                                                                                         \
    const int xpx##i = (tid + 256 * i) >> 3;                                                                    \
    const bool xit##i = xpx##i < XPIX;                                                                          \
    const int xr##i = xpx##i / CW, xc##i = xpx##i % CW;                                                         \
    const int xls##i = (xpx##i >> 1) * 64 + 32 * ((xpx##i & 1) ^ ((xc##i >> 1) & 1)) + xch * 4;                 \
    const int gpx##i = (tid + 256 * i) >> 4;                                                                    \
    const int gr##i = gpx##i / GW, gc##i = gpx##i % GW;                                                         \
    const int gls##i = gpx##i * 64 + ((gch * 4) ^ (32 * ((gc##i >> 1) & 1)));                                   \
    const int xoffb##i = xit##i ? ((xr##i * W + xc##i) * p.K + ci0 + xch * 4) * 4 : OOB;                        \
    const int xe0##i = xc##i == 0 ? OOB : 0, xe1##i = xc##i == CW - 1 ? OOB : 0;   /* halo columns of the patch */ \
    const int goffb##i = ((gr##i * W + gc##i) * p.N + co0 + gch * 4) * 4;                                       \
    float4 xreg##i = make_float4(0.f, 0.f, 0.f, 0.f), greg##i = xreg##i;                                        \
    bool xv##i = false;
    SED_WW2_META(0) SED_WW2_META(1) SED_WW2_META(2) SED_WW2_META(3)
#undef SED_WW2_META

#define SED_WW2_LOAD(i)                                                                                         \
    {                                                                                                           \
        const int vo = (xu + xoffb##i) | (xe0##i & mfirst) | (xe1##i & mlast);                                  \
        if (INT) xv##i = (unsigned)vo < x_img_bytes;                                                            \
        xreg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0));             \
        greg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, gu + goffb##i, 0, 0));  \
    }
#define ww2_load(LIVE)                                                                                          \
    {                                                                                                           \
        const bool live = (LIVE);                                                                               \
        const int h0 = 2 * rg * RPK - 1, w0 = 2 * sg * TCK - 1;                                                 \
        const int bb = b < p.B ? b : p.B - 1;                                                                   \
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(                                   \
            const_cast<float*>(p.x) + (long)bb * p.H * W * p.K, 0, (int)x_img_bytes, 0x00020000);               \
        const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(                                   \
            const_cast<float*>(p.gy) + (long)bb * p.H * W * p.N, 0, (int)g_img_bytes, 0x00020000);              \
        const int xu = live ? (h0 * W + w0) * p.K * 4 : OOB;            /* unit origin within the image, bytes */ \
        const int gu = live ? ((h0 + 1) * W + w0 + 1) * p.N * 4 : OOB;                                          \
        const int mfirst = sg == 0 ? -1 : 0, mlast = sg == p.nsg - 1 ? -1 : 0;                                  \
        SED_WW2_LOAD(0) SED_WW2_LOAD(1) SED_WW2_LOAD(2) SED_WW2_LOAD(3)                                         \
        if (++sg == p.nsg) { sg = 0; if (++rg == p.nrg) { rg = 0; ++b; } }                                      \
    }
#define SED_WW2_PIN(r) asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
#define SED_WW2_STORE(BUF, i)                                                                                   \
    {                                                                                                           \
        SED_WW2_PIN(greg##i) SED_WW2_PIN(xreg##i)                                                               \
        float4 v = xreg##i;                                                                                     \
        if (INT) {                                                                                              \
            f2 v01 = f2{v.x, v.y} * xsc01 + xsh01, v23 = f2{v.z, v.w} * xsc23 + xsh23;                          \
            v01 = __builtin_elementwise_max(v01, f2{0.f, 0.f}); v23 = __builtin_elementwise_max(v23, f2{0.f, 0.f}); \
            v.x = xv##i ? v01.x : 0.f; v.y = xv##i ? v01.y : 0.f; v.z = xv##i ? v23.x : 0.f; v.w = xv##i ? v23.y : 0.f; \
        }                                                                                                       \
        *reinterpret_cast<float4*>(&Gs[(BUF)][gls##i]) = greg##i;                                               \
        if (xit##i) *reinterpret_cast<float4*>(&Xs[(BUF)][xls##i]) = v;                                        \
    }
#define ww2_store(BUF) { SED_WW2_STORE(BUF, 0) SED_WW2_STORE(BUF, 1) SED_WW2_STORE(BUF, 2) SED_WW2_STORE(BUF, 3) }

    ww2_load(nsteps > 0);
    ww2_store(0);
    __syncthreads();

    // ---- fragment addressing: lane = channel (lane & 31), lane half = second tile of the k pair (next tile column)
    const int half = lane >> 5, cl = lane & 31;
    const int gbase = half * 128 + ((cob * 32 + cl) ^ (32 * half));
    const int xbase0 = half * 64 + 32 * half + cl;          // patch columns 0 and 3 of a tile
    const int xbase1 = half * 64 + 32 * (1 ^ half) + cl;    // patch columns 1 and 2
    // eta half 0: dM rows (d0, d0 + d1), V rows (r0 - r2, r1 + r2); half 1: (d0 - d1, +d1), (r2 - r1, r3 - r1)
    // (the -d1 * (r1 - r3) product of the fourth eta row is carried as (+d1) * (r3 - r1): one instruction less).
    // All transforms run on float PAIRS (v_pk_*_f32: one VALU instruction per two values): the pairs are the two
    // horizontal outputs of a tile row (dq) and the patch columns (0,3) / (1,2) as the paired LDS reads deliver them.
    const f2 al2 = eh ? f2{-1.f, -1.f} : f2{0.f, 0.f}, be2 = eh ? f2{0.f, 0.f} : f2{1.f, 1.f};
    const f2 ga2 = eh ? f2{-1.f, -1.f} : f2{1.f, 1.f};
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
        // raw LDS values of k pair J: the 2x2 output-gradient tile (rows D0, D1) and the three patch rows this eta
        // half needs, each as the column pairs (0,3) and (1,2)
#define SED_WW2_RAW(J, D0, D1, A03, A12, B03, B12, C03, C12)                                                    \
    {                                                                                                           \
        constexpr int rp_k = (2 * (J)) / TCK, tcb = (2 * (J)) % TCK;                                            \
        const float* gp = Gb + ((2 * rp_k) * GW + 2 * tcb) * 64;                                                \
        D0 = f2{gp[0], gp[64]}; D1 = f2{gp[GW * 64], gp[GW * 64 + 64]};                                         \
        constexpr int eo = (rp_k * CW + tcb) * 64;          /* line of patch row 2*rp_k, column 2*tcb */         \
        A03 = f2{Xa0[eo], Xa0[eo + 64]}; A12 = f2{Xa1[eo], Xa1[eo + 64]};                                       \
        B03 = f2{Xq0[eo], Xq0[eo + 64]}; B12 = f2{Xq1[eo], Xq1[eo + 64]};                                       \
        C03 = f2{Xc0[eo], Xc0[eo + 64]}; C12 = f2{Xc1[eo], Xc1[eo + 64]};                                       \
    }
        f2 d0, d1, a03, a12, b03, b12, c03, c12, d0n, d1n, a03n, a12n, b03n, b12n, c03n, c12n;
        SED_WW2_RAW(0, d0, d1, a03, a12, b03, b12, c03, c12)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // reads of the NEXT k pair are in flight while this pair's MFMAs run
            if (j == 0) SED_WW2_RAW(1, d0n, d1n, a03n, a12n, b03n, b12n, c03n, c12n)
            if (j == 1) SED_WW2_RAW(2, d0n, d1n, a03n, a12n, b03n, b12n, c03n, c12n)
            if (j == 2) SED_WW2_RAW(3, d0n, d1n, a03n, a12n, b03n, b12n, c03n, c12n)
            if (j == 3) SED_WW2_RAW(4, d0n, d1n, a03n, a12n, b03n, b12n, c03n, c12n)
            if (j == 4) SED_WW2_RAW(5, d0n, d1n, a03n, a12n, b03n, b12n, c03n, c12n)
            if (j == 5) SED_WW2_RAW(6, d0n, d1n, a03n, a12n, b03n, b12n, c03n, c12n)
            if (j == 6) SED_WW2_RAW(7, d0n, d1n, a03n, a12n, b03n, b12n, c03n, c12n)
            __builtin_amdgcn_sched_barrier(0);
            // dM: eta rows ea = d0 + al*d1, eb = be*d0 + ga*d1 (pairs over the two tile columns q), then the xi columns
            // (e.x, e.x + e.y, e.x - e.y, -e.y); the sign of the last one is moved into the V operand.
            // V: eta rows ca = ra - rc, cb = rc +- rb; xi columns (c0 - c2, c1 + c2, c2 - c1, c3 - c1).
            f2 ea, eb, ma, mb, va03, va12, vb03, vb12;
            wgrad_wino2_transforms(d0, d1, a03, a12, b03, b12, c03, c12, al2, be2, ga2, ea, eb, ma, mb, va03, va12, vb03, vb12);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ea.x, va03.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma.x, va12.x, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma.y, va12.y, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ea.y, va03.y, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(eb.x, vb03.x, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(mb.x, vb12.x, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(mb.y, vb12.y, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(eb.y, vb03.y, acc[7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j == 6) {                                  // measured best of j = 2..7
                ww2_store(buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            d0 = d0n; d1 = d1n; a03 = a03n; a12 = a12n; b03 = b03n; b12 = b12n; c03 = c03n; c12 = c12n;
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
    if (!wwino2_geometry(B, H, W, Cin, Cout, &nrg, &nsg, &U) || (long)B * H * W >= (1L << 31) ||
        (long)H * W * (Cin > Cout ? Cin : Cout) * 4 >= (1L << 31))      // one image must fit a 31-bit buffer-descriptor range
        return SED_EINVAL;
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

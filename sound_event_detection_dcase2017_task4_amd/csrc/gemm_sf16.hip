// Dense NT GEMM with split-f16 operands on the f16 MFMA pipe:  y[M][N] = x[M][K] * w[N][K]^T (+ bias[N]).
//
// Replaces (reference pytorch/models.py): the input projections of nn.GRU(512, 256, bidirectional) (:529-530, :565-567:
// gi = x W_ih^T + b_ih for both directions, 32 000 x 1536 x 512 at B = 256) and their input gradient dx = dgi W_ih -- until
// round 4 these ran on the fp32 MFMA kernel of csrc/conv.hip (sed_gemm_nt, ~100 TFLOP/s = 0.5 ms each).
//
// Same arithmetic as csrc/conv_sf16.hip: every fp32 operand is (hi + lo) / s with hi = f16(s*v), lo = f16(s*v - hi), s a power of
// two from a device-side amax, and a product is hi*hi + hi*lo + lo*hi -- three v_mfma_f32_32x32x16_f16 with exact products and
// fp32 accumulation: the rounding error of an fp32 dot product at 3/16 of the fp32 MFMA issue time.
//
// Why it is not conversion-bound like the round-3 attempt (256 x 64 tiles, BOTH operands converted by every tile: 9x the
// conversion work per MFMA of the convolutions, 166-186 TFLOP/s):
//  * the WEIGHT operand is pre-split once per optimiser step (sed_gemm_pack_sf16: [K/16][hi, lo][N][16] f16, cached per
//    parameter like the convolution packs) and streams into LDS by LDS-DMA (global_load_lds_dwordx4, source-side swizzle): no
//    VALU, no registers;
//  * the ACTIVATION operand is converted when staged (raw buffer loads -> scale -> 3-VALU split -> 8-byte LDS stores), and a
//    workgroup covers 128 output columns, so a row block is converted N / 128 times (12 for the GRU projection) at 1.7 VALU per
//    MFMA -- the budget of the convolutions' fused-input variant.
// Workgroup = 256 rows x 128 columns, 4 waves x (64 x 128) = 8 accumulator tiles of 32 x 32 per wave; K-stage = 32 (two 16-wide
// slabs); A single-buffered behind a register prefetch, B double-buffered; 64 KB of LDS -> two workgroups per CU.
// LDS images and fragment reads as in the convolution kernel: 32-byte rows per slab and plane, 16-byte chunk XOR (row >> 3) & 1.
#include "common.h"
#include "sed_hip.h"
SED_OBJECT_FLAGS(gemm_sf16)

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct GemmSfP {
    const float* a;            // [M][K]
    const _Float16* bp;        // [K/16][2 planes][N][16]
    const float* bscale;       // [SED_AMAX_SLOTS + 1]: the weights' amax slots, then their scale (written by the pack kernel)
    const float* a_amax;       // device amax vector of a
    const float* bias;         // nullable [N]
    float* c;                  // [M][N]
    int M, N, K;
    int* err_host;             // nullable, host-mapped: set to 1 when an operand is not finite
    int* err_dev;              // nullable, device: same (read by sed_adam_amsgrad)
    float* out_amax;           // nullable: amax slots of |c| as written
};

__device__ __forceinline__ int gsw(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4); }

__device__ __forceinline__ int xcd_remap_g(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

constexpr int GBM = 256, GBN = 128;
constexpr int G_ASLAB = GBM * 32, G_APLANE = 2 * G_ASLAB;              // bytes: one 16-wide slab of 256 rows; hi plane (lo follows)
constexpr int G_BSLAB = GBN * 32, G_BPLANE = 2 * G_BSLAB, G_BSTAGE = 2 * G_BPLANE;

__global__ __launch_bounds__(256, 2) void gemm_sf16_kernel(GemmSfP p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G_APLANE + 2 * G_BSTAGE];   // 32 KB + 32 KB
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + 2 * G_APLANE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = p.N / GBN;
    const int logical = xcd_remap_g(blockIdx.x, gridDim.x);
    const int n0 = (logical % nb) * GBN;
    const int m0 = (logical / nb) * GBM;
    const int T = p.K >> 5;                                    // stages of 32

    const float sa = sed_sf_scale_of(amax_read(p.a_amax));
    const float inv = 1.0f / (sa * p.bscale[SED_AMAX_SLOTS]);

    // ---- A staging: item e = tid + 256*i (i < 8): row e >> 3, float4 quad e & 7 of the 32-wide stage
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.a), 0, (int)((unsigned)p.M * (unsigned)p.K * 4u), 0x00020000);
    const int q = tid & 7;
    int aoff[8], lso[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (tid + 256 * i) >> 3;
        aoff[i] = (m0 + row < p.M) ? ((m0 + row) * p.K + q * 4) * 4 : OOB;       // rows past M read as zero
        lso[i] = (q >> 2) * G_ASLAB + gsw(row, (q & 3) >> 1) + (q & 1) * 8;
    }
    float4 areg[8];
    bool overflow = false;
#define G_ALOAD(STG)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                               \
        areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ars, aoff[i], (STG) * 128, 0));
#define G_ASTORE()                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                             \
        float4 v = areg[i];                                                                                     \
        v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa;                                                             \
        overflow |= !((fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w)) < 3.0e5f);   /* inf AND NaN propagate */ \
        unsigned h01, l01, h23, l23;                                                                            \
        sed_sf_split2(v.x, v.y, h01, l01);                                                                      \
        sed_sf_split2(v.z, v.w, h23, l23);                                                                      \
        *reinterpret_cast<uint2*>(As + lso[i]) = make_uint2(h01, h23);                                          \
        *reinterpret_cast<uint2*>(As + G_APLANE + lso[i]) = make_uint2(l01, l23);                               \
    }

    // ---- B DMA: per stage 2 planes x 2 slabs x 4 blocks of 32 rows = 16 instructions of 64 lanes x 16 B; 4 per wave
    const int brow_in = lane >> 1;
    const int boff = (n0 + brow_in) * 32 + (((lane & 1) ^ ((brow_in >> 3) & 1)) << 4);     // bytes, thread-constant
    const unsigned bs_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Bs);
    const long b_plane_stride = (long)p.N * 32;               // bytes: next plane of a slab
    const long b_slab_stride = 2L * p.N * 32;                 // next 16-wide slab
#define G_BDMA1(STG, ST, j)                                                                                     \
    {                                                                                                           \
        const int qi = wv * 4 + (j);                                                                            \
        const int pl = qi >> 3, sl = (qi >> 2) & 1, rb = qi & 3;                                                \
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.bp) + (long)(2 * (STG) + sl) * b_slab_stride + \
                                   pl * b_plane_stride + rb * 32 * 32;                                          \
        unsigned keep_;                                                                                         \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_)                                                                             \
                     : "v"(boff), "s"(bs_base + (unsigned)((ST) * G_BSTAGE + pl * G_BPLANE + sl * G_BSLAB + rb * 32 * 32)), \
                       "s"(src)                                                                                 \
                     : "memory");                                                                               \
    }
#define G_BDMA(STG, ST) { G_BDMA1(STG, ST, 0) G_BDMA1(STG, ST, 1) G_BDMA1(STG, ST, 2) G_BDMA1(STG, ST, 3) }

    G_ALOAD(0)
    G_BDMA(0, 0)
    G_ASTORE()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- fragment addressing
    const int kh = lane >> 5;
    int aoffs[2], boffs[4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) aoffs[mb] = gsw(64 * wv + 32 * mb + (lane & 31), kh);
#pragma unroll
    for (int nk = 0; nk < 4; ++nk) boffs[nk] = gsw(32 * nk + (lane & 31), kh);

    floatx16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    for (int t = 0; t < T; ++t) {
        const int st = t & 1;
        if (t + 1 < T) {
            if (st) { G_BDMA(t + 1, 0) } else { G_BDMA(t + 1, 1) }
            G_ALOAD(t + 1)
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        const unsigned char* const Bst = Bs + st * G_BSTAGE;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            half8 ah[2], al[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                ah[mb] = *reinterpret_cast<const half8*>(As + sl * G_ASLAB + aoffs[mb]);
                al[mb] = *reinterpret_cast<const half8*>(As + G_APLANE + sl * G_ASLAB + aoffs[mb]);
            }
#pragma unroll
            for (int np = 0; np < 2; ++np) {                   // two column blocks at a time: 16 operand registers
                half8 bh[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bh[u] = *reinterpret_cast<const half8*>(Bst + sl * G_BSLAB + boffs[2 * np + u]);
                    bl[u] = *reinterpret_cast<const half8*>(Bst + G_BPLANE + sl * G_BSLAB + boffs[2 * np + u]);
                }
                // the three products of a tile are spread over the four tiles: no MFMA waits for its predecessor's result
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[mb][2 * np + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], bh[u], acc[mb][2 * np + u], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[mb][2 * np + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bl[u], acc[mb][2 * np + u], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        acc[mb][2 * np + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bh[u], acc[mb][2 * np + u], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < T) {                                       // every wave is done with this stage's A image: replace it
            G_ASTORE()
            __syncthreads();
        }
    }
#undef G_ALOAD
#undef G_ASTORE
#undef G_BDMA1
#undef G_BDMA

    if (overflow) {
        if (p.err_host) __hip_atomic_store(p.err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (p.err_dev) __hip_atomic_store(p.err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- epilogue: accumulator register r of a 32-row block holds row (r & 3) + 8 * (r >> 2) + 4 * kh, column lane & 31
    float amax = 0.f;
#pragma unroll
    for (int nk = 0; nk < 4; ++nk) {
        const int col = n0 + 32 * nk + (lane & 31);
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wv + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float v = fmaf(acc[mb][nk][r], inv, bv);
                if (row < p.M) {
                    p.c[(long)row * p.N + col] = v;
                    amax = fmaxf(amax, fabsf(v));
                }
            }
    }
    if (p.out_amax) {
        amax = wave_max(amax);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(p.out_amax) + ((blockIdx.x * 4 + wv) & (SED_AMAX_SLOTS - 1)), __float_as_uint(amax));
    }
}

// w [N][K] fp32 -> pack [K/16][hi, lo][N][16] f16 scaled by the power of two of wscale's amax slots; thread = (n, 4 k values)
__global__ __launch_bounds__(256) void gemm_pack_sf16_kernel(const float* __restrict__ w, int N, int K, float* __restrict__ wscale,
                                                             _Float16* __restrict__ wp) {
    const float sw = sed_sf_scale_of(amax_read(wscale));
    if (blockIdx.x == 0 && threadIdx.x == 0) wscale[SED_AMAX_SLOTS] = sw;
    const long total = (long)N * (K >> 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k4 = (int)(i % (K >> 2)), n = (int)(i / (K >> 2));
        const float4 v = *reinterpret_cast<const float4*>(w + (long)n * K + 4 * k4);
        unsigned h01, l01, h23, l23;
        sed_sf_split2(v.x * sw, v.y * sw, h01, l01);
        sed_sf_split2(v.z * sw, v.w * sw, h23, l23);
        const int ks = k4 >> 2, kk = (k4 & 3) * 4;                          // slab, position inside its 16
        _Float16* hi = wp + (((long)ks * 2 + 0) * N + n) * 16 + kk;
        _Float16* lo = wp + (((long)ks * 2 + 1) * N + n) * 16 + kk;
        *reinterpret_cast<uint2*>(hi) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(lo) = make_uint2(l01, l23);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// TN form: dw[n][k] = sum_m gy[m][n] * x[m][k] -- the weight gradients of the same dense layers (nn.GRU's W_ih / W_hh over
// 32 000 rows, MultiHead's four projections).  Both operands are row-major over m, i.e. K-major for this product, while an MFMA
// lane wants 8 consecutive m of ONE column: as in the convolutions' weight-gradient kernel (csrc/conv_sf16.hip) the tiles are
// staged as they come -- one 256-byte LDS row per m and 128-column operand tile, hi and lo planes -- and read with
// ds_read_b64_tr_b16, the LDS transpose read (a 16-lane group hands in the 8-byte pieces of four rows' 16 columns, lane c
// receives column c of the four rows).  16-column blocks are XOR-ed with 2 * (m & 3): the four rows of a transpose read then sit
// in four different 64-byte bank groups, and a 16-lane store group still writes 128 contiguous bytes.
// Workgroup = 128 n x 128 k, 4 waves = 2 n halves x 2 k halves (64 x 64 = four accumulators each), stage = 64 rows converted when
// staged (3.3 VALU per MFMA), register prefetch of the next stage; row slices over an XCD-aware grid, partial sums
// [slice][n][k] reduced in fp64 and unscaled by gemm_tn_sf16_reduce_kernel.
typedef _Float16 half4t __attribute__((ext_vector_type(4)));
typedef short short4t __attribute__((__vector_size__(4 * sizeof(short))));

struct GemmTnP {
    const float* gy;           // [M][N]
    const float* x;            // [M][K]
    float* partial;            // [nslices][N][K]
    const float* g_amax;
    const float* x_amax;
    int M, N, K;
    int stages_per_slice, nstages;
    int* err_host;
    int* err_dev;
};

__device__ __forceinline__ half4t tn_tr_read(const unsigned char* p) {
    return __builtin_bit_cast(half4t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4t*)(p)));
}

constexpr int TN_PLANE = 64 * 256;                       // bytes: 64 rows x 128 columns of f16

__global__ __launch_bounds__(256, 2) void gemm_tn_sf16_kernel(GemmTnP p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TN_PLANE];       // G hi, G lo, X hi, X lo: 64 KB
    unsigned char* const Gs = smem;
    unsigned char* const Xs = smem + 2 * TN_PLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv & 1, wk = wv >> 1;
    const int kt = p.K >> 7, ntile = kt * (p.N >> 7);
    const int logical = xcd_remap_g(blockIdx.x, gridDim.x);
    const int tile = logical % ntile, slice = logical / ntile;
    const int k0 = (tile % kt) * 128, n0 = (tile / kt) * 128;
    const int s0 = slice * p.stages_per_slice, s1 = min(p.nstages, s0 + p.stages_per_slice);
    const float sg = sed_sf_scale_of(amax_read(p.g_amax)), sx = sed_sf_scale_of(amax_read(p.x_amax));

    // ---- staging: item e = tid + 256*i (i < 8): row (tid >> 5) + 8*i of the stage, column quad tid & 31 (both operands)
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.gy), 0, (int)((unsigned)p.M * (unsigned)p.N * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((unsigned)p.M * (unsigned)p.K * 4u), 0x00020000);
    const int q = tid & 31, r0 = tid >> 5;
    const int ls0 = r0 * 256 + ((((q >> 2) ^ ((r0 & 3) << 1))) << 5) + (q & 3) * 8;    // + i * 8 * 256 (8*i keeps row & 3)
    float4 greg[8], xreg[8];
    bool overflow = false;
#define TN_LOAD(STG)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                             \
        const int row = (STG) * 64 + r0 + 8 * i;                                                                \
        const bool ok = row < p.M;                                                                              \
        greg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, ok ? (row * p.N + n0 + q * 4) * 4 : OOB, 0, 0)); \
        xreg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? (row * p.K + k0 + q * 4) * 4 : OOB, 0, 0)); \
    }
#define TN_SPLIT_STORE(V, S, BASE)                                                                              \
    {                                                                                                           \
        float4 v = V;                                                                                           \
        v.x *= S; v.y *= S; v.z *= S; v.w *= S;                                                                 \
        overflow |= !((fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w)) < 3.0e5f);                         \
        unsigned h01, l01, h23, l23;                                                                            \
        sed_sf_split2(v.x, v.y, h01, l01);                                                                      \
        sed_sf_split2(v.z, v.w, h23, l23);                                                                      \
        *reinterpret_cast<uint2*>(BASE + ls0 + i * 2048) = make_uint2(h01, h23);                                \
        *reinterpret_cast<uint2*>(BASE + TN_PLANE + ls0 + i * 2048) = make_uint2(l01, l23);                     \
    }
#define TN_STORE()                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                             \
        TN_SPLIT_STORE(greg[i], sg, Gs)                                                                         \
        TN_SPLIT_STORE(xreg[i], sx, Xs)                                                                         \
    }

    // ---- fragment addressing (lane-static): 16-lane group g16 -> column block (g16 & 1), k half (g16 >> 1)
    const int g16 = lane >> 4, cbl = g16 & 1, khalf = g16 >> 1, r4 = (lane >> 2) & 3, ch = lane & 3;
    int a_off[2], b_off[2];                              // + kk * 16 * 256 (+ 1024: rows 4..7 of the half) (+ TN_PLANE: lo)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        a_off[u] = (8 * khalf + r4) * 256 + (((4 * wn + 2 * u + cbl) ^ (r4 << 1)) << 5) + ch * 8;
        b_off[u] = (8 * khalf + r4) * 256 + (((4 * wk + 2 * u + cbl) ^ (r4 << 1)) << 5) + ch * 8;
    }

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    if (s0 < s1) {
        TN_LOAD(s0)
        TN_STORE()
        __syncthreads();
        for (int s = s0; s < s1; ++s) {
            const bool more = s + 1 < s1;
            if (more) { TN_LOAD(s + 1) }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const unsigned char* ap = Gs + a_off[u] + kk * 4096;
                    const half4t h0 = tn_tr_read(ap), h1 = tn_tr_read(ap + 1024);
                    const half4t l0 = tn_tr_read(ap + TN_PLANE), l1 = tn_tr_read(ap + TN_PLANE + 1024);
                    ah[u] = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    al[u] = half8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    const unsigned char* bp = Xs + b_off[u] + kk * 4096;
                    const half4t g0 = tn_tr_read(bp), g1 = tn_tr_read(bp + 1024);
                    const half4t m0 = tn_tr_read(bp + TN_PLANE), m1 = tn_tr_read(bp + TN_PLANE + 1024);
                    bh[u] = half8{g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                    bl[u] = half8{m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[c], acc[a][c], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[c], acc[a][c], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[c], acc[a][c], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (more) {
                TN_STORE()
                __syncthreads();
            }
        }
    }
#undef TN_LOAD
#undef TN_SPLIT_STORE
#undef TN_STORE

    if (overflow) {
        if (p.err_host) __hip_atomic_store(p.err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (p.err_dev) __hip_atomic_store(p.err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float* out = p.partial + (long)slice * p.N * p.K;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 64 * wn + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                out[(long)n * p.K + k0 + 64 * wk + 32 * c + (lane & 31)] = acc[a][c][r];
            }
}

// dw[e] (=|+=) sum over the slices (fp64, four independent chains, fixed order) / (sg * sx)
__global__ __launch_bounds__(256) void gemm_tn_sf16_reduce_kernel(const float* __restrict__ partial, int nslices, long nk,
                                                                  const float* __restrict__ g_amax, const float* __restrict__ x_amax,
                                                                  float* __restrict__ dw) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const double inv = 1.0 / ((double)sed_sf_scale_of(amax_read(g_amax)) * (double)sed_sf_scale_of(amax_read(x_amax)));
    if (e >= nk) return;
    double s[4] = {0., 0., 0., 0.};
    int q = 0;
    for (; q + 4 <= nslices; q += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += (double)partial[(long)(q + u) * nk + e];
    }
    for (int u = 0; q < nslices; ++q, ++u) s[u] += (double)partial[(long)q * nk + e];
    dw[e] = (float)(((s[0] + s[1]) + (s[2] + s[3])) * inv);
}

static void tn_slicing(long M, int N, int K, int* stages_per_slice, int* nstages, long* nslices) {
    const int ns = (int)((M + 63) / 64);
    const long tiles = (long)(N / 128) * (K / 128);
    long want = 1024 / tiles;                           // ~1024 workgroups: two rounds of the 512 resident ones
    if (want > ns / 4) want = ns / 4;                   // at least 4 stages per slice
    if (want < 1) want = 1;
    const int sps = (int)((ns + want - 1) / want);
    *stages_per_slice = sps; *nstages = ns; *nslices = (ns + sps - 1) / sps;
}

}  // namespace

SED_API int sed_gemm_nt_sf16_supported(long M, int N, int K) {
    return (M > 0 && N >= GBN && N % GBN == 0 && K >= 32 && K % 32 == 0 && (double)M * K * 4.0 < 2147483648.0) ? 1 : 0;
}

SED_API long sed_gemm_pack_sf16_halfs(int N, int K) { return 2L * N * K; }

// wscale: float[SED_AMAX_SLOTS + 1] (amax slots of w, then the scale the pack was written with); zeroed here unless it lies in a
// registered pre-zeroed pool.  Two launches: amax, pack.
SED_API int sed_gemm_pack_sf16(const float* w, int N, int K, float* wscale, void* wp, hipStream_t stream) {
    if (!w || !wscale || !wp || N <= 0 || K <= 0 || (K & 15)) return SED_EINVAL;
    int rc = sed_amax(w, (long)N * K, wscale, stream);
    if (rc != 0) return rc;
    long blocks = ((long)N * (K >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_pack_sf16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, N, K, wscale, (_Float16*)wp);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_gemm_nt_sf16(const float* x, const void* wp, const float* wscale, const float* bias, float* y, long M, int N, int K,
                             const float* x_amax, int* err_host, int* err_dev, float* out_amax, hipStream_t stream) {
    if (!x || !wp || !wscale || !y || !x_amax || !sed_gemm_nt_sf16_supported(M, N, K)) return SED_EINVAL;
    if (out_amax) {
        hipError_t e = sed_amax_clear(out_amax, stream);
        if (e != hipSuccess) return (int)e;
    }
    GemmSfP p;
    p.a = x; p.bp = (const _Float16*)wp; p.bscale = wscale; p.a_amax = x_amax; p.bias = bias; p.c = y;
    p.M = (int)M; p.N = N; p.K = K; p.err_host = err_host; p.err_dev = err_dev; p.out_amax = out_amax;
    const long nblk = ((M + GBM - 1) / GBM) * (N / GBN);
    if (nblk > 0x7fffffffL) return SED_EINVAL;
    hipLaunchKernelGGL(gemm_sf16_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, p);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_gemm_tn_sf16_supported(long M, int N, int K) {
    return (M > 0 && N >= 128 && N % 128 == 0 && K >= 128 && K % 128 == 0 && (double)M * N * 4.0 < 2147483648.0 &&
            (double)M * K * 4.0 < 2147483648.0) ? 1 : 0;
}

SED_API long sed_gemm_tn_sf16_partial_floats(long M, int N, int K) {
    if (!sed_gemm_tn_sf16_supported(M, N, K)) return 0;
    int sps, ns; long nsl;
    tn_slicing(M, N, K, &sps, &ns, &nsl);
    return nsl * N * K;
}

SED_API int sed_gemm_tn_sf16(const float* x, const float* gy, float* dw, float* partial, long M, int N, int K, const float* x_amax,
                             const float* gy_amax, int* err_host, int* err_dev, hipStream_t stream) {
    if (!x || !gy || !dw || !partial || !x_amax || !gy_amax || !sed_gemm_tn_sf16_supported(M, N, K)) return SED_EINVAL;
    GemmTnP p;
    p.gy = gy; p.x = x; p.partial = partial; p.g_amax = gy_amax; p.x_amax = x_amax; p.M = (int)M; p.N = N; p.K = K;
    p.err_host = err_host; p.err_dev = err_dev;
    long nsl;
    tn_slicing(M, N, K, &p.stages_per_slice, &p.nstages, &nsl);
    const long nblk = nsl * (N / 128) * (K / 128);
    hipLaunchKernelGGL(gemm_tn_sf16_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, p);
    SED_LAUNCH_CHECK();
    const long nk = (long)N * K;
    hipLaunchKernelGGL(gemm_tn_sf16_reduce_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, stream, partial, (int)nsl, nk,
                       gy_amax, x_amax, dw);
    SED_LAUNCH_CHECK();
    return 0;
}

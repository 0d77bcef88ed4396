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

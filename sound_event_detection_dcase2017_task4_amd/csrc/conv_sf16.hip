// 3x3 convolution on the f16 MFMA pipe with SPLIT operands (EXPERIMENTAL, round 2: forward-like use only, not wired
// into the models; tools/split_f16_study.py is the numerical study, tests/test_gpu_sf16.py the parity test).
//
//     x = (hi + lo) / s,  hi = f16(s*x),  lo = f16(s*x - hi)            (s = a power of two per tensor)
//     a*b ~= hi_a*hi_b + hi_a*lo_b + lo_a*hi_b                          (f16 x f16 is exact in the fp32 accumulator)
//
// Three v_mfma_f32_32x32x16_f16 replace one K = 16 slab of fp32 MACs: 3/16 of the fp32 MFMA issue time for the same
// direct-convolution MAC count, 0.42x of the fused Winograd F(2x2,3x3) kernels' MFMA time, at fp32-level error (1.5e-7
// relative L2 on this model's layer statistics, the same as a direct fp32 convolution; the dropped lo*lo term is 2^-22).
//
// Dataflow.  A workgroup owns 128 output pixels (TR = 128/W rows x W columns of one image) x 128 output channels; the
// four waves are 2 pixel halves x 2 channel halves, each a 64 x 64 register tile (4 accumulators).  K-step = 16 input
// channels: the (TR+2) x (W+2) input patch is converted ONCE to (hi, lo) f16 pairs when it is staged (32-byte LDS row per
// pixel and plane; the nine taps read shifted windows of it, so the conversion cost is amortised 9 x 128 times), the
// pre-split weights stream through LDS by LDS-DMA, three taps (one kernel row) per stage, double-buffered.
// 16-byte chunk index XOR ((row >> 3) & 1) keeps every ds_read_b128 conflict-free (consecutive pixels = consecutive
// rows, any tap shift).
#include "common.h"
#include "sed_hip.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int SF_AROWS = 264;                  // >= (TR+2) * (W+2): 4*66 at W = 64
constexpr int SF_APLANE = SF_AROWS * 32;       // bytes per plane (hi / lo)
constexpr int SF_BN = 128;                     // output channels per workgroup
constexpr int SF_BPLANE = 3 * SF_BN * 32;      // bytes: 3 taps x 128 co x 16 ci f16
constexpr int SF_BSTAGE = 2 * SF_BPLANE;

struct Sf16P {
    const float* x;            // [B][H][W][K]
    const _Float16* wp;        // [K/16][3 dy][2 planes][3 dx][N][16]
    float* y;                  // [B][H][W][N]
    const float* in_scale;
    const float* in_shift;
    int B, H, W, K, N;
    int logW, TR, ntile;
    float sa, inv;             // activation scale (power of two), 1 / (sa * sw)
};

__device__ __forceinline__ int sf_sw(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4); }

__device__ __forceinline__ int xcd_remap_sf(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <bool INT>
__global__ __launch_bounds__(256, 2) void conv_sf16_kernel(Sf16P p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SF_APLANE + 2 * SF_BSTAGE];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + 2 * SF_APLANE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wvu & 1, wn = wvu >> 1;
    const int nb = p.N / SF_BN;
    const int logical = xcd_remap_sf(blockIdx.x, gridDim.x);
    const int n0 = (logical % nb) * SF_BN;
    const int t = logical / nb;
    const int b = t / p.ntile, tile = t % p.ntile;
    const int W = p.W, logW = p.logW, TR = p.TR, WP = W + 2;
    const int h0 = tile * TR;
    const int KT = p.K >> 4;

    // ---- A staging: item e = tid + 256*i -> patch pixel e >> 2 (row rr, column c), channel quad e & 3
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (long)b * p.H * W * p.K, 0, (int)((unsigned)p.H * W * p.K * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(INT ? p.in_scale : p.x), 0, p.K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(INT ? p.in_shift : p.x), 0, p.K * 4, 0x00020000);
    const int q4 = tid & 3;
#define SF_META(i)                                                                                              \
    bool val##i, sok##i;                                                                                        \
    int lso##i, aoff##i;                                                                                        \
    float4 areg##i = make_float4(0.f, 0.f, 0.f, 0.f);                                                           \
    {                                                                                                           \
        const int pe = (tid + 256 * i) >> 2;                                                                    \
        const int rr = pe >> logW, c = pe & (W - 1);                                                            \
        const int h = h0 - 1 + rr;                                                                              \
        val##i = rr < TR + 2;                                                                                   \
        sok##i = val##i && (unsigned)h < (unsigned)p.H;                                                         \
        const int ridx = rr * WP + c + 1;                                                                       \
        lso##i = sf_sw(ridx, q4 >> 1) + (q4 & 1) * 8;                                                           \
        aoff##i = sok##i ? ((h * W + c) * p.K + q4 * 4) * 4 : OOB;                                              \
    }
    SF_META(0) SF_META(1) SF_META(2) SF_META(3)
#undef SF_META
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);

#define SF_ALOAD(i) areg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff##i, k_off, 0));
#define sf_aload(KS)                                                                                            \
    {                                                                                                           \
        const int k_off = (KS) * 64;                                                                            \
        if (INT) {                                                                                              \
            sc4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srs, q4 * 16, k_off, 0));    \
            sh4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hrs, q4 * 16, k_off, 0));    \
        }                                                                                                       \
        SF_ALOAD(0) SF_ALOAD(1) SF_ALOAD(2) SF_ALOAD(3)                                                         \
    }
#define SF_ASTORE(i)                                                                                            \
    if (val##i) {                                                                                               \
        float4 v = areg##i;                                                                                     \
        if (INT) {                                                                                              \
            v.x = sok##i ? fmaxf(fmaf(v.x, sc4.x, sh4.x), 0.f) : 0.f;                                           \
            v.y = sok##i ? fmaxf(fmaf(v.y, sc4.y, sh4.y), 0.f) : 0.f;                                           \
            v.z = sok##i ? fmaxf(fmaf(v.z, sc4.z, sh4.z), 0.f) : 0.f;                                           \
            v.w = sok##i ? fmaxf(fmaf(v.w, sc4.w, sh4.w), 0.f) : 0.f;                                           \
        }                                                                                                       \
        v.x *= p.sa; v.y *= p.sa; v.z *= p.sa; v.w *= p.sa;                                                     \
        const half4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};                          \
        const half4 lo = {(_Float16)(v.x - (float)hi.x), (_Float16)(v.y - (float)hi.y),                         \
                          (_Float16)(v.z - (float)hi.z), (_Float16)(v.w - (float)hi.w)};                        \
        *reinterpret_cast<half4*>(As + lso##i) = hi;                                                            \
        *reinterpret_cast<half4*>(As + SF_APLANE + lso##i) = lo;                                                \
    }
#define sf_astore() { SF_ASTORE(0) SF_ASTORE(1) SF_ASTORE(2) SF_ASTORE(3) }

    // ---- B DMA: 24 instructions of 64 lanes x 16 B per stage (2 planes x 3 taps x 4 blocks of 32 rows), 6 per wave
    const int brow_in = lane >> 1;
    const int boff = ((n0 + brow_in) * 32 + (((lane & 1) ^ ((brow_in >> 3) & 1)) << 4));     // bytes, thread-constant
    const unsigned bs_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Bs);
    const long b_dx_stride = (long)p.N * 32;            // bytes: next dx
    const long b_plane_stride = 3L * p.N * 32;          // next plane
    const long b_step_stride = 6L * p.N * 32;           // next (ks, dy)
#define SF_BDMA(STEP, ST, j)                                                                                    \
    {                                                                                                           \
        const int qi = wvu * 6 + (j);                                                                           \
        const int pl = qi / 12, dxx = (qi >> 2) % 3, rb = qi & 3;                                               \
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wp) + (long)(STEP) * b_step_stride + \
                                   pl * b_plane_stride + dxx * b_dx_stride + rb * 32 * 32;                      \
        unsigned keep_;                                                                                         \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_)                                                                             \
                     : "v"(boff), "s"(bs_base + (unsigned)((ST) * SF_BSTAGE + pl * SF_BPLANE + (dxx * SF_BN + rb * 32) * 32)), \
                       "s"(src)                                                                                 \
                     : "memory");                                                                               \
    }
#define sf_bdma(STEP, ST) { SF_BDMA(STEP, ST, 0) SF_BDMA(STEP, ST, 1) SF_BDMA(STEP, ST, 2) SF_BDMA(STEP, ST, 3) SF_BDMA(STEP, ST, 4) SF_BDMA(STEP, ST, 5) }

    // zero the halo columns (column -1 and column W of every patch row, both planes): never written by the staging
    for (int i = tid; i < (TR + 2) * 8; i += 256) {
        const int rr = i >> 3, side = (i >> 2) & 1, pl = (i >> 1) & 1, ch = i & 1;
        const int ridx = rr * WP + (side ? W + 1 : 0);
        *reinterpret_cast<float4*>(As + pl * SF_APLANE + sf_sw(ridx, ch)) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    sf_aload(0);
    sf_bdma(0, 0);
    sf_astore();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- fragment addressing
    const int kh = lane >> 5;
    int aoffs[2][9], boffs[2][3];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int pix = 64 * wm + 32 * mb + (lane & 31);
        const int r = pix >> logW, c = pix & (W - 1);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) aoffs[mb][tp] = sf_sw((r + tp / 3) * WP + c + tp % 3, kh);
    }
#pragma unroll
    for (int nk = 0; nk < 2; ++nk) {
        const int row = 64 * wn + 32 * nk + (lane & 31);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) boffs[nk][dx] = sf_sw(dx * SF_BN + row, kh);
    }

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    const int nsteps = KT * 3;
    int step = 0;
    for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy, ++step) {
            const int st = step & 1;
            if (step + 1 < nsteps) {
                if (st) { sf_bdma(step + 1, 0) } else { sf_bdma(step + 1, 1) }
            }
            if (dy == 0 && ks + 1 < KT) sf_aload(ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            const unsigned char* const Bst = Bs + st * SF_BSTAGE;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    ah[mb] = *reinterpret_cast<const half8*>(As + aoffs[mb][dy * 3 + dx]);
                    al[mb] = *reinterpret_cast<const half8*>(As + SF_APLANE + aoffs[mb][dy * 3 + dx]);
                }
#pragma unroll
                for (int nk = 0; nk < 2; ++nk) {
                    bh[nk] = *reinterpret_cast<const half8*>(Bst + boffs[nk][dx]);
                    bl[nk] = *reinterpret_cast<const half8*>(Bst + SF_BPLANE + boffs[nk][dx]);
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nk = 0; nk < 2; ++nk) {
                        acc[mb][nk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bh[nk], acc[mb][nk], 0, 0, 0);
                        acc[mb][nk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mb], bl[nk], acc[mb][nk], 0, 0, 0);
                        acc[mb][nk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mb], bh[nk], acc[mb][nk], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (dy == 2 && ks + 1 < KT) {          // every wave is done with this k-step's patch: replace it
                sf_astore();
                __syncthreads();
            }
        }
    }
#undef SF_ALOAD
#undef sf_aload
#undef SF_ASTORE
#undef sf_astore
#undef SF_BDMA
#undef sf_bdma

    // ---- epilogue: unscale, store (rows past the image fall outside the descriptor and are dropped)
    const unsigned y_img_bytes = (unsigned)p.H * W * p.N * 4u;
    const __amdgpu_buffer_rsrc_t yrs =
        __builtin_amdgcn_make_buffer_rsrc(p.y + (long)b * p.H * W * p.N, 0, (int)y_img_bytes, 0x00020000);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pix = 64 * wm + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int h = h0 + (pix >> logW), c = pix & (W - 1);
            const int off = ((h * W + c) * p.N + n0 + 64 * wn + (lane & 31)) * 4;
#pragma unroll
            for (int nk = 0; nk < 2; ++nk)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mb][nk][r] * p.inv), yrs, off, nk * 128, 0);
        }
}

// OIHW fp32 -> [K/16][3 dy][2 planes (hi, lo)][3 dx][N][16] f16, values scaled by sw.  dgrad = 1: the operand of the
// transposed convolution (roles of the channel axes swapped, taps flipped).
__global__ __launch_bounds__(256) void pack_sf16_kernel(const float* __restrict__ w, int Cout, int Cin, int dgrad, float sw,
                                                        _Float16* __restrict__ wp) {
    const int No = dgrad ? Cin : Cout, Ki = dgrad ? Cout : Cin;
    const long total = 9L * No * Ki;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int il = (int)(e & 15);
        long q = e >> 4;
        const int o = (int)(q % No); q /= No;
        const int dx = (int)(q % 3); q /= 3;
        const int dy = (int)(q % 3); q /= 3;
        const int ks = (int)q;
        const int i = ks * 16 + il;
        const float v = (dgrad ? w[(((long)i * Cin + o) * 3 + (2 - dy)) * 3 + (2 - dx)]
                               : w[(((long)o * Cin + i) * 3 + dy) * 3 + dx]) * sw;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const long base = ((long)(ks * 3 + dy) * 2) * 3 * No * 16;
        wp[base + ((long)dx * No + o) * 16 + il] = hi;
        wp[base + 3L * No * 16 + ((long)dx * No + o) * 16 + il] = lo;
    }
}

}  // namespace

SED_API int sed_conv3x3_sf16_supported(int H, int W, int Cin, int Cout) {
    return (W == 8 || W == 16 || W == 32 || W == 64) && H >= 1 && Cin % 16 == 0 && Cout % SF_BN == 0;
}

SED_API long sed_conv_sf16_pack_halfs(int Cin, int Cout) { return 18L * Cin * Cout; }

SED_API int sed_pack_conv_weights_sf16(const float* w_oihw, int Cout, int Cin, int dgrad, float sw, void* wp,
                                       sed_stream_t stream) {
    if (!w_oihw || !wp || Cout <= 0 || Cin <= 0 || (dgrad ? Cout : Cin) % 16) return SED_EINVAL;
    const long total = 9L * Cout * Cin;
    hipLaunchKernelGGL(pack_sf16_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w_oihw, Cout, Cin, dgrad, sw, (_Float16*)wp);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_conv3x3_sf16(const float* x, const void* wp, float* y, int B, int H, int W, int Cin, int Cout,
                             const float* in_scale, const float* in_shift, float sa, float sw, sed_stream_t stream) {
    if (!x || !wp || !y || B <= 0 || !sed_conv3x3_sf16_supported(H, W, Cin, Cout) || !(sa > 0.f) || !(sw > 0.f))
        return SED_EINVAL;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return SED_EINVAL;
    Sf16P p;
    p.x = x; p.wp = (const _Float16*)wp; p.y = y; p.in_scale = in_scale; p.in_shift = in_shift;
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout;
    p.logW = W == 64 ? 6 : W == 32 ? 5 : W == 16 ? 4 : 3;
    p.TR = 128 >> p.logW;
    p.ntile = (H + p.TR - 1) / p.TR;
    p.sa = sa; p.inv = 1.0f / (sa * sw);
    const long nblk = (long)B * p.ntile * (Cout / SF_BN);
    if (nblk > 0x7fffffffL) return SED_EINVAL;
    if (in_scale)
        hipLaunchKernelGGL(conv_sf16_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(conv_sf16_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    SED_LAUNCH_CHECK();
    return 0;
}

// 3x3 convolution (forward and dgrad) on the f16 MFMA pipe with SPLIT operands.
//
//     x = (hi + lo) / s,  hi = f16(s*x),  lo = f16(s*x - hi)            (s = a power of two per tensor)
//     a*b ~= hi_a*hi_b + hi_a*lo_b + lo_a*hi_b                          (f16 x f16 is exact in the fp32 accumulator)
//
// Three v_mfma_f32_32x32x16_f16 replace one K = 16 slab of fp32 MACs: 3/16 of the fp32 MFMA issue time for the same
// direct-convolution MAC count, 0.42x of the fused Winograd F(2x2,3x3) kernels' MFMA time, at the error of a direct fp32
// convolution (tools/split_f16_study.py: 1.5e-7 relative L2 on this model's layer statistics; the dropped lo*lo term is
// 2^-22; tests/test_gpu_sf16.py checks the kernel against float64).  Scales: the weights' from their amax (pack kernels,
// kept on the device), every activation / gradient operand's from an amax its PRODUCER kernel left on the device (pooled
// outputs: the pool kernel; relu(bn(y)) operands that are never materialised: per-channel max / min of y from the conv
// epilogue, pushed through the BatchNorm affine by sed_act_amax; gradients: the BatchNorm-backward apply kernels) -- so
// a finite operand can neither overflow nor fall into the f16 subnormals at any magnitude, and there is no host
// synchronisation.  A NON-FINITE operand raises the error words (host-mapped + device; the Adam kernel skips its update).
//
// Dataflow.  A workgroup owns 64*MW output pixels (TR rows x W columns of one image) x 64*NW output channels, MW x NW = 4
// waves, each a 64 x 64 register tile (4 accumulators): MW = 4 (256 pixels x 64 channels, three workgroups per CU) by
// default, MW = 2 (128 x 128, two per CU) on request.  K-step = 16 input channels: the (TR+2) x (W+2) input patch is converted ONCE to (hi, lo) f16
// planes when it is staged (32-byte LDS row per pixel and plane; the nine taps read shifted windows of it, so the
// conversion is amortised over 9 taps x all output channels), the pre-split weights stream through LDS by LDS-DMA, three
// taps (one kernel row) per stage, double-buffered.  16-byte chunk index XOR ((row >> 3) & 1) keeps every ds_read_b128
// conflict-free (consecutive pixels = consecutive rows, for any tap shift).
//
// Epilogue 3 (inference): eval-mode BatchNorm (folded to scale / shift) + ReLU + average pooling in
// registers -- the full-resolution conv output is never written; the pooled tensor and its amax are.
// Fusions as in conv_wino2.hip: input relu(scale*x+shift); epilogue 1 = BN statistics (sum, M2) per WORKGROUP tile (the
// waves' 64-pixel statistics merged through LDS, Chan's formula) + the per-part pixel count; epilogue 2 = ReLU mask of the
// previous activation + BN-backward sums.
#include "common.h"
#include "sed_hip.h"
#include <stdlib.h>
SED_OBJECT_FLAGS(conv_sf16)

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct Sf16P {
    const float* x;            // [B][H][W][K]
    const _Float16* wp;        // [K/16][3 dy][2 planes][3 dx][N][16]
    const float* wscale;       // [SED_AMAX_SLOTS + 1]: weights' amax slots, then the scale sw (written by the pack kernels)
    const float* x_amax;       // device: amax of the operand AS THE MFMAs SEE IT (relu(scale*x+shift) when fused)
    float* y;                  // [B][H][W][N]
    const float* in_scale;
    const float* in_shift;
    float* partials;           // EPI 1: [nparts][2][N] + [nparts] counts; EPI 2: [nparts][2][N]; nparts = B*ntile
    const float* yprev;
    const float* p_scale;
    const float* p_shift;
    const float* p_mean;
    const float* p_invstd;
    int B, H, W, K, N;
    int logW, TR, ntile;
    float* mm;                 // nullable: per-part (max, min) of the outputs per channel, [nparts][2][N] (EPI 0 / 1)
    int* err_host;             // nullable, host-mapped: set to 1 when an operand is not finite
    int* err_dev;              // nullable, device: same (read by sed_adam_amsgrad)
    float* pool_amax;          // EPI 3: amax slots of the pooled output
    float* out_amax;           // nullable (EPI 0 / 1 / 2 / 4): amax slots of |y| as written (masked, unscaled) -- one atomic per wave
    int ph, pw;                // EPI 3: pooling window, (2, 2) or (1, W)
    // EPI 4 (block 1's dgrad): the previous activations y1 = conv1(x0) are RECOMPUTED from the one-channel input
    const float* x0;           // [B][H][W]
    const float* w1;           // conv1 weights [64][9] (OIHW with I = 1)
    // SPLITK: ksplit workgroups share one output tile, each over its range of K-steps; they leave their raw accumulators in ws
    // [split tile][ksplit][4 waves][64 registers][64 lanes] and the LAST one to arrive (per-tile ticket in `tickets`,
    // self-resetting) adds the others' and runs the epilogue.  Workgroups [0, nfull) of the grid are ordinary UN-split tiles, the
    // rest are the shares of tiles nfull ..: nfull = 0 is the small-M form (fewer tiles than resident slots: every tile is split),
    // nfull = a multiple of the 768 resident slots is the TAIL split (the tiles of the last, partial round of the chip are split
    // so that their shares fill it; the full rounds in front of them run exactly as in the un-split kernel)
    int ksplit;
    int nfull;
    float* ws;
    int* tickets;
};

__device__ __forceinline__ int sf_sw(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4); }
// Patch (A) rows.  A ds_read_b128 is served in groups of 16 lanes -- pixels {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a
// 32-pixel MFMA block -- and is conflict-free when their 16-byte pieces fall on 16 different slots of a 256-byte bank row:
// every LDS row index mod 8 exactly twice, the two in different halves (the XOR key).  With W >= 32 the 32 pixels are
// consecutive LDS rows and key = (row >> 3) & 1 does that for any tap shift.  With W = 16 a block is two image rows and
// with W = 8 four, WP = W + 2 LDS rows apart: the pairs sharing a row index mod 8 then sit in image rows of different
// parity, so the key is the PATCH ROW's parity, and W = 8 needs a row pitch = 4 (mod 8) -- 12 instead of 10 -- for the
// "exactly twice" part.  (Round 2 used (row >> 3) & 1 everywhere: 47 % of the LDS cycles of the W = 8 layers were bank
// conflicts, profiles/r03.)
__device__ __forceinline__ int sf_swA(int rr, int f, int chunk, bool rowkey) {
    return f * 32 + ((chunk ^ ((rowkey ? rr : (f >> 3)) & 1)) << 4);
}

__device__ __forceinline__ int xcd_remap_sf(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// power of two that brings a tensor of this amax to [2^13, 2^14)
__device__ __forceinline__ float sf_scale_of(float amax) {
    if (!(amax > 0.f) || !(amax < __builtin_inff())) return 1.f;     // zero tensor; NaN / inf are reported by the staging code
    int e;
    frexpf(amax, &e);
    e = 14 - e;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return ldexpf(1.f, e);
}

// (hi, lo) f16 pairs of two fp32 values, hi = f16(v) (round to nearest even), lo = f16(v - hi): the difference is exact
// in fp32, so ONE mixed-precision fma per value (f16 hi read straight from its packed half, fp32 v, f16 result written into
// its half of the packed destination) replaces cvt-back + subtract + cvt + pack -- 3 VALU per pair instead of 8.
__device__ __forceinline__ void sf_split2(float a, float b, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(a), "v"(b));
}

// PRE: the input is ALREADY split (per channel pair {hi0 | hi1 << 16, lo0 | lo1 << 16}, scaled by the power of two of
// x_amax: conv1_act_sf16_kernel) -- the staging is a plain copy
// One v_fma_f32, opaque to the SLP vectoriser: left to itself it pairs neighbouring pixels' fmas into v_pk_fma_f32 with the
// weight broadcast from the HIGH half of src1 (op_sel:[0,1,0]) -- the operand form that returns wrong values beside LDS-fed
// f16 MFMAs (tests/test_isa_audit.py caught exactly that in the first build of epilogue 4; the co-resident workgroups of
// this very kernel are such neighbours)
__device__ __forceinline__ float fma_scalar(float a, float b, float c) {
    float d;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// SPLITK = the SMALL-LAUNCH form (round 6): launches of at most two workgroups per CU -- 4 to 8 clips per GPU, the strong-scaling
// regime -- where nothing co-resident hides a workgroup's own latencies.  It (i) may split the K range of a tile over workgroups
// (below) and (ii) streams the weights through FOUR stage buffers instead of two, requested three stages ahead: with one or two
// waves per SIMD a stage computes for ~0.55 us and then sat ~0.7 us behind the LDS-DMA it had requested at its own start (L2 /
// Infinity-Cache round trip); now only the last stage of a K-step waits, for the request it made itself.  74 KB of LDS and up to
// 256 VGPRs: two workgroups per CU, which is all such a launch has.
template <int MW, bool INT, int EPI, bool PRE = false, bool SPLITK = false>
__global__ __launch_bounds__(256, (MW == 4 && !SPLITK) ? 3 : 2) void conv_sf16_kernel(Sf16P p) {
    constexpr int NW = 4 / MW, BN = 64 * NW, RB = BN / 32;
    constexpr int AROWS = MW == 2 ? 264 : 408;         // >= (TR+2) * WP over the supported W (W = 8: 34 * 12)
    constexpr int APLANE = AROWS * 32;
    constexpr int BPLANE = 3 * BN * 32, BSTAGE = 2 * BPLANE;
    constexpr int NB = SPLITK ? 4 : 2;                 // weight stage buffers; stages are requested NB - 1 ahead
    constexpr int NI = MW == 2 ? 4 : 6;                // staging items per thread
    constexpr int NDMA = 6 * RB / 4;                   // LDS-DMA instructions per wave and stage
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * APLANE + NB * BSTAGE];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + 2 * APLANE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wvu % MW, wn = wvu / MW;
    const int nb = p.N / BN;
    // SPLITK: workgroups [0, nfull) are un-split tiles (dispatched first: the full rounds), the rest are shares; the ksplit shares
    // of a tile are neighbours in the logical order (same XCD, same time)
    int ksid = 0, logical, ks_here = 1;
    if (SPLITK && (int)blockIdx.x >= p.nfull) {
        const int tl = xcd_remap_sf((int)blockIdx.x - p.nfull, (int)gridDim.x - p.nfull);
        ks_here = p.ksplit;
        ksid = tl % ks_here;
        logical = p.nfull + tl / ks_here;
    } else {
        logical = xcd_remap_sf(blockIdx.x, SPLITK ? p.nfull : (int)gridDim.x);
    }
    const int n0 = (logical % nb) * BN;
    const int t = logical / nb;
    const int b = t / p.ntile, tile = t % p.ntile;
    const int W = p.W, logW = p.logW, TR = p.TR, WP = W == 8 ? 12 : W + 2;
    const bool rowkey = W < 32;
    const int h0 = tile * TR;
    const int KTALL = p.K >> 4;
    const int k_begin = SPLITK ? (int)((long)ksid * KTALL / ks_here) : 0;
    const int KT = SPLITK ? (int)((long)(ksid + 1) * KTALL / ks_here) : KTALL;       // K-steps [k_begin, KT)

    const float sa = sf_scale_of(amax_read(p.x_amax));
    const float inv = 1.0f / (sa * p.wscale[SED_AMAX_SLOTS]);

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
    bool val##i = false, sok##i = false;                                                                        \
    int lso##i = 0, aoff##i = OOB;                                                                              \
    float4 areg##i = make_float4(0.f, 0.f, 0.f, 0.f);                                                           \
    if (i < NI) {                                                                                               \
        const int pe = (tid + 256 * i) >> 2;                                                                    \
        const int rr = pe >> logW, c = pe & (W - 1);                                                            \
        const int h = h0 - 1 + rr;                                                                              \
        val##i = rr < TR + 2;                                                                                   \
        sok##i = val##i && (unsigned)h < (unsigned)p.H;                                                         \
        const int ridx = rr * WP + c + 1;                                                                       \
        lso##i = sf_swA(rr, ridx, q4 >> 1, rowkey) + (q4 & 1) * 8;                                              \
        aoff##i = sok##i ? ((h * W + c) * p.K + q4 * 4) * 4 : OOB;                                              \
    }
    SF_META(0) SF_META(1) SF_META(2) SF_META(3) SF_META(4) SF_META(5)
#undef SF_META
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool overflow = false;      // a scaled operand outside the f16 range (or not finite): reported, never silent

#define SF_ALOAD(i) if (i < NI) areg##i = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff##i, k_off, 0));
#define sf_aload(KS)                                                                                            \
    {                                                                                                           \
        const int k_off = (KS) * 64;                                                                            \
        if (INT) {                                                                                              \
            sc4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srs, q4 * 16, k_off, 0));    \
            sh4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hrs, q4 * 16, k_off, 0));    \
            /* the power-of-two operand scale rides on the affine: relu(sa*sc*x + sa*sh) == sa*relu(sc*x + sh) bit for bit */ \
            sc4.x *= sa; sc4.y *= sa; sc4.z *= sa; sc4.w *= sa; sh4.x *= sa; sh4.y *= sa; sh4.z *= sa; sh4.w *= sa; \
        }                                                                                                       \
        SF_ALOAD(0) SF_ALOAD(1) SF_ALOAD(2) SF_ALOAD(3) SF_ALOAD(4) SF_ALOAD(5)                                 \
    }
#define SF_ASTORE(i)                                                                                            \
    if (PRE) {                                                                                                  \
        if (i < NI && val##i) {                    /* {h01, l01, h23, l23}; out-of-image loads returned 0 */     \
            const uint4 u = __builtin_bit_cast(uint4, areg##i);                                                 \
            *reinterpret_cast<uint2*>(As + lso##i) = make_uint2(u.x, u.z);                                      \
            *reinterpret_cast<uint2*>(As + APLANE + lso##i) = make_uint2(u.y, u.w);                             \
        }                                                                                                       \
    } else if (i < NI && (INT ? sok##i : val##i)) {   /* INT: rows outside the image were zeroed once and are never written */ \
        float4 v = areg##i;                                                                                     \
        if (INT) {                                                                                              \
            v.x = bn_relu(v.x, sc4.x, sh4.x); v.y = bn_relu(v.y, sc4.y, sh4.y);                                 \
            v.z = bn_relu(v.z, sc4.z, sh4.z); v.w = bn_relu(v.w, sc4.w, sh4.w);                                 \
        } else {                                                                                                \
            v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa;        /* out-of-image loads returned 0 */              \
        }                                                                                                       \
        overflow |= !((fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w)) < 3.0e5f);  /* 3 adds (|.| is a source modifier): inf AND NaN propagate; finite values are < 2^14 each */ \
        unsigned h01, l01, h23, l23;                                                                            \
        sf_split2(v.x, v.y, h01, l01);                                                                          \
        sf_split2(v.z, v.w, h23, l23);                                                                          \
        *reinterpret_cast<uint2*>(As + lso##i) = make_uint2(h01, h23);                                          \
        *reinterpret_cast<uint2*>(As + APLANE + lso##i) = make_uint2(l01, l23);                                 \
    }
#define sf_astore() { SF_ASTORE(0) SF_ASTORE(1) SF_ASTORE(2) SF_ASTORE(3) SF_ASTORE(4) SF_ASTORE(5) }

    // ---- B DMA: 6*RB instructions of 64 lanes x 16 B per stage (2 planes x 3 taps x RB blocks of 32 rows)
    const int brow_in = lane >> 1;
    const int boff = ((n0 + brow_in) * 32 + (((lane & 1) ^ ((brow_in >> 3) & 1)) << 4));     // bytes, thread-constant
    const unsigned bs_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Bs);
    const long b_dx_stride = (long)p.N * 32;            // bytes: next dx
    const long b_plane_stride = 3L * p.N * 32;          // next plane
    const long b_step_stride = 6L * p.N * 32;           // next (ks, dy)
#define SF_BDMA(STEP, ST, j)                                                                                    \
    if ((j) < NDMA) {                                                                                           \
        const int qi = wvu * NDMA + (j);                                                                        \
        const int pl = qi / (3 * RB), dxx = (qi / RB) % 3, rb = qi % RB;                                        \
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wp) + (long)(STEP) * b_step_stride + \
                                   pl * b_plane_stride + dxx * b_dx_stride + rb * 32 * 32;                      \
        unsigned keep_;                                                                                         \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_)                                                                             \
                     : "v"(boff), "s"(bs_base + (unsigned)((ST) * BSTAGE + pl * BPLANE + (dxx * BN + rb * 32) * 32)), \
                       "s"(src)                                                                                 \
                     : "memory");                                                                               \
    }
#define sf_bdma(STEP, ST) { SF_BDMA(STEP, ST, 0) SF_BDMA(STEP, ST, 1) SF_BDMA(STEP, ST, 2) SF_BDMA(STEP, ST, 3) SF_BDMA(STEP, ST, 4) SF_BDMA(STEP, ST, 5) }

    // zero the halo columns (column -1 and column W of every patch row, both planes): never written by the staging
    for (int i = tid; i < (TR + 2) * 8; i += 256) {
        const int rr = i >> 3, side = (i >> 2) & 1, pl = (i >> 1) & 1, ch = i & 1;
        const int ridx = rr * WP + (side ? W + 1 : 0);
        *reinterpret_cast<float4*>(As + pl * APLANE + sf_swA(rr, ridx, ch, rowkey)) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (INT) {      // patch rows outside the image (first / last tile): zero, once -- the staging skips them
#define SF_AZERO(i)                                                                                             \
        if (i < NI && val##i && !sok##i) {                                                                      \
            *reinterpret_cast<uint2*>(As + lso##i) = make_uint2(0u, 0u);                                        \
            *reinterpret_cast<uint2*>(As + APLANE + lso##i) = make_uint2(0u, 0u);                               \
        }
        SF_AZERO(0) SF_AZERO(1) SF_AZERO(2) SF_AZERO(3) SF_AZERO(4) SF_AZERO(5)
#undef SF_AZERO
    }
    sf_aload(k_begin);
    sf_bdma(k_begin * 3, 0);
    if (NB == 4) {                                     // (every K range holds at least one K-step = three stages)
        sf_bdma(k_begin * 3 + 1, 1);
        sf_bdma(k_begin * 3 + 2, 2);
    }
    sf_astore();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- fragment addressing
    const int kh = lane >> 5;
    int aoffs[2][9], boffs[2][3];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int pix = 64 * wm + 32 * mb + (lane & 31);
        int r = pix >> logW, c = pix & (W - 1);
        if (EPI == 3 && logW == 6) {       // pooling epilogue at W = 64: a wave owns 2 rows x 32 columns (whole 2x2 windows)
            r = 2 * (wm >> 1) + mb;
            c = 32 * (wm & 1) + (lane & 31);
        }
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) aoffs[mb][tp] = sf_swA(r + tp / 3, (r + tp / 3) * WP + c + tp % 3, kh, rowkey);
    }
#pragma unroll
    for (int nk = 0; nk < 2; ++nk) {
        const int row = 64 * wn + 32 * nk + (lane & 31);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) boffs[nk][dx] = sf_sw(dx * BN + row, kh);
    }

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    const int nsteps = (KT - k_begin) * 3, step0 = k_begin * 3;
    int step = 0;
    for (int ks = k_begin; ks < KT; ++ks) {
        half8 ah[2][2], al[2][2], bh[2][2], bl[2][2];          // [register set][block]; the small-launch form carries them across stages
#pragma unroll
        for (int dy = 0; dy < 3; ++dy, ++step) {
            const int st = step & (NB - 1);
            if (NB == 2) {
                if (step + 1 < nsteps) {
                    if (st) { sf_bdma(step0 + step + 1, 0) } else { sf_bdma(step0 + step + 1, 1) }
                }
            } else if (step + 3 < nsteps) {
                const int st3 = __builtin_amdgcn_readfirstlane((step + 3) & 3);      // the buffer the previous stage has just released
                sf_bdma(step0 + step + 3, st3);
            }
            if (dy == 0 && ks + 1 < KT) sf_aload(ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            const unsigned char* const Bst = Bs + st * BSTAGE;
            // (requesting the fragments of tap dx+1 before the MFMAs of tap dx -- two register sets -- measured +-1 % at three waves
            // per SIMD: the other waves already hide the LDS latency, and one set keeps the kernel at three waves.)
            // Small-launch form: one or two waves per SIMD and 256 registers.  Left to itself the scheduler sinks every ds_read
            // next to the MFMA that consumes it (~17 lgkmcnt waits per stage, each an exposed LDS round trip with nobody else on
            // the SIMD): here the nine taps of a K-step form ONE software pipeline over two register sets -- the fragments of tap
            // t + 1 are requested before the MFMAs of tap t, ACROSS the stage boundaries (the next stage's weights landed a K-step
            // ago and the patch does not change inside a K-step), pinned in place by scheduling barriers.  The LDS round trip is
            // exposed once per K-step.
#define SF_FRAGS(SET, DYV, DX, BSTV)                                                                            \
            {                                                                                                   \
                _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) {                                              \
                    ah[SET][mb] = *reinterpret_cast<const half8*>(As + aoffs[mb][(DYV) * 3 + (DX)]);            \
                    al[SET][mb] = *reinterpret_cast<const half8*>(As + APLANE + aoffs[mb][(DYV) * 3 + (DX)]);   \
                }                                                                                               \
                _Pragma("unroll") for (int nk = 0; nk < 2; ++nk) {                                              \
                    bh[SET][nk] = *reinterpret_cast<const half8*>((BSTV) + boffs[nk][DX]);                      \
                    bl[SET][nk] = *reinterpret_cast<const half8*>((BSTV) + BPLANE + boffs[nk][DX]);             \
                }                                                                                               \
            }
            const unsigned char* const Bnx = Bs + ((step + 1) & (NB - 1)) * BSTAGE;
            if (SPLITK && dy == 0) SF_FRAGS(0, 0, 0, Bst)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int cur = SPLITK ? ((3 * dy + dx) & 1) : 0;
                if (SPLITK) {
                    if (dx < 2) { if (cur) SF_FRAGS(0, dy, dx + 1, Bst) else SF_FRAGS(1, dy, dx + 1, Bst) }
                    else if (dy < 2) { if (cur) SF_FRAGS(0, dy + 1, 0, Bnx) else SF_FRAGS(1, dy + 1, 0, Bnx) }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    if (dx == 0) SF_FRAGS(0, dy, 0, Bst) else if (dx == 1) SF_FRAGS(0, dy, 1, Bst) else SF_FRAGS(0, dy, 2, Bst)
                }
                // the three products of a tile are spread over the four tiles: no MFMA waits for its predecessor's result
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nk = 0; nk < 2; ++nk)
                        acc[mb][nk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][mb], bh[cur][nk], acc[mb][nk], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nk = 0; nk < 2; ++nk)
                        acc[mb][nk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][mb], bl[cur][nk], acc[mb][nk], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nk = 0; nk < 2; ++nk)
                        acc[mb][nk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][mb], bh[cur][nk], acc[mb][nk], 0, 0, 0);
                if (SPLITK) __builtin_amdgcn_sched_barrier(0);
            }
#undef SF_FRAGS
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            // NB = 2: the next stage's weights (requested at the start of this one) must have landed.  NB = 4: the three stages of
            // the NEXT K-step were requested during this one, each into the buffer released a stage earlier; everything a stage
            // reads was requested one K-step ago and is drained HERE, at the end of every K-step (with the patch loads the
            // staging below needs anyway) -- the first two stages of a K-step wait for nothing
            if (NB == 2 || dy == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

    if (overflow) {
        if (p.err_host) __hip_atomic_store(p.err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (p.err_dev) __hip_atomic_store(p.err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    if (SPLITK && ks_here > 1) {
        // ---- the K ranges of this tile meet: everyone leaves its accumulators, the last to take a ticket collects them.  The
        // shares travel with agent-scope (sc1) stores and loads -- they go to the coherence point, whichever XCDs the workgroups
        // run on -- and NO fence: an agent-scope release / acquire pair writes back and invalidates a whole L2 (~10 us per
        // workgroup on this 8-XCD part, csrc/gru.hip; the first build of this path used them and was 10-100 % slower than not splitting)
        // (What this rests on, stated once: a relaxed agent-scope store is written THROUGH to the device coherence point and
        // `s_waitcnt vmcnt(0)` returns when that write has been acknowledged; the ticket is an agent-scope atomic executed at the
        // same point; the shares are read back with sc1 loads, which bypass the reader's own non-coherent cache levels.  The HIP
        // memory model promises this ordering only through fences; gfx950 delivers it for this pattern, and the stress test
        // tests/test_gpu_sf16.py::test_sf16_split_k_exchange_across_xcds holds it to bit equality over thousands of groups that
        // straddle XCDs.  The tickets are per stream and self-resetting; ops.clear_nonfinite_flags() / recover() zero them, so an
        // aborted launch cannot leave a later one without its epilogue.)
        __shared__ int sk_last;
        const long tile_id = logical - p.nfull;
        float* const mine = p.ws + ((tile_id * p.ksplit + ksid) * 4 + wvu) * 4096 + lane * 4;      // [16 chunks][64 lanes][4 floats]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float2 v2 = make_float2(acc[a][c][r], acc[a][c][r + 1]);
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine + (((a * 2 + c) * 16 + r) >> 2) * 256 + (r & 3)),
                                       __builtin_bit_cast(unsigned long long, v2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the stores have reached the coherence point
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(p.tickets + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sk_last = old == p.ksplit - 1;
            if (sk_last) __hip_atomic_store(p.tickets + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        if (!sk_last) return;
        // sum in the FIXED order 0 .. ksplit - 1, this workgroup's own share re-read like the others': whoever happens to be last,
        // the result has the same bits (training runs are reproducible, graph replays equal eager steps)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
        for (int o = 0; o < p.ksplit; ++o) {
            const float* const theirs = p.ws + ((tile_id * p.ksplit + o) * 4 + wvu) * 4096 + lane * 4;
            floatx4 q[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(q[j]) : "v"(theirs + j * 256) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(q[j]));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][c][r] += q[((a * 2 + c) * 16 + r) >> 2][r & 3];
        }
        __syncthreads();                   // (the epilogues re-use the staging memory behind a barrier of their own; keep the waves together)
    }
    if (EPI == 3) {
        // ---- inference epilogue: relu(scale * y + shift) -> average pool -> store (accumulator register r of a 32-pixel block
        // holds pixel i = (r & 3) + 4 * kh + 8 * (r >> 2) of it, channel = lane & 31)
        const int Hp = p.H / p.ph, Wq = W / p.pw;
        float* const outb = p.y + (long)b * Hp * Wq * p.N;
        float amax = 0.f;
#pragma unroll
        for (int nk = 0; nk < 2; ++nk) {
            const int col = n0 + 64 * wn + (lane & 31) + 32 * nk;
            const float sc = p.p_scale[col], sh = p.p_shift[col];
            float a[2][16];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) a[mb][r] = bn_relu(acc[mb][nk][r] * inv, sc, sh);
            if (p.pw == 2) {
                if (logW >= 5) {            // W = 64 (remapped above) / 32: block mb = one image row, 32 columns
                    const int prow = (h0 >> 1) + (logW == 6 ? (wm >> 1) : wm);
                    const int pc0 = logW == 6 ? 16 * (wm & 1) : 0;
                    if (prow < Hp) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const float v = 0.25f * ((a[0][4 * q + 2 * j] + a[0][4 * q + 2 * j + 1]) +
                                                         (a[1][4 * q + 2 * j] + a[1][4 * q + 2 * j + 1]));
                                outb[((long)prow * Wq + pc0 + j + 4 * q + 2 * kh) * p.N + col] = v;
                                amax = fmaxf(amax, v);
                            }
                    }
                } else {                    // W = 16: block mb = image rows (2 mb, 2 mb + 1) of the wave's four
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const int prow = (h0 >> 1) + 2 * wm + mb;
                        if (prow < Hp) {
#pragma unroll
                            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    const int r0 = 4 * qq + 2 * j;
                                    const float v = 0.25f * ((a[mb][r0] + a[mb][r0 + 1]) + (a[mb][r0 + 8] + a[mb][r0 + 9]));
                                    outb[((long)prow * Wq + j + 2 * kh + 4 * qq) * p.N + col] = v;
                                    amax = fmaxf(amax, v);
                                }
                        }
                    }
                }
            } else {                        // (1, W) average at W = 8: block = four image rows of eight columns
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float sm = (a[mb][4 * q] + a[mb][4 * q + 1]) + (a[mb][4 * q + 2] + a[mb][4 * q + 3]);
                        sm += __shfl_xor(sm, 32, 64);
                        const int prow = h0 + 8 * wm + 4 * mb + q;
                        const float v = sm * 0.125f;
                        if (kh == 0 && prow < Hp) {
                            outb[(long)prow * p.N + col] = v;
                            amax = fmaxf(amax, v);
                        }
                    }
            }
        }
        if (p.pool_amax) amax_publish_block(p.pool_amax, amax);
        return;
    }
    // ---- epilogue: unscale, (mask,) statistics, store (rows past the image fall outside the descriptor and are dropped)
    const unsigned y_img_bytes = (unsigned)p.H * W * p.N * 4u;
    const __amdgpu_buffer_rsrc_t yrs =
        __builtin_amdgcn_make_buffer_rsrc(p.y + (long)b * p.H * W * p.N, 0, (int)y_img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(EPI == 2 ? p.yprev + (long)b * p.H * W * p.N : p.x), 0, (int)y_img_bytes, 0x00020000);
    const int colb = n0 + 64 * wn + (lane & 31);       // + 32*nk
    const long part = (long)b * p.ntile + tile;        // ONE part per workgroup: the MW waves' sums are merged below
    float vabs = 0.f;                                  // max |y| this lane stores (p.out_amax)
    // per-wave (sum, M2 | second sum, max, min) of its 64 pixels per channel -> LDS (the staging buffers are free behind the
    // loop's last barrier) -> merged by wave wm = 0 of every channel half: a quarter of the partial rows in memory and in the
    // BatchNorm merge kernels (131 -> 33 MB of partials per launch on block 1 at B = 256; step time unchanged: measured)
    float* const red = reinterpret_cast<float*>(smem);             // [wave][nk][32 columns][4]
    // valid pixels of wave w's 64 (whole rows of W <= 64 pixels)
    auto wave_count = [&](int w) -> float {
        const int rows_w = 64 >> logW;
        int nv = p.H - (h0 + ((64 * w) >> logW));
        nv = nv < 0 ? 0 : (nv > rows_w ? rows_w : nv);
        return (float)(nv * W);
    };
    // EPI 4: the (TR + 2) x (W + 2) patch of the one-channel input x0 around this tile, zero outside the image, behind the
    // 4 KB of per-wave statistics in the (now free) staging memory
    float* const xpatch = reinterpret_cast<float*>(smem) + 2048;
    if (EPI == 4) {
        const float* const x0b = p.x0 + (long)b * p.H * W;
        for (int i = tid; i < (TR + 2) * (W + 2); i += 256) {
            const int rr = i / (W + 2), cc = i - rr * (W + 2);
            const int h = h0 - 1 + rr, w = cc - 1;
            xpatch[i] = ((unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)W) ? x0b[h * W + w] : 0.f;
        }
        __syncthreads();
    }
    int yoff[2][16];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pix = 64 * wm + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * kh;
            yoff[mb][r] = (((h0 + (pix >> logW)) * W + (pix & (W - 1))) * p.N + colb) * 4;
        }
#pragma unroll
    for (int nk = 0; nk < 2; ++nk) {
        const int col = colb + 32 * nk;
        float s1 = 0.f, s2 = 0.f;
        float vmx = -__builtin_inff(), vmn = __builtin_inff();
        float e_sc = 0.f, e_sh = 0.f, e_mu = 0.f, e_is = 0.f;
        if (EPI == 2) { e_sc = p.p_scale[col]; e_sh = p.p_shift[col]; e_mu = p.p_mean[col]; e_is = p.p_invstd[col]; }
        float w1r[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (EPI == 4) {
            e_sc = p.p_scale[col]; e_sh = p.p_shift[col]; e_mu = p.p_mean[col]; e_is = p.p_invstd[col];
#pragma unroll
            for (int t = 0; t < 9; ++t) w1r[t] = p.w1[col * 9 + t];
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float yp[16];
            if (EPI == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    yp[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, yoff[mb][r], nk * 128, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (EPI == 4 && (r & 3) == 0) {
                    // y1 = conv1(x0) at the next four of this lane's pixels (consecutive pixels of one image row), recomputed from
                    // the x0 patch in LDS with the fma sequence of conv1_fwd_kernel (taps in order, from zero): bit-identical
                    // to the tensor the round-3 dataflow read back here -- 4.2 GB per step at batch 256 that never exist now
                    const int pix0 = 64 * wm + 32 * mb + 2 * r + 4 * kh;
                    const float* wp0 = xpatch + (pix0 >> logW) * (W + 2) + (pix0 & (W - 1));
                    float win[3][6];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const float2 t2 = *reinterpret_cast<const float2*>(wp0 + dy * (W + 2) + 2 * j);
                            win[dy][2 * j] = t2.x; win[dy][2 * j + 1] = t2.y;
                        }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float o = 0.f;
#pragma unroll
                        for (int t = 0; t < 9; ++t) o = fma_scalar(win[t / 3][j + t % 3], w1r[t], o);
                        yp[r + j] = o;
                    }
                }
                float v = acc[mb][nk][r] * inv;
                const bool ok = (unsigned)yoff[mb][r] < y_img_bytes;
                if (EPI == 2 || EPI == 4) {
                    v = (ok && bn_relu_active(yp[r], e_sc, e_sh)) ? v : 0.f;
                    s2 = fmaf(v, (yp[r] - e_mu) * e_is, s2);
                    s1 += v;
                } else if (EPI == 1) {
                    v = ok ? v : 0.f;
                    s1 += v;
                }
                if (EPI != 2 && EPI != 4 && p.mm) {
                    vmx = fmaxf(vmx, ok ? v : -__builtin_inff());
                    vmn = fminf(vmn, ok ? v : __builtin_inff());
                }
                acc[mb][nk][r] = v;
                vabs = fmaxf(vabs, ok ? fabsf(v) : 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, yoff[mb][r], nk * 128, 0);
            }
        }
        float* const mine = red + ((wvu * 2 + nk) * 32 + (lane & 31)) * 4;
        if (EPI != 2 && EPI != 4 && p.mm) {      // range of this wave's 64 pixels per channel: the consumer's operand amax comes from it
            vmx = fmaxf(vmx, __shfl_xor(vmx, 32, 64));
            vmn = fminf(vmn, __shfl_xor(vmn, 32, 64));
            if (kh == 0) { mine[2] = vmx; mine[3] = vmn; }
        }
        if (EPI == 1) {
            const float cnt = wave_count(wm);
            s1 += __shfl_xor(s1, 32, 64);
            const float mean = cnt > 0.f ? s1 / cnt : 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = (unsigned)yoff[mb][r] < y_img_bytes;
                    const float d = ok ? acc[mb][nk][r] - mean : 0.f;
                    s2 = fmaf(d, d, s2);
                }
            s2 += __shfl_xor(s2, 32, 64);
            if (kh == 0) { mine[0] = s1; mine[1] = s2; }
        }
        if (EPI == 2 || EPI == 4) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (kh == 0) { mine[0] = s1; mine[1] = s2; }
        }
    }
    if (p.out_amax) {      // the consumer's split-f16 scale (or the bound of what a BatchNorm-backward pass makes of this tensor)
        vabs = wave_max(vabs);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(p.out_amax) + ((blockIdx.x * 4 + wvu) & (SED_AMAX_SLOTS - 1)), __float_as_uint(vabs));
    }
    if (EPI == 0 && !p.mm) return;
    __syncthreads();
    if (wm != 0) return;
    {
        const int nk = lane >> 5, col = colb + 32 * nk;            // lane -> one of this wave's 64 columns
        const float* const theirs = red + (((wn * MW) * 2 + nk) * 32 + (lane & 31)) * 4;       // + w * 256: wave (w, wn)
        if (EPI != 2 && EPI != 4 && p.mm) {
            float vmx = theirs[2], vmn = theirs[3];
#pragma unroll
            for (int w = 1; w < MW; ++w) { vmx = fmaxf(vmx, theirs[w * 256 + 2]); vmn = fminf(vmn, theirs[w * 256 + 3]); }
            p.mm[(part * 2 + 0) * p.N + col] = vmx;
            p.mm[(part * 2 + 1) * p.N + col] = vmn;
        }
        if (EPI == 1) {
            // Chan's merge of the waves' (count, sum, M2): M2 = sum_w M2_w + n_w (mean_w - mean)^2
            float S = 0.f, Nn = 0.f;
#pragma unroll
            for (int w = 0; w < MW; ++w) { S += theirs[w * 256]; Nn += wave_count(w); }
            const float mean = Nn > 0.f ? S / Nn : 0.f;
            float M2 = 0.f;
#pragma unroll
            for (int w = 0; w < MW; ++w) {
                const float nw = wave_count(w);
                const float d = nw > 0.f ? theirs[w * 256] / nw - mean : 0.f;
                M2 += theirs[w * 256 + 1] + nw * d * d;
            }
            p.partials[(part * 2 + 0) * p.N + col] = S;
            p.partials[(part * 2 + 1) * p.N + col] = M2;
            if (wn == 0 && n0 == 0 && lane == 0) p.partials[(long)p.B * p.ntile * 2 * p.N + part] = Nn;
        }
        if (EPI == 2 || EPI == 4) {
            float a = 0.f, c = 0.f;
#pragma unroll
            for (int w = 0; w < MW; ++w) { a += theirs[w * 256]; c += theirs[w * 256 + 1]; }
            p.partials[(part * 2 + 0) * p.N + col] = a;
            p.partials[(part * 2 + 1) * p.N + col] = c;
        }
    }
}

// ---- weights: amax -> power-of-two scale (device resident), then OIHW fp32 -> [K/16][3 dy][2 planes (hi, lo)][3 dx][N][16]
// f16.  dgrad = 1: the operand of the transposed convolution (roles of the channel axes swapped, taps flipped).
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    float m = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[e]));
    amax_publish_block(out, m);
}
// the same for activation-sized tensors (the dense layers' operands: 16 M elements): 16-byte loads, four of them in flight per
// thread, a grid that fills the chip -- the weight-sized form above walks 65 MB at 1.2 TB/s (45-65 us per call, seven calls per
// step of the Transformer heads)
__global__ __launch_bounds__(256) void amax_big_kernel(const float* __restrict__ x, long n4, long n, float* __restrict__ out) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float m = 0.f;
    const long stride = (long)gridDim.x * 256;
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    for (; e + 3 * stride < n4; e += 4 * stride) {
        const float4 a = x4[e], b = x4[e + stride], c = x4[e + 2 * stride], d = x4[e + 3 * stride];
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                           fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))),
                           fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
    }
    for (; e < n4; e += stride) {
        const float4 a = x4[e];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
    }
    for (long t = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; t < n; t += stride) m = fmaxf(m, fabsf(x[t]));
    amax_publish_block(out, m);
}

// Nine elements -- one (output channel, input channel) pair -- per task: task q = (ks * No + o) * 16 + il reads the 36
// contiguous bytes of its 3 x 3 filter and writes its 18 halves (a wave's stores of one tap and plane are 128 contiguous bytes).
// One element per trip with three divisions each made the per-step pack of the seven conv weights cost 50 us.
__device__ __forceinline__ void pack_sf16_nine(const float* __restrict__ w, int Cout, int Cin, int dgrad, float sw, long q,
                                               _Float16* __restrict__ wp) {
    const int No = dgrad ? Cin : Cout;
    const int il = (int)(q & 15);
    const long t = q >> 4;
    const int o = (int)(t % No), ks = (int)(t / No);
    const int i = ks * 16 + il;
    const float* src = w + (dgrad ? ((long)i * Cin + o) : ((long)o * Cin + i)) * 9;
    float r[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = src[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = k / 3, dx = k % 3;
        const float v = r[dgrad ? 8 - k : k] * sw;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const long base = ((long)(ks * 3 + dy) * 2) * 3 * No * 16;
        wp[base + ((long)dx * No + o) * 16 + il] = hi;
        wp[base + 3L * No * 16 + ((long)dx * No + o) * 16 + il] = lo;
    }
}

// mode 0 / 1: the forward / dgrad layout into wp; mode 2: BOTH, the forward pack followed by the dgrad pack (18*Cin*Cout
// halfs each) -- one launch per weight and optimiser step
__global__ __launch_bounds__(256) void pack_sf16_kernel(const float* __restrict__ w, int Cout, int Cin, int mode,
                                                        float* __restrict__ wscale, _Float16* __restrict__ wp) {
    const float sw = sf_scale_of(amax_read(wscale));
    if (blockIdx.x == 0 && threadIdx.x == 0) wscale[SED_AMAX_SLOTS] = sw;
    const long total = 9L * Cout * Cin;
    const long pairs = (long)Cout * Cin;
    const long n = mode == 2 ? 2 * pairs : pairs;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        if (mode == 2) {
            if (e < pairs) pack_sf16_nine(w, Cout, Cin, 0, sw, e, wp);
            else pack_sf16_nine(w, Cout, Cin, 1, sw, e - pairs, wp + 2 * total);
        } else {
            pack_sf16_nine(w, Cout, Cin, mode, sw, e, wp);
        }
    }
}

// ---- all conv weights of a model in TWO launches (one amax pass, one pack pass) instead of two per weight: 14 -> 2 tiny
// launches per optimiser step, each of which costs the stream its ~6 us dependent-launch gap on top of its run time
constexpr int SF_MULTI_MAX = 16;
struct SfMultiP {
    const float* w[SF_MULTI_MAX];
    float* wscale[SF_MULTI_MAX];
    _Float16* wp[SF_MULTI_MAX];
    int cout[SF_MULTI_MAX], cin[SF_MULTI_MAX], mode[SF_MULTI_MAX];
};

__global__ __launch_bounds__(256) void amax_multi_kernel(SfMultiP p) {
    const int t = blockIdx.y;
    const float* __restrict__ x = p.w[t];
    const long n = 9L * p.cout[t] * p.cin[t];
    float m = 0.f;
    const long stride = (long)gridDim.x * 256;
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    if ((reinterpret_cast<unsigned long long>(x) & 15) == 0) {     // 16-byte loads, four in flight (one dependent 4-byte load per
        const float4* x4 = reinterpret_cast<const float4*>(x);    // trip walked the 2.4 M floats of a 512 x 512 weight in 35 us)
        const long n4 = n >> 2;
        for (; e + 3 * stride < n4; e += 4 * stride) {
            const float4 a = x4[e], b = x4[e + stride], c = x4[e + 2 * stride], d = x4[e + 3 * stride];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                               fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))),
                               fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
        }
        for (; e < n4; e += stride) {
            const float4 a = x4[e];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        }
        e = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x;
    }
    for (; e < n; e += stride) m = fmaxf(m, fabsf(x[e]));
    amax_publish_block(p.wscale[t], m);
}

__global__ __launch_bounds__(256) void pack_sf16_multi_kernel(SfMultiP p) {
    const int t = blockIdx.y;
    const float* __restrict__ w = p.w[t];
    float* __restrict__ wscale = p.wscale[t];
    _Float16* __restrict__ wp = p.wp[t];
    const int Cout = p.cout[t], Cin = p.cin[t], mode = p.mode[t];
    const float sw = sf_scale_of(amax_read(wscale));
    if (blockIdx.x == 0 && threadIdx.x == 0) wscale[SED_AMAX_SLOTS] = sw;
    const long total = 9L * Cout * Cin;
    const long pairs = (long)Cout * Cin;
    const long n = mode == 2 ? 2 * pairs : pairs;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        if (mode == 2) {
            if (e < pairs) pack_sf16_nine(w, Cout, Cin, 0, sw, e, wp);
            else pack_sf16_nine(w, Cout, Cin, 1, sw, e - pairs, wp + 2 * total);
        } else {
            pack_sf16_nine(w, Cout, Cin, mode, sw, e, wp);
        }
    }
}

static int sf_log2w(int W) { return W == 64 ? 6 : W == 32 ? 5 : W == 16 ? 4 : 3; }
// tile choice: 256 px x 64 co (MW = 4: 50 KB LDS, <= 160 VGPRs, THREE workgroups per CU) for every layer -- 5 % faster than
// 128 px x 128 co (MW = 2: 66 KB, two per CU) on the >= 128-channel layers although it converts each patch twice as
// often: the third wave per SIMD hides more than the reuse saves (re-measured at the metric's batch size at the end of round 4, where
// MW = 2 would balance 1000 workgroups better over its 512 slots than MW = 4 over 768: 8.65 vs 8.44 ms per step; the run-time switch
// of rounds 2-4 is gone, the kernel template still takes MW).
// (Round 3 also built 192 px x 128 co with SIX waves -- three per SIMD AND the 128-channel reuse, 69 KB, two workgroups per CU;
// parity-green: 323-339 TFLOP/s against 408-420 for MW = 4 and 384-390 for MW = 2 over the six >= 128-channel layers,
// profiles/r03/experiment_tile_192x128_six_waves.txt -- six waves do not spread evenly over four SIMDs and only two barrier
// domains share a CU.  And 256 px x 128 co with EIGHT waves at four waves per SIMD (128 VGPRs: 9-41 spilled registers): the
// dgrad variant (11 spills) +0.7 %, the fused-input variants -10 %, experiment_tile_256x128_eight_waves.txt.  Removed again.)
static int sf_mw(int) { return 4; }

}  // namespace

SED_API int sed_conv3x3_sf16_supported(int H, int W, int Cin, int Cout) {
    return (W == 8 || W == 16 || W == 32 || W == 64) && H >= 1 && Cin >= 16 && Cin % 16 == 0 && Cout >= 64 && Cout % 64 == 0;
}

SED_API long sed_conv_sf16_pack_halfs(int Cin, int Cout) { return 18L * Cin * Cout; }

SED_API long sed_conv_sf16_num_parts(int B, int H, int W, int Cout) {
    if (!(W == 8 || W == 16 || W == 32 || W == 64) || Cout % 64) return 0;
    const int mw = sf_mw(Cout), tr = (64 * mw) >> sf_log2w(W);
    return (long)B * ((H + tr - 1) / tr);
}

SED_API int sed_amax(const float* x, long n, float* amax_out, sed_stream_t stream) {
    if (!x || !amax_out || n <= 0) return SED_EINVAL;
    hipError_t e = sed_amax_clear(amax_out, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    const long nb = (n + 1023) / 1024;               // >= 4 elements per thread; few blocks: one atomic per wave, all on one word
    if (n >= (1L << 20) && (reinterpret_cast<unsigned long long>(x) & 15) == 0) {
        const long nb4 = (n / 4 + 4 * 256 - 1) / (4 * 256);          // 16 elements per thread and trip
        hipLaunchKernelGGL(amax_big_kernel, dim3((unsigned)(nb4 > 2048 ? 2048 : nb4)), dim3(256), 0, (hipStream_t)stream, x, n / 4, n,
                           amax_out);
    } else {
        hipLaunchKernelGGL(amax_kernel, dim3((unsigned)(nb > 256 ? 256 : nb)), dim3(256), 0, (hipStream_t)stream, x, n, amax_out);
    }
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_pack_conv_weights_sf16(const float* w_oihw, int Cout, int Cin, int dgrad, float* wscale, void* wp,
                                       sed_stream_t stream) {
    // dgrad: bit 0 = layout of the transposed convolution; bit 1 = wscale already holds the amax slots of w (skip that
    // pass); bit 2 = BOTH layouts, wp = forward pack followed by the dgrad pack (sed_conv_sf16_pack_halfs(...) halfs each)
    const int both = dgrad & 4, dg = dgrad & 1, have_amax = dgrad & 2;
    if (!w_oihw || !wp || !wscale || Cout <= 0 || Cin <= 0 || (dgrad & ~7) || (both && dg)) return SED_EINVAL;
    if (((both || !dg) && Cin % 16) || ((both || dg) && Cout % 16)) return SED_EINVAL;
    const long total = 9L * Cout * Cin;
    if (!have_amax) {
        int rc = sed_amax(w_oihw, total, wscale, stream);
        if (rc) return rc;
    }
    const long nb = ((both ? 2 : 1) * (total / 9) + 255) / 256;
    hipLaunchKernelGGL(pack_sf16_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                       Cout, Cin, both ? 2 : dg, wscale, (_Float16*)wp);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_pack_conv_weights_sf16_multi(int n, const float* const* w_oihw, const int* Cout, const int* Cin, const int* dgrad,
                                             float* const* wscale, void* const* wp, sed_stream_t stream) {
    if (n <= 0 || n > SF_MULTI_MAX || !w_oihw || !Cout || !Cin || !dgrad || !wscale || !wp) return SED_EINVAL;
    SfMultiP p;
    long most = 0;
    for (int t = 0; t < SF_MULTI_MAX; ++t) {
        const int s = t < n ? t : 0;                   // unused entries repeat the first one (never read: gridDim.y = n)
        const int both = dgrad[s] & 4, dg = dgrad[s] & 1;
        if (!w_oihw[s] || !wscale[s] || !wp[s] || Cout[s] <= 0 || Cin[s] <= 0 || (dgrad[s] & ~5) || (both && dg)) return SED_EINVAL;
        if (((both || !dg) && Cin[s] % 16) || ((both || dg) && Cout[s] % 16)) return SED_EINVAL;
        p.w[t] = w_oihw[s]; p.wscale[t] = wscale[s]; p.wp[t] = (_Float16*)wp[s];
        p.cout[t] = Cout[s]; p.cin[t] = Cin[s]; p.mode[t] = both ? 2 : dg;
        const long tot = (both ? 2 : 1) * 9L * Cout[s] * Cin[s];
        if (t < n && tot > most) most = tot;
        if (t < n) {
            hipError_t e = sed_amax_clear(wscale[s], (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
    }
    const long nb_a = (most / 2 + 1023) / 1024, nb_p = (most / 9 + 255) / 256;
    hipLaunchKernelGGL(amax_multi_kernel, dim3((unsigned)(nb_a > 64 ? 64 : (nb_a < 1 ? 1 : nb_a)), n), dim3(256), 0,
                       (hipStream_t)stream, p);
    hipLaunchKernelGGL(pack_sf16_multi_kernel, dim3((unsigned)(nb_p > 1024 ? 1024 : nb_p), n), dim3(256), 0, (hipStream_t)stream, p);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_conv3x3_sf16_eval_pool_supported(int H, int W, int Cin, int Cout, int ph, int pw) {
    if (!sed_conv3x3_sf16_supported(H, W, Cin, Cout) || sf_mw(Cout) != 4) return 0;
    return ((ph == 2 && pw == 2 && W >= 16 && H >= 2) || (ph == 1 && pw == W && W == 8)) ? 1 : 0;
}

SED_API int sed_conv3x3_sf16_eval_pool(const float* x, const void* wp, const float* wscale, float* out, int B, int H, int W,
                                       int Cin, int Cout, const float* in_scale, const float* in_shift, const float* o_scale,
                                       const float* o_shift, int ph, int pw, const float* x_amax, float* out_amax,
                                       int* err_host, int* err_dev, sed_stream_t stream) {
    if (!x || !wp || !wscale || !out || !x_amax || !o_scale || !o_shift || B <= 0 ||
        !sed_conv3x3_sf16_eval_pool_supported(H, W, Cin, Cout, ph, pw))
        return SED_EINVAL;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return SED_EINVAL;
    if (out_amax) {
        hipError_t e = sed_amax_clear(out_amax, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    Sf16P p;
    p.x = x; p.wp = (const _Float16*)wp; p.wscale = wscale; p.x_amax = x_amax; p.y = out;
    p.in_scale = in_scale; p.in_shift = in_shift; p.partials = nullptr; p.yprev = nullptr;
    p.p_scale = o_scale; p.p_shift = o_shift; p.p_mean = nullptr; p.p_invstd = nullptr;
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout;
    p.logW = sf_log2w(W);
    p.TR = 256 >> p.logW;
    p.ntile = (H + p.TR - 1) / p.TR;
    p.mm = nullptr; p.err_host = err_host; p.err_dev = err_dev;
    p.pool_amax = out_amax; p.ph = ph; p.pw = pw;
    p.x0 = p.w1 = nullptr; p.out_amax = nullptr;
    p.ksplit = 1; p.nfull = 0; p.ws = nullptr; p.tickets = nullptr;
    const long nblk = (long)B * p.ntile * (Cout / 64);
    if (nblk > 0x7fffffffL) return SED_EINVAL;
    if (in_scale) hipLaunchKernelGGL((conv_sf16_kernel<4, true, 3>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_sf16_kernel<4, false, 3>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    SED_LAUNCH_CHECK();
    return 0;
}

namespace {
int conv_sf16_launch(const float* x, const void* wp, const float* wscale, float* y, int B, int H, int W, int Cin,
                     int Cout, const float* in_scale, const float* in_shift, int epi, float* partials,
                     const float* yprev, const float* p_scale, const float* p_shift, const float* p_mean,
                     const float* p_invstd, const float* x_amax, float* minmax, int* err_host, int* err_dev,
                     int flags, float* out_amax, int ksplit, int nfull, float* ws, int* tickets, sed_stream_t stream) {
    if (ksplit < 1 || ksplit > (Cin >> 4) || (ksplit > 1 && (!ws || !tickets)) || nfull < 0 || (nfull & 7)) return SED_EINVAL;
    if (!x || !wp || !wscale || !y || !x_amax || B <= 0 || !sed_conv3x3_sf16_supported(H, W, Cin, Cout) || epi < 0 || epi > 2)
        return SED_EINVAL;
    // flags bit 0: x holds split-f16 pairs already (sed_conv1_act_sf16 format, scaled by x_amax): epi 0 / 1, no input transform
    const bool pre = (flags & 1) != 0;
    if ((flags & ~1) || (pre && (in_scale || sf_mw(Cout) != 4))) return SED_EINVAL;
    if (out_amax) {
        hipError_t e = sed_amax_clear(out_amax, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    if (minmax && epi == 2) return SED_EINVAL;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return SED_EINVAL;
    if (epi >= 1 && !partials) return SED_EINVAL;
    if (epi == 2 && (!yprev || !p_scale || !p_shift || !p_mean || !p_invstd)) return SED_EINVAL;
    if (epi == 2 && in_scale) return SED_EINVAL;        // not instantiated (never needed by the models)
    Sf16P p;
    p.x = x; p.wp = (const _Float16*)wp; p.wscale = wscale; p.x_amax = x_amax; p.y = y;
    p.in_scale = in_scale; p.in_shift = in_shift; p.partials = partials; p.yprev = yprev;
    p.p_scale = p_scale; p.p_shift = p_shift; p.p_mean = p_mean; p.p_invstd = p_invstd;
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout;
    p.logW = sf_log2w(W);
    const int mw = sf_mw(Cout);
    p.TR = (64 * mw) >> p.logW;
    p.ntile = (H + p.TR - 1) / p.TR;
    p.mm = minmax; p.err_host = err_host; p.err_dev = err_dev;
    p.pool_amax = nullptr; p.ph = p.pw = 1;
    p.x0 = p.w1 = nullptr; p.out_amax = out_amax;
    p.ksplit = ksplit; p.nfull = ksplit > 1 ? nfull : 0; p.ws = ws; p.tickets = tickets;
    const long ntiles = (long)B * p.ntile * (Cout / (mw == 2 ? 128 : 64));
    if (ksplit > 1 && nfull >= ntiles) return SED_EINVAL;            // nothing left to split
    const long nblk = ksplit > 1 ? nfull + (ntiles - nfull) * ksplit : ntiles;
    if (nblk > 0x7fffffffL) return SED_EINVAL;
    const dim3 g((unsigned)nblk), blk(256);
    hipStream_t s = (hipStream_t)stream;
    // the small-launch form (deep weight pipeline, two workgroups per CU) for every launch of at most two workgroups per CU,
    // split or not
    // (measured with the form forced on every launch at bs=32: 4-8 % slower than three plain workgroups per CU on the large layers,
    // level on the 1024-workgroup ones)
    if (ksplit > 1 || nblk <= 2 * 256) {
        const bool itk = in_scale != nullptr;
#define SF_LAUNCHK(INTV, EPIV, PREV) hipLaunchKernelGGL((conv_sf16_kernel<4, INTV, EPIV, PREV, true>), g, blk, 0, s, p)
        if (pre) { if (epi == 0) SF_LAUNCHK(false, 0, true); else if (epi == 1) SF_LAUNCHK(false, 1, true); else SF_LAUNCHK(false, 2, true); }
        else if (epi == 0) { if (itk) SF_LAUNCHK(true, 0, false); else SF_LAUNCHK(false, 0, false); }
        else if (epi == 1) { if (itk) SF_LAUNCHK(true, 1, false); else SF_LAUNCHK(false, 1, false); }
        else SF_LAUNCHK(false, 2, false);
#undef SF_LAUNCHK
        SED_LAUNCH_CHECK();
        return 0;
    }
#define SF_LAUNCH(MWV, INTV, EPIV) hipLaunchKernelGGL((conv_sf16_kernel<MWV, INTV, EPIV>), g, blk, 0, s, p)
    const bool it = in_scale != nullptr;
    if (pre) {
        if (epi == 0) hipLaunchKernelGGL((conv_sf16_kernel<4, false, 0, true>), g, blk, 0, s, p);
        else if (epi == 1) hipLaunchKernelGGL((conv_sf16_kernel<4, false, 1, true>), g, blk, 0, s, p);
        else hipLaunchKernelGGL((conv_sf16_kernel<4, false, 2, true>), g, blk, 0, s, p);
    } else {
        if (epi == 0) { if (it) SF_LAUNCH(4, true, 0); else SF_LAUNCH(4, false, 0); }
        else if (epi == 1) { if (it) SF_LAUNCH(4, true, 1); else SF_LAUNCH(4, false, 1); }
        else SF_LAUNCH(4, false, 2);
    }
#undef SF_LAUNCH
    SED_LAUNCH_CHECK();
    return 0;
}
}  // namespace

SED_API int sed_conv3x3_sf16(const float* x, const void* wp, const float* wscale, float* y, int B, int H, int W, int Cin,
                             int Cout, const float* in_scale, const float* in_shift, int epi, float* partials,
                             const float* yprev, const float* p_scale, const float* p_shift, const float* p_mean,
                             const float* p_invstd, const float* x_amax, float* minmax, int* err_host, int* err_dev,
                             int flags, float* out_amax, sed_stream_t stream) {
    return conv_sf16_launch(x, wp, wscale, y, B, H, W, Cin, Cout, in_scale, in_shift, epi, partials, yprev, p_scale, p_shift, p_mean,
                            p_invstd, x_amax, minmax, err_host, err_dev, flags, out_amax, 1, 0, nullptr, nullptr, stream);
}

// Split-K forms.  `ksplit` workgroups share one output tile, each over 1/ksplit of the K-steps, and the last one to arrive adds the
// others' accumulators (fixed order: the same bits whoever is last) and runs the (unchanged) epilogue.  Tiles [0, nfull) of the
// launch stay un-split.
//   small-M (round 5; nfull = 0): a launch with fewer workgroups than the chip has resident slots (4 clips per GPU: the
//     512-channel layers are 128 workgroups of 96 dependent stages on 256 CUs) splits EVERY tile;
//   tail (round 6; nfull = a multiple of the 768 resident slots): at the metric's batch (bs = 32) the 125 x 8 layers are 1024
//     workgroups = 1.33 rounds of the chip; the 256 workgroups of the last round, split three ways, are 768 workgroups of a third
//     of the work each -- a full round again.  Built, parity-green, measured, and NOT selected by the library (see below).
// sed_conv_sf16_split_plan: the split this library would choose -- returns ksplit (1: do not split) and *nfull; ws:
// sed_conv_sf16_splitk_floats(...) floats; tickets: sed_conv_sf16_splitk_tickets(...) ints, ZERO before the first use (every launch
// leaves them zero again).
namespace {
constexpr int SF_SLOTS = 768;        // resident workgroups of conv_sf16_kernel<4, ...>: 256 CUs x 3
}  // namespace
// tail_ks: 0 = the library's rule for tails, 2 .. 8 = split a partial last round that many ways wherever one exists (A/B runs, tests).
// THE LIBRARY'S RULE IS "NEVER" (round 6, profiles/r06/tail_split_ab.txt): at the metric's batch every production shape was
// timed un-split and with its tail split 2 / 3 / 4 ways -- the split form is 0-4 % SLOWER on the 125 x 8 layers it was built for
// (512 -> 512: 0.349 ms un-split, 0.350 / 0.366 / 0.356 split) and 3-30 % slower elsewhere; whole step 8.16 -> 8.24 ms.  The
// dispatcher already fills a partial round well: a workgroup alone on a CU gets close to the MFMA share three get together, so
// the tail of 256 workgroups costs ~0.4 of a round, not 1, and the exchange (64 KB per share through the coherence point + the
// re-read by the last arriver, ~15 us) is more than what is left to win.
SED_API int sed_conv_sf16_split_plan(int B, int H, int W, int Cin, int Cout, int tail_ks, int* nfull) {
    if (nfull) *nfull = 0;
    if (B <= 0 || !sed_conv3x3_sf16_supported(H, W, Cin, Cout)) return 1;
    const int tr = 256 >> sf_log2w(W);
    const long wgs = (long)B * ((H + tr - 1) / tr) * (Cout / 64);
    const int kt = Cin >> 4;
    if (wgs <= 160) {
        // small-M: only launches that leave most CUs EMPTY are split (<= 160 workgroups on 256 CUs; measured at 4 clips per GPU:
        // the 250 x 16 layers -- 252 workgroups -- lose 20-40 % when split, the 125 x 8 layers -- 64-128 workgroups -- gain
        // 15-50 %), up to the 768 resident slots, never below 6 K-steps per share (prologue + epilogue of a workgroup cost about 2)
        int s = 1;
        while (s < 8 && wgs * (s * 2) <= SF_SLOTS && kt / (s * 2) >= 6) s *= 2;
        return s;
    }
    if (tail_ks < 2 || tail_ks > 8 || wgs <= SF_SLOTS) return 1;
    const long full = (wgs / SF_SLOTS) * SF_SLOTS;
    if (full == wgs) return 1;
    int s = tail_ks;
    while (s > 1 && kt / s < 2) --s;             // at least two K-steps per share
    if (s <= 1) return 1;
    if (nfull) *nfull = (int)full;
    return s;
}
// the round-5 name: the small-M recommendation only (1 wherever the launch has more than 160 workgroups)
SED_API int sed_conv_sf16_ksplit(int B, int H, int W, int Cin, int Cout) {
    return sed_conv_sf16_split_plan(B, H, W, Cin, Cout, 0, nullptr);
}
SED_API long sed_conv_sf16_splitk_floats(int B, int H, int W, int Cout, int ksplit, int nfull) {
    const int tr = 256 >> sf_log2w(W);
    const long tiles = (long)B * ((H + tr - 1) / tr) * (Cout / 64) - nfull;
    return tiles > 0 ? tiles * ksplit * 16384L : 0;
}
SED_API long sed_conv_sf16_splitk_tickets(int B, int H, int W, int Cout, int nfull) {
    const int tr = 256 >> sf_log2w(W);
    const long tiles = (long)B * ((H + tr - 1) / tr) * (Cout / 64) - nfull;
    return tiles > 0 ? tiles : 0;
}
SED_API int sed_conv3x3_sf16_splitk(const float* x, const void* wp, const float* wscale, float* y, int B, int H, int W, int Cin,
                                    int Cout, const float* in_scale, const float* in_shift, int epi, float* partials,
                                    const float* yprev, const float* p_scale, const float* p_shift, const float* p_mean,
                                    const float* p_invstd, const float* x_amax, float* minmax, int* err_host, int* err_dev,
                                    int flags, float* out_amax, int ksplit, int nfull, float* ws, int* tickets, sed_stream_t stream) {
    return conv_sf16_launch(x, wp, wscale, y, B, H, W, Cin, Cout, in_scale, in_shift, epi, partials, yprev, p_scale, p_shift, p_mean,
                            p_invstd, x_amax, minmax, err_host, err_dev, flags, out_amax, ksplit, nfull, ws, tickets, stream);
}

// Block 1's dgrad (round 4): g_y1 = conv_transpose(gy, w2) masked by relu'(bn1(y1)) + the BatchNorm-backward sums of bn1, with
// y1 = conv1(x0) RECOMPUTED in the epilogue from the one-channel input (an x0 patch in LDS, 9 fmas per element, the fma
// sequence of sed_conv1_fwd: the same bits) instead of read back: same results as sed_conv3x3_sf16(epi = 2, yprev = y1),
// without y1 ever existing.  Cin (channels of gy) % 16 == 0, Cout = 64.
SED_API int sed_conv3x3_sf16_dgrad_b1(const float* gy, const void* wp, const float* wscale, float* gx, int B, int H, int W,
                                      int Cin, int Cout, float* partials, const float* p_scale, const float* p_shift,
                                      const float* p_mean, const float* p_invstd, const float* x0, const float* w1_oihw,
                                      const float* gy_amax, int* err_host, int* err_dev, int flags, float* out_amax,
                                      sed_stream_t stream) {
    if (!gy || !wp || !wscale || !gx || !partials || !p_scale || !p_shift || !p_mean || !p_invstd || !x0 || !w1_oihw || !gy_amax ||
        B <= 0 || Cout != 64 || !sed_conv3x3_sf16_supported(H, W, Cin, Cout) || sf_mw(Cout) != 4 || (flags & ~1))
        return SED_EINVAL;
    if (out_amax) {
        hipError_t e = sed_amax_clear(out_amax, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    Sf16P p;
    p.x = gy; p.wp = (const _Float16*)wp; p.wscale = wscale; p.x_amax = gy_amax; p.y = gx;
    p.in_scale = nullptr; p.in_shift = nullptr; p.partials = partials; p.yprev = nullptr;
    p.p_scale = p_scale; p.p_shift = p_shift; p.p_mean = p_mean; p.p_invstd = p_invstd;
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout;
    p.logW = sf_log2w(W);
    p.TR = 256 >> p.logW;
    p.ntile = (H + p.TR - 1) / p.TR;
    p.mm = nullptr; p.err_host = err_host; p.err_dev = err_dev;
    p.pool_amax = nullptr; p.ph = p.pw = 1;
    p.x0 = x0; p.w1 = w1_oihw; p.out_amax = out_amax;
    p.ksplit = 1; p.nfull = 0; p.ws = nullptr; p.tickets = nullptr;
    const long nblk = (long)B * p.ntile * (Cout / 64);
    if (nblk > 0x7fffffffL) return SED_EINVAL;
    if (flags & 1) hipLaunchKernelGGL((conv_sf16_kernel<4, false, 4, true>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_sf16_kernel<4, false, 4>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    SED_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient with split-f16 operands:
//     dW[co][ci][ky][kx] = sum over pixels of gy[p][co] * a[p + (ky-1, kx-1)][ci],   a = relu(scale*x + shift) or x.
// GEMM per tap with M = co, N = ci, K = pixels.  Both operands live in memory pixel-major (NHWC), i.e. K-major, while an
// MFMA lane wants 8 consecutive k (pixels) of ONE channel: the tiles are staged as they come ([pixel][16 channels], 32 B
// per pixel and 16-channel plane, hi and lo planes) and read with ds_read_b64_tr_b16, the LDS transpose read -- a 16-lane
// group hands in the four 8-byte pieces of four pixels' 16 channels and lane c gets channel c of the four pixels.
// LDS images: one 64-byte row per pixel for x (its two channel blocks side by side), one 128-byte row for gy (four blocks,
// block index XOR (pixel & 2)): stores and transpose reads are both bank-conflict free (see the kernel).
//
// Workgroup = 64 co x 32 ci x all 9 taps; waves = 2 co halves x 2 k halves (each wave: 32 x 32 x 9 taps = 9 accumulators,
// the k halves are separate partial slices).  A stage = 64 pixels (64/W image rows): their gy rows are staged afresh, the
// x rows go into a ring of RING >= 64/W + 2 image rows (power of two), of which every stage only loads the 64/W new ones;
// rows -1 / H and the halo columns are zero.  A slice = a range of stages of one image or a range of whole images; the
// partial sums [slice][tap][co][ci] (the two k halves of a workgroup added in LDS) are reduced in fp64 by wgrad_sf16_reduce_kernel, which also unscales.
// (Splitting the TAPS over six waves -- 2 co halves x 3 kernel rows, 3 accumulators and 120-160 VGPRs per wave, three to four
// waves per SIMD -- measured 300-320 TFLOP/s against 335-360 for this layout: the gy fragments are then read three times.
// Double-buffering gy and enlarging the ring to 2*64/W + 2 rows so that a stage needs ONE barrier instead of two measured
// +-4 % layer by layer, no net gain (round 3 rebuilt it with the staging moved between the two k-steps: -DWSF_OVERLAP, 0 %).  A 64 co x
// 64 ci variant for the 64-input-channel layers was tried: 256 VGPRs with 10-119 spilled registers and no gain where it did
// not spill.  Since round 3 this kernel serves ALL seven layers, the two 64-input-channel ones included.)
namespace {

typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

struct WSf16P {
    const float* x;            // [B][H][W][K]
    const float* gy;           // [B][H][W][N]
    float* partial;            // [nslices][9][N][K]
    const float* in_scale;
    const float* in_shift;
    const float* g_amax;       // device: amax of gy
    const float* x_amax;       // device: amax of the activation operand (relu(scale*x+shift) when fused)
    int B, H, W, K, N;
    int spi, ips;              // slices per image (>= 1) XOR images per slice (>= 1)
    int stages_per_image;
    int* err_host;             // nullable, host-mapped: set to 1 when an operand is not finite
    int* err_dev;              // nullable, device: same
};

__device__ __forceinline__ half4 sf_tr_read(const unsigned char* p) {
    return __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) short4v*)(p)));
}

// XPRE: the activation operand x is ALREADY split (sed_conv1_act_sf16 format, scaled by x_amax): its staging is a plain copy
// GPRE: likewise for the gradient operand gy (pairs written by the BatchNorm-backward apply kernels, scaled by g_amax)
template <int LOGW, bool INT, bool XPRE = false, bool GPRE = false>
__global__ __launch_bounds__(256, 2) void wgrad_sf16_kernel(WSf16P p) {
    constexpr int W = 1 << LOGW, TRS = 64 >> LOGW, WP = W + 2;
    constexpr int RING = TRS <= 2 ? 4 : (TRS == 4 ? 8 : 16);
    constexpr int GBUF = 1;
    // LDS images (round 3: conflict-free for the STORES too -- the per-channel-block planes of round 2, 128 B (mod 256)
    // apart for the transpose reads, put the 16 lanes of a ds_write_b64 group 2-way (x) / 4-way (gy) on the same banks:
    // 35 % of the LDS cycles were conflicts).  x: one 64-byte row per pixel = its two 16-channel blocks side by side;
    // gy: one 128-byte row per pixel = its four blocks, block index XOR (pixel & 2): a 16-lane store group writes one
    // contiguous row (pair), and a transpose read -- 4 consecutive pixels x 32 B of one block per 16 lanes, the
    // neighbouring block in the other 16 -- still covers all 64 banks exactly once.
    constexpr int XPL = RING * WP * 64;             // x plane (hi); lo follows
    constexpr int GPL = 64 * 128;                   // gy plane (hi); lo follows
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * XPL + 2 * GPL * GBUF];
    unsigned char* const Xs = smem;
    unsigned char* const Gs = smem + 2 * XPL;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wvu & 1, wk = wvu >> 1;
    const int ci_tiles = p.K >> 5;
    const int ntile = ci_tiles * (p.N >> 6);
    const int logical = xcd_remap_sf(blockIdx.x, gridDim.x);
    const int tile = logical % ntile, slice = logical / ntile;
    const int ci0 = (tile % ci_tiles) * 32, co0 = (tile / ci_tiles) * 64;
    int b0, b1, s0, s1;                                  // images [b0, b1), stages [s0, s1) of each
    if (p.ips >= 1) {
        b0 = slice * p.ips; b1 = min(p.B, b0 + p.ips); s0 = 0; s1 = p.stages_per_image;
    } else {
        b0 = slice / p.spi; b1 = b0 + 1;
        const int per = (p.stages_per_image + p.spi - 1) / p.spi;
        s0 = (slice % p.spi) * per; s1 = min(p.stages_per_image, s0 + per);
    }
    const float sg = sf_scale_of(amax_read(p.g_amax)), sa = sf_scale_of(amax_read(p.x_amax));

    // ---- staging maps.  x: item e = tid + 256*i (i < 2): pixel e >> 3 of the 64 new ones, channel quad e & 7 (32 ci);
    //      gy: item e (i < 4): pixel e >> 4, channel quad e & 15 (64 co)
    constexpr int OOB = (int)0x80000000;
    const int xq = tid & 7, gq = tid & 15;
    float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (INT) {
        xsc = *reinterpret_cast<const float4*>(p.in_scale + ci0 + xq * 4);
        xsh = *reinterpret_cast<const float4*>(p.in_shift + ci0 + xq * 4);
        // the power-of-two operand scale rides on the affine: relu(sa*sc*x + sa*sh) == sa*relu(sc*x + sh) bit for bit
        xsc.x *= sa; xsc.y *= sa; xsc.z *= sa; xsc.w *= sa; xsh.x *= sa; xsh.y *= sa; xsh.z *= sa; xsh.w *= sa;
    }
    const unsigned x_img_bytes = (unsigned)p.H * W * p.K * 4u, g_img_bytes = (unsigned)p.H * W * p.N * 4u;
    bool overflow = false;
    int xrr[2], xcc[2], grr[4], gls[4], goff[4];
    float4 xreg[2], greg[4];
    bool xok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pe = (tid + 256 * i) >> 3;
        xrr[i] = pe >> LOGW; xcc[i] = pe & (W - 1);
        xreg[i] = make_float4(0.f, 0.f, 0.f, 0.f); xok[i] = false;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pe = (tid + 256 * i) >> 4;
        grr[i] = pe >> LOGW;
        gls[i] = pe * 128 + (((gq >> 2) ^ (pe & 2)) << 5) + (gq & 3) * 8;
        goff[i] = ((grr[i] * W + (pe & (W - 1))) * p.N + co0 + gq * 4) * 4;
        greg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __amdgpu_buffer_rsrc_t xrs, grs;
#define WSF_IMAGE(BB)                                                                                           \
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (long)(BB) * p.H * W * p.K, 0, (int)x_img_bytes, 0x00020000); \
    grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gy) + (long)(BB) * p.H * W * p.N, 0, (int)g_img_bytes, 0x00020000);
    // x rows [ROW0, ROW0 + TRS) -> registers (rows outside the image read as zero)
#define WSF_XLOAD(ROW0)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                             \
        const int row = (ROW0) + xrr[i];                                                                        \
        xok[i] = (unsigned)row < (unsigned)p.H;                                                                 \
        const int off = xok[i] ? ((row * W + xcc[i]) * p.K + ci0 + xq * 4) * 4 : OOB;                           \
        xreg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));            \
    }
#define WSF_XSTORE(ROW0)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                             \
        if (XPRE) {                                /* {h01, l01, h23, l23}; rows outside the image loaded as 0 */ \
            const uint4 u = __builtin_bit_cast(uint4, xreg[i]);                                                 \
            const int slot = ((ROW0) + xrr[i] + 1) & (RING - 1);                                                \
            const int o = (slot * WP + xcc[i] + 1) * 64 + xq * 8;                                               \
            *reinterpret_cast<uint2*>(Xs + o) = make_uint2(u.x, u.z);                                           \
            *reinterpret_cast<uint2*>(Xs + XPL + o) = make_uint2(u.y, u.w);                                     \
            continue;                                                                                           \
        }                                                                                                       \
        float4 v = xreg[i];                                                                                     \
        if (INT) {                                                                                              \
            v.x = bn_relu(v.x, xsc.x, xsh.x); v.y = bn_relu(v.y, xsc.y, xsh.y);                                 \
            v.z = bn_relu(v.z, xsc.z, xsh.z); v.w = bn_relu(v.w, xsc.w, xsh.w);                                 \
        } else {                                                                                                \
            v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa;                                                         \
        }                                                                                                       \
        overflow |= !((fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w)) < 3.0e5f);  /* inf and NaN propagate through the adds */ \
        unsigned h01, l01, h23, l23;                                                                            \
        sf_split2(v.x, v.y, h01, l01);                                                                          \
        sf_split2(v.z, v.w, h23, l23);                                                                          \
        if (INT && !xok[i]) { h01 = h23 = l01 = l23 = 0u; }    /* rows outside the image: the ring slot gets zeros */ \
        const int slot = ((ROW0) + xrr[i] + 1) & (RING - 1);                                                    \
        const int o = (slot * WP + xcc[i] + 1) * 64 + xq * 8;                                                   \
        *reinterpret_cast<uint2*>(Xs + o) = make_uint2(h01, h23);                                               \
        *reinterpret_cast<uint2*>(Xs + XPL + o) = make_uint2(l01, l23);                                         \
    }
#define WSF_GLOAD(H0)                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
        greg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, goff[i], (H0) * W * p.N * 4, 0));
#define WSF_GSTORE(GB)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                             \
        if (GPRE) {                                                                                             \
            const uint4 u = __builtin_bit_cast(uint4, greg[i]);                                                 \
            *reinterpret_cast<uint2*>(Gs + (GB) + gls[i]) = make_uint2(u.x, u.z);                               \
            *reinterpret_cast<uint2*>(Gs + (GB) + GPL + gls[i]) = make_uint2(u.y, u.w);                         \
            continue;                                                                                           \
        }                                                                                                       \
        float4 v = greg[i];                                                                                     \
        v.x *= sg; v.y *= sg; v.z *= sg; v.w *= sg;                                                             \
        overflow |= !((fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w)) < 3.0e5f);                         \
        unsigned h01, l01, h23, l23;                                                                            \
        sf_split2(v.x, v.y, h01, l01);                                                                          \
        sf_split2(v.z, v.w, h23, l23);                                                                          \
        *reinterpret_cast<uint2*>(Gs + (GB) + gls[i]) = make_uint2(h01, h23);                                   \
        *reinterpret_cast<uint2*>(Gs + (GB) + GPL + gls[i]) = make_uint2(l01, l23);                             \
    }

    // halo columns of every ring row, both x planes: zero once
    for (int i = tid; i < RING * 2 * 2 * 4; i += 256) {
        const int piece = i & 3, pl = (i >> 2) & 1, side = (i >> 3) & 1, slot = i >> 4;
        *reinterpret_cast<float4*>(Xs + pl * XPL + (slot * WP + (side ? W + 1 : 0)) * 64 + piece * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // ---- fragment addressing (lane-static parts)
    const int g16 = lane >> 4, cbl = g16 & 1, khalf = g16 >> 1, r4 = (lane >> 2) & 3, ch = lane & 3;
    const int a_base = (8 * khalf + r4) * 128 + (((2 * wc + cbl) ^ (r4 & 2)) << 5) + ch * 8;   // + ks*2048 + rd*512 (+ GPL: lo)
    const int pin = 8 * khalf + r4;                                                // pixel within the k-step (rd adds 4)
    const int b_lane = cbl * 32 + ch * 8;

    floatx16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    for (int b = b0; b < b1; ++b) {
        WSF_IMAGE(b)
        const int hs0 = s0 * TRS;                          // first output row of this slice in the image
        __syncthreads();                                   // previous image fully consumed
        // prologue: rows hs0-1 .. hs0 (+ what comes with them), then the first stage's new rows and its gy rows
        WSF_XLOAD(hs0 - 1) WSF_XSTORE(hs0 - 1)
        if (TRS == 1) { WSF_XLOAD(hs0) WSF_XSTORE(hs0) }
        WSF_XLOAD(hs0 + 1) WSF_GLOAD(hs0)
        WSF_XSTORE(hs0 + 1) WSF_GSTORE(0)
        __syncthreads();
        for (int s = s0; s < s1; ++s) {
            const int h0 = s * TRS;
            const bool more = s + 1 < s1;
            if (more) { WSF_XLOAD(h0 + TRS + 1) WSF_GLOAD(h0 + TRS) }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ks = 2 * wk + kk;                // wave-uniform (wk scalar)
                const unsigned char* const Gcur = Gs;
                half8 ah, al;
                {
                    const unsigned char* ap = Gcur + a_base + ks * 2048;
                    const half4 h0v = sf_tr_read(ap), h1v = sf_tr_read(ap + 512);
                    const half4 l0v = sf_tr_read(ap + GPL), l1v = sf_tr_read(ap + GPL + 512);
                    ah = half8{h0v[0], h0v[1], h0v[2], h0v[3], h1v[0], h1v[1], h1v[2], h1v[3]};
                    al = half8{l0v[0], l0v[1], l0v[2], l0v[3], l1v[0], l1v[1], l1v[2], l1v[3]};
                }
                const int pix = ks * 16 + pin;             // pixel of the stage (rd = 0)
                const int row_s = pix >> LOGW, col = pix & (W - 1);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int slot = (h0 + row_s + dy) & (RING - 1);
                    const unsigned char* bp = Xs + b_lane + (slot * WP + col) * 64;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const half4 h0v = sf_tr_read(bp + dx * 64), h1v = sf_tr_read(bp + dx * 64 + 256);
                        const half4 l0v = sf_tr_read(bp + XPL + dx * 64), l1v = sf_tr_read(bp + XPL + dx * 64 + 256);
                        const half8 bh = {h0v[0], h0v[1], h0v[2], h0v[3], h1v[0], h1v[1], h1v[2], h1v[3]};
                        const half8 bl = {l0v[0], l0v[1], l0v[2], l0v[3], l1v[0], l1v[1], l1v[2], l1v[3]};
                        const int tp = dy * 3 + dx;
                        acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[tp], 0, 0, 0);
                        acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[tp], 0, 0, 0);
                        acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[tp], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (more) {
                WSF_XSTORE(h0 + TRS + 1) WSF_GSTORE(0)
                __syncthreads();
            }
        }
    }
#undef WSF_IMAGE
#undef WSF_XLOAD
#undef WSF_XSTORE
#undef WSF_GLOAD
#undef WSF_GSTORE

    if (overflow) {
        if (p.err_host) __hip_atomic_store(p.err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (p.err_dev) __hip_atomic_store(p.err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- the two k halves of the workgroup meet in LDS (round 4: the staging buffers are free now; three accumulators per
    // round = 24 KB), so a slice leaves ONE partial [tap][co][ci] instead of two: half the partial-sum bytes written here and
    // read by the reduce kernel (300 -> 150 MB per launch on 512 -> 512 at batch 256)
    float* const red = reinterpret_cast<float*>(smem);
    static_assert(sizeof(smem) >= 2 * 3 * 16 * 64 * sizeof(float), "the k-half exchange needs 24 KB of LDS");
    __syncthreads();
#pragma unroll
    for (int g3 = 0; g3 < 3; ++g3) {
        if (wk == 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wc * 3 + a) * 16 + r) * 64 + lane] = acc[3 * g3 + a][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[3 * g3 + a][r] += red[((wc * 3 + a) * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (wk != 0) return;
    float* out = p.partial + ((long)slice * 9) * p.N * p.K;
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[((long)a * p.N + co) * p.K + ci0 + (lane & 31)] = acc[a][r];
        }
}

// sum the slices in fp64, unscale, scatter to OIHW.  Two cuts of the same sum (fixed orders: deterministic), chosen by the slice
// count -- the partials are always ~38 MB (512 workgroups x 74 KB), what changes is their shape:
//   MANY slices x few elements (the <= 128-channel layers: 32 .. 256 slices of 37 K .. 295 K elements): 64 elements x G = 4 part
//     groups per block, group g sums parts g, g + 4, ... with eight independent loads in flight, the groups meet in LDS: 7.7-8.5 us
//     for 38 MB.  (G = 16 / 1024 threads was built in round 6 and is never faster; kept as a test-hook variant.  One thread per
//     element walking all parts in a dependent chain took 315 us for the 1000 partials of the 64 -> 64 layer at batch 32.)
//   FEW slices x many elements (>= 256 input channels: 4 .. 16 slices of 0.6 .. 2.4 M elements): a block owns one output channel
//     x 64 input channels x all 9 taps, every thread sums the parts of two or three taps (all loads independent), the 576 sums
//     meet in LDS and leave as ONE contiguous 2304-byte run of the OIHW tensor -- the element-per-thread form wrote 4 bytes every
//     36 bytes: 2.4 M write transactions for the 512 -> 512 layer.
template <int G>
__global__ __launch_bounds__(64 * G) void wgrad_sf16_reduce_kernel(const float* __restrict__ partial, int nparts, int N, int K,
                                                                   const float* __restrict__ g_amax,
                                                                   const float* __restrict__ x_amax,
                                                                   float* __restrict__ dw) {
    __shared__ double red[64 * G];
    const long nk = (long)9 * N * K;                       // a multiple of 64
    const long e = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int grp = threadIdx.x >> 6;
    const double inv = 1.0 / ((double)sf_scale_of(amax_read(g_amax)) * (double)sf_scale_of(amax_read(x_amax)));
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int q = grp;
    // eight independent loads per trip (all requested before the first add: one memory round trip per eight parts), four fp64 chains
    for (; q + 7 * G < nparts; q += 8 * G) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = partial[(long)(q + j * G) * nk + e];
        s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
        s0 += (double)v[4]; s1 += (double)v[5]; s2 += (double)v[6]; s3 += (double)v[7];
    }
    for (; q + 3 * G < nparts; q += 4 * G) {
        s0 += (double)partial[(long)q * nk + e];
        s1 += (double)partial[(long)(q + G) * nk + e];
        s2 += (double)partial[(long)(q + 2 * G) * nk + e];
        s3 += (double)partial[(long)(q + 3 * G) * nk + e];
    }
    for (; q < nparts; q += G) s0 += (double)partial[(long)q * nk + e];
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < G; g += 4)
            s += (red[threadIdx.x + 64 * g] + red[threadIdx.x + 64 * (g + 1)]) + (red[threadIdx.x + 64 * (g + 2)] + red[threadIdx.x + 64 * (g + 3)]);
        const int ci = (int)(e % K);
        const long t = e / K;
        const int co = (int)(t % N), tap = (int)(t / N);
        dw[((long)co * K + ci) * 9 + tap] = (float)(s * inv);
    }
}

__global__ __launch_bounds__(256) void wgrad_sf16_reduce_rows_kernel(const float* __restrict__ partial, int nparts, int N, int K,
                                                                     const float* __restrict__ g_amax,
                                                                     const float* __restrict__ x_amax,
                                                                     float* __restrict__ dw) {
    __shared__ float outs[64 * 9];
    const int kb = K >> 6;                                  // 64-channel chunks of the input channels (K % 64 == 0 here)
    const int co = blockIdx.x / kb, ci0 = (blockIdx.x % kb) << 6;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long nk = (long)9 * N * K;
    const double inv = 1.0 / ((double)sf_scale_of(amax_read(g_amax)) * (double)sf_scale_of(amax_read(x_amax)));
    for (int tap = g; tap < 9; tap += 4) {                  // group 0: taps 0, 4, 8; groups 1 .. 3: two taps each
        const float* src = partial + ((long)tap * N + co) * K + ci0 + lane;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int q = 0;
        for (; q + 3 < nparts; q += 4) {
            s0 += (double)src[(long)q * nk];
            s1 += (double)src[(long)(q + 1) * nk];
            s2 += (double)src[(long)(q + 2) * nk];
            s3 += (double)src[(long)(q + 3) * nk];
        }
        for (; q < nparts; ++q) s0 += (double)src[(long)q * nk];
        outs[lane * 9 + tap] = (float)(((s0 + s1) + (s2 + s3)) * inv);
    }
    __syncthreads();
    float* dst = dw + ((long)co * K + ci0) * 9;             // 64 input channels x 9 taps of this output channel: contiguous in OIHW
    for (int i = threadIdx.x; i < 576; i += 256) dst[i] = outs[i];
}

// variant 0: the library's choice by slice count; 1 .. 3: forced (sed_test_wgrad_sf16_reduce)
static void wsf_reduce_launch(const float* partial, int ns, int Cout, int Cin, const float* gy_amax, const float* x_amax,
                              float* dw_oihw, int variant, hipStream_t s) {
    const long nk = 9L * Cin * Cout;
    // measured per cut on the production shapes (tools/wgrad_reduce_bench.py, profiles/r06/wgrad_reduce_bench.txt; 38 MB of partials):
    // rows 8-10 us where the 4-group form takes 11 / 19 / 31 us (256 -> 256, 256 -> 512, 512 -> 512), 4 groups 7.7-8.5 us on the
    // <= 128-channel layers where rows takes 11-18 us; 16 groups never wins (8.7 ... 117 us)
    if (variant == 0) variant = (nk >= 9L * 256 * 256 && ns <= 64 && Cin % 64 == 0) ? 3 : 1;
    if (variant == 3)
        hipLaunchKernelGGL(wgrad_sf16_reduce_rows_kernel, dim3((unsigned)((long)Cout * (Cin / 64))), dim3(256), 0, s, partial, ns,
                           Cout, Cin, gy_amax, x_amax, dw_oihw);
    else if (variant == 2)
        hipLaunchKernelGGL(wgrad_sf16_reduce_kernel<16>, dim3((unsigned)(nk / 64)), dim3(1024), 0, s, partial, ns, Cout, Cin,
                           gy_amax, x_amax, dw_oihw);
    else
        hipLaunchKernelGGL(wgrad_sf16_reduce_kernel<4>, dim3((unsigned)(nk / 64)), dim3(256), 0, s, partial, ns, Cout, Cin,
                           gy_amax, x_amax, dw_oihw);
}

static void wsf_slicing(int B, int H, int W, int Cin, int Cout, int* spi, int* ips, int* spimg, long* nslices) {
    const int trs = 64 / W;
    const int g = (H + trs - 1) / trs;               // stages (64 pixels) per image
    const long tiles = (long)(Cin / 32) * (Cout / 64);
    *spimg = g;
    // A slice is 1/s of an image (s = 1 .. g/8: at least 8 stages) or i whole images.  512 workgroups are resident at a time (two
    // per CU), so a launch costs  rounds(workgroups / 512) x (stages per slice + c)  with c ~ 4 stages for a workgroup's prologue
    // and its 74 KB partial write: take the cheapest cut with at most 256 stages (16384 pixels) per fp32 accumulation chain
    // (7e-7 relative at 125 stages, tests/test_gpu_sf16.py; beyond a slice the sums continue in fp64 in the reduce kernel).
    // (Until late round 4: ~2048 workgroups, at least 64 stages, rounded DOWN -- 384 / 768 / 960 workgroups at the metric's batch
    // size, i.e. half-empty last rounds, and 256 for 128 -> 256 @ 250 x 16.)
    long best_cost = -1, best_blk = 0;
    int best_s = 1, best_i = 0;
    auto consider = [&](int s, int i) {
        const long nsl = s ? (long)B * s : (B + i - 1) / i;
        const long stages = s ? (g + s - 1) / s : (long)i * g;
        if (stages > 256 && !(s && s == (g / 8 > 0 ? g / 8 : 1))) return;      // too long a chain (unless nothing shorter exists)
        const long blk = nsl * tiles;
        const long cost = ((blk + 511) / 512) * (stages + 4);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && blk < best_blk)) {
            best_cost = cost; best_blk = blk; best_s = s; best_i = i;
        }
    };
    const int smax = g / 8 > 0 ? g / 8 : 1;
    for (int s = 1; s <= smax; ++s) consider(s, 0);
    for (int i = 2; i <= B; ++i) consider(0, i);
    *spi = best_s; *ips = best_i;
    *nslices = best_s ? (long)B * best_s : (B + best_i - 1) / best_i;
}

}  // namespace

SED_API int sed_wgrad_sf16_supported(int H, int W, int Cin, int Cout) {
    return (W == 8 || W == 16 || W == 32 || W == 64) && H >= 1 && Cin >= 32 && Cin % 32 == 0 && Cout >= 64 && Cout % 64 == 0;
}

SED_API long sed_wgrad_sf16_partial_floats(int B, int H, int W, int Cin, int Cout) {
    if (!sed_wgrad_sf16_supported(H, W, Cin, Cout) || B <= 0) return 0;
    int spi, ips, g; long ns;
    wsf_slicing(B, H, W, Cin, Cout, &spi, &ips, &g, &ns);
    return ns * 9 * Cin * Cout;
}

SED_API int sed_conv3x3_wgrad_sf16(const float* x, const float* gy, float* dw_oihw, float* partial, int B, int H, int W,
                                   int Cin, int Cout, const float* in_scale, const float* in_shift, const float* gy_amax,
                                   const float* x_amax, int* err_host, int* err_dev, int flags, sed_stream_t stream) {
    if (!x || !gy || !dw_oihw || !partial || !gy_amax || !x_amax || B <= 0 || !sed_wgrad_sf16_supported(H, W, Cin, Cout))
        return SED_EINVAL;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return SED_EINVAL;
    // flags bit 0: x holds split-f16 pairs already (sed_conv1_act_sf16 format, scaled by x_amax); no input transform then;
    // bit 1: gy holds such pairs (scaled by gy_amax)
    const bool xpre = (flags & 1) != 0, gpre = (flags & 2) != 0;
    if ((flags & ~3) || (xpre && in_scale)) return SED_EINVAL;
    WSf16P p;
    p.x = x; p.gy = gy; p.partial = partial; p.in_scale = in_scale; p.in_shift = in_shift; p.g_amax = gy_amax;
    p.x_amax = x_amax;
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout; p.err_host = err_host; p.err_dev = err_dev;
    long ns;
    wsf_slicing(B, H, W, Cin, Cout, &p.spi, &p.ips, &p.stages_per_image, &ns);
    const long nblk = ns * (Cin / 32) * (Cout / 64);
    if (nblk > 0x7fffffffL) return SED_EINVAL;
    const dim3 g((unsigned)nblk), blk(256);
    hipStream_t s = (hipStream_t)stream;
    const bool it = in_scale != nullptr;
#define WSF_LAUNCH(LW)                                                                                          \
    if (gpre) {                                                                                                 \
        if (xpre) hipLaunchKernelGGL((wgrad_sf16_kernel<LW, false, true, true>), g, blk, 0, s, p);              \
        else if (it) hipLaunchKernelGGL((wgrad_sf16_kernel<LW, true, false, true>), g, blk, 0, s, p);           \
        else hipLaunchKernelGGL((wgrad_sf16_kernel<LW, false, false, true>), g, blk, 0, s, p);                  \
    } else if (xpre) hipLaunchKernelGGL((wgrad_sf16_kernel<LW, false, true>), g, blk, 0, s, p);                 \
    else if (it) hipLaunchKernelGGL((wgrad_sf16_kernel<LW, true>), g, blk, 0, s, p);                            \
    else hipLaunchKernelGGL((wgrad_sf16_kernel<LW, false>), g, blk, 0, s, p);
    if (W == 64) { WSF_LAUNCH(6) } else if (W == 32) { WSF_LAUNCH(5) } else if (W == 16) { WSF_LAUNCH(4) } else { WSF_LAUNCH(3) }
#undef WSF_LAUNCH
    SED_LAUNCH_CHECK();
    wsf_reduce_launch(partial, (int)ns, Cout, Cin, gy_amax, x_amax, dw_oihw, 0, s);
    SED_LAUNCH_CHECK();
    return 0;
}

#include "sed_hip_test.h"
SED_API int sed_test_wgrad_sf16_reduce(const float* partial, int nparts, int Cout, int Cin, const float* gy_amax, const float* x_amax,
                                       float* dw_oihw, int variant, sed_stream_t stream) {
    if (!partial || !gy_amax || !x_amax || !dw_oihw || nparts <= 0 || Cin % 32 || Cout % 64 || variant < 0 || variant > 3 ||
        (variant == 3 && Cin % 64))
        return SED_EINVAL;
    wsf_reduce_launch(partial, nparts, Cout, Cin, gy_amax, x_amax, dw_oihw, variant, (hipStream_t)stream);
    SED_LAUNCH_CHECK();
    return 0;
}

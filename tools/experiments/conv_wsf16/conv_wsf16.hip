// 3x3 convolution (forward and dgrad) as a FUSED 2-D Winograd F(2x2,3x3) on the f16 MFMA pipe with SPLIT operands:
//
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          16 multiplies per 2x2 output tile and (ci, co) pair instead of 36
//     each Winograd-domain product as hi*hi + hi*lo + lo*hi of (hi, lo) f16 pairs (csrc/conv_sf16.hip)
//
// i.e. 2.25x fewer v_mfma_f32_32x32x16_f16 than the direct split-f16 kernel for the same convolution -- the round-4 review's
// lever against the 3x emulation tax (reference work replaced: nn.Conv2d 3x3 of ConvBlock, pytorch/models.py:77-85, :102-103).
//
// Why the shape below (DESIGN.md section 5 has the budget): a Winograd-domain GEMM re-uses nothing ACROSS the 16 positions, so
// per MFMA it needs twice the LDS fragment bytes of the direct kernel, and a transformed input tile is 4x the bytes of the raw
// patch it comes from.  Staging the TRANSFORMED tile in LDS (16 positions x hi/lo) would cost 64 KB of ds_write per K-step and
// 128 KB of LDS for one un-buffered stage.  Instead:
//   * a workgroup owns 128 tiles (512 output pixels: OR = 512/W rows x W columns of one image) x 32 output channels; each of its
//     four waves owns 32 tiles x 32 channels x ALL 16 positions = 16 accumulators (256 registers, the AGPR half of the file);
//   * the RAW fp32 patch of a K-step (16 input channels; (OR+2) x (W+2) pixels; 42-46 KB) is staged once (buffer loads ->
//     optional relu(scale*x+shift) -> operand scale -> ds_write_b128), even and odd columns in separate planes so that the tiles
//     of a wave are consecutive 64-byte rows; double-buffered;
//   * every lane reads the 4x4 patch of ITS tile (8 channels: its half of the K-step), forms V = B^T d B in registers, splits
//     it (3 VALU per value pair) and feeds the MFMAs directly: the transformed tile never touches LDS, and one raw pixel is
//     read by the four tiles that share it instead of being transformed and stored four times;
//   * the pre-transformed, pre-split weights U[K/16][16 pos][hi,lo][Cout][16] stream by LDS-DMA (32 KB per K-step and
//     workgroup, shared by its 128 tiles), double-buffered; one barrier per K-step;
//   * the output transform is lane-local (a lane holds all 16 positions of its (tile, channel) pairs in the same register index).
// Operand scales: the input's power of two is TWO binades lower than the direct kernel's (|V| <= 4 max|d| must stay below 2^15);
// the weights' comes from the amax of the TRANSFORMED weights.
// EXPERIMENT (round 5; tools/experiments/conv_wsf16/README.md has the measurements and the verdict): built and run stand-alone by
// run.py, NOT part of libsed_hip.so.
#include "common.h"
#include <stdlib.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

struct WsfP {
    const float* x;            // [B][H][W][K]
    const _Float16* up;        // [K/16][16 pos][2 planes][N][16]
    const float* wscale;       // [SED_AMAX_SLOTS + 1]: amax slots of the transformed weights, then their scale
    const float* x_amax;       // device: amax of the operand as the transform sees it (relu(scale*x+shift) when fused)
    float* y;                  // [B][H][W][N]
    const float* in_scale;
    const float* in_shift;
    int B, H, W, K, N;
    int logW, OR, nrb, PW, ntb;
    int* err_host;
    int* err_dev;
    int abl;                   // TEMP experiment switch (WSF_ABL): 1 no DMA after the prologue, 2 no split, 4 no MFMA, 8 no raw reads
};

constexpr int WSF_RAWROWS = 720;                 // >= (OR + 2) * 2 * PW over W in {8, 16, 32, 64}
constexpr int WSF_RAWBYTES = WSF_RAWROWS * 64;
constexpr int WSF_USTAGE = 16 * 2 * 32 * 32;     // 16 positions x (hi, lo) x 32 channels x 32 bytes
constexpr int WSF_NI = 10;                       // staging items (16 bytes) per thread and K-step

// raw-patch rows: 64 bytes (16 channels fp32); 16-byte chunk index XOR ((row >> 2) & 3): the 16 lanes of a ds_read_b128 group
// read rows that are distinct mod 16 (consecutive tiles; the plane pitch PW is chosen per W so that this also holds where a
// 32-tile block spans several tile rows), i.e. 16 different 16-byte slots of the 256-byte bank line
__device__ __forceinline__ int wsf_raw(int R, int chunk) { return R * 64 + ((chunk ^ ((R >> 2) & 3)) << 4); }
__device__ __forceinline__ int wsf_usw(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4); }

// (hi, lo) f16 pairs of TWO fp32 values times a power of two s: hi = f16(s v), lo = f16(s v - hi) -- four mixed-precision fmas (the
// un-scaled form of common.h needs three, but here the raw patch arrives by LDS-DMA and nobody has multiplied it yet)
__device__ __forceinline__ void wsf_split2s(float a, float b, float s, unsigned& hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(a), "v"(b), "v"(s));
}

template <bool INT, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv_wsf16_kernel(WsfP p) {
    // (+ 4 KB behind the buffers: where the staging stores of pixels outside the image go -- branch-free stores)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WSF_RAWBYTES + 2 * WSF_USTAGE + 4096];
    unsigned char* const Raw = smem;
    unsigned char* const Us = smem + 2 * WSF_RAWBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = p.W, logW = p.logW, OR = p.OR, PW = p.PW;
    const int KT = p.K >> 4;
    // ---- workgroup -> (tile block, channel block), XCD-aware: an XCD keeps to a FEW channel blocks (their U slabs stay in its
    // L2) and walks the tile blocks in the same order as the other XCDs (the raw patches are shared through the MALL)
    const int nb = p.N >> 5;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    int cb, tb;
    if (nb >= 8) { const int cpx = nb >> 3; cb = xcd * cpx + j % cpx; tb = j / cpx; }
    else { const int share = 8 / nb; cb = xcd % nb; tb = j * share + xcd / nb; }
    if (tb >= p.ntb) return;
    const int n0 = cb * 32;
    const int b = tb / p.nrb, h0 = (tb % p.nrb) * OR;

    const float sa = sed_sf_scale_of(amax_read(p.x_amax)) * 0.25f;
    const float inv = 1.0f / (sa * p.wscale[SED_AMAX_SLOTS]);

    // ---- raw-patch staging: item e = tid + 256 i -> patch pixel e >> 2 (row prow, image column c), channel quad e & 3
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (long)b * p.H * W * p.K, 0, (int)((unsigned)p.H * W * p.K * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(INT ? p.in_scale : p.x), 0, p.K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(INT ? p.in_shift : p.x), 0, p.K * 4, 0x00020000);
    const int q4 = tid & 3;
    bool val[WSF_NI], sok[WSF_NI];
    int lso[WSF_NI], aoff[WSF_NI];
    float4 areg[WSF_NI];
#pragma unroll
    for (int i = 0; i < WSF_NI; ++i) {
        const int pe = (tid + 256 * i) >> 2;
        const int prow = pe >> logW, c = pe & (W - 1);
        const int h = h0 - 1 + prow;
        val[i] = prow < OR + 2;
        sok[i] = val[i] && (unsigned)h < (unsigned)p.H;
        const int pc = c + 1;
        lso[i] = wsf_raw((prow * 2 + (pc & 1)) * PW + (pc >> 1), q4);
        aoff[i] = sok[i] ? ((h * W + c) * p.K + q4 * 4) * 4 : OOB;
        areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool overflow = false;

    auto aload = [&](int ks) {
        const int k_off = ks * 64;
        if (INT) {
            sc4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srs, q4 * 16, k_off, 0));
            sh4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hrs, q4 * 16, k_off, 0));
            sc4.x *= sa; sc4.y *= sa; sc4.z *= sa; sc4.w *= sa; sh4.x *= sa; sh4.y *= sa; sh4.z *= sa; sh4.w *= sa;
        }
#pragma unroll
        for (int i = 0; i < WSF_NI; ++i) areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, aoff[i], k_off, 0));
    };
    const int dump = 2 * WSF_RAWBYTES + 2 * WSF_USTAGE + tid * 16;
    auto astore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WSF_NI; ++i) {
            // rows outside the image were zeroed once (both buffers): their items go to the dump slot instead (no branch)
            float4 v = areg[i];
            if (INT) {
                v.x = bn_relu(v.x, sc4.x, sh4.x); v.y = bn_relu(v.y, sc4.y, sh4.y);
                v.z = bn_relu(v.z, sc4.z, sh4.z); v.w = bn_relu(v.w, sc4.w, sh4.w);
            } else {
                v.x *= sa; v.y *= sa; v.z *= sa; v.w *= sa;
            }
            overflow |= sok[i] && !((fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w)) < 3.0e5f);
            *reinterpret_cast<float4*>(smem + (sok[i] ? buf * WSF_RAWBYTES + lso[i] : dump)) = v;
        }
    };

    // ---- U by LDS-DMA: 32 slabs of 1 KB (position, plane) per stage, 8 per wave; lane i lands at slab + 16 i = row i >> 1,
    // chunk slot i & 1, and fetches the chunk that belongs there
    const int brow = lane >> 1;
    const int boff = (n0 + brow) * 32 + ((((lane & 1) ^ ((brow >> 3) & 1))) << 4);
    const unsigned us_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Us);
    const long slab_stride = (long)p.N * 32;                     // bytes between (pos, plane) slabs in global memory
    auto udma = [&](int ks, int st) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int slab = wv * 8 + jj;
            const unsigned char* src = reinterpret_cast<const unsigned char*>(p.up) + ((long)ks * 32 + slab) * slab_stride;
            unsigned keep_;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep_)
                         : "v"(boff), "s"(us_base + (unsigned)(st * WSF_USTAGE + slab * 1024)), "s"(src)
                         : "memory");
        }
    };

    // ---- raw patch by LDS-DMA (no operand transform to apply): instruction n of a wave fills LDS rows 16 n' .. 16 n' + 15
    // (n' = wave + 4 n), lane i -> row 16 n' + (i >> 2), 16-byte slot i & 3, fetching the chunk that belongs there; halo columns,
    // pad entries and rows outside the image carry an out-of-range offset and arrive as zeros: nothing to clear, ever
    constexpr int WSF_NRD = 12;
    int voff[WSF_NRD];
    const int rows_used = (OR + 2) * 2 * PW;
    if (!INT) {
#pragma unroll
        for (int n = 0; n < WSF_NRD; ++n) {
            const int R = 16 * (wv + 4 * n) + (lane >> 2);
            const int prow = R / (2 * PW), rem = R - prow * 2 * PW;
            const int plane = rem >= PW ? 1 : 0, idx = rem - plane * PW;
            const int pc = 2 * idx + plane, h = h0 - 1 + prow;
            const int q = (lane & 3) ^ ((R >> 2) & 3);
            const bool ok = R < rows_used && pc >= 1 && pc <= W && (unsigned)h < (unsigned)p.H;
            voff[n] = ok ? ((h * W + pc - 1) * p.K + q * 4) * 4 : OOB;
        }
    }
    const unsigned raw_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Raw);
    auto rdma = [&](int ks, int buf) {
#pragma unroll
        for (int n = 0; n < WSF_NRD; ++n) {
            if (16 * (wv + 4 * n) < rows_used) {                 // wave-uniform
                unsigned keep_;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep_)
                             : "v"(voff[n]), "s"(raw_base + (unsigned)(buf * WSF_RAWBYTES + (wv + 4 * n) * 1024)), "s"(xrs), "s"(ks * 64)
                             : "memory");
            }
        }
    };

    if (INT) {
        // ---- zero what the register staging never writes: the halo columns (patch column 0 and W + 1) and the rows outside the image
        for (int i = tid; i < (OR + 2) * 16; i += 256) {
            const int prow = i >> 4, side = (i >> 3) & 1, buf = (i >> 2) & 1, ch = i & 3;
            const int pc = side ? W + 1 : 0;
            *reinterpret_cast<float4*>(Raw + buf * WSF_RAWBYTES + wsf_raw((prow * 2 + (pc & 1)) * PW + (pc >> 1), ch)) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < WSF_NI; ++i)
            if (val[i] && !sok[i]) {
                *reinterpret_cast<float4*>(Raw + lso[i]) = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(Raw + WSF_RAWBYTES + lso[i]) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        aload(0);
        udma(0, 0);
        astore(0);
    } else {
        udma(0, 0);
        rdma(0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- fragment addressing: this lane's tile and its half of the K-step
    const int kh = lane >> 5;
    const int t = 32 * wv + (lane & 31);
    const int tx = t & ((W >> 1) - 1), ty = t >> (logW - 1);
    int roff[4][4];                     // chunk 2 kh of the lane's 32 bytes; chunk 2 kh + 1 sits at roff ^ 16
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) roff[r][c] = wsf_raw(((2 * ty + r) * 2 + (c & 1)) * PW + tx + (c >> 1), 2 * kh);
    const int uoff = wsf_usw(lane & 31, kh);

    floatx16 acc[16];
#pragma unroll
    for (int a = 0; a < 16; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    for (int ks = 0; ks < KT; ++ks) {
        const int st = ks & 1;
        if (ks + 1 < KT && !(ABL & 1)) {
            udma(ks + 1, st ^ 1);
            if (INT) aload(ks + 1); else rdma(ks + 1, st ^ 1);
        }
        const unsigned char* const Rb = Raw + st * WSF_RAWBYTES;
        const unsigned char* const Ub = Us + st * WSF_USTAGE;
        // Software pipeline (one wave per SIMD: nothing else hides an LDS round trip): every LDS read is REQUESTED one phase
        // before its first use -- raw rows 0 / 2 and U[eta = 0] first; row 1 and U[eta = 1] under the transform + MFMAs of eta = 0;
        // row 3 and U[eta = 2], U[eta = 3] under those of eta = 1, 2 -- and (register staging) the next K-step's patch is
        // written to the other raw buffer in front of the last MFMA group.
        float d[4][4][8];
        half8 ub[16][2];
        auto rd_row = [&](int r) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 a0 = *reinterpret_cast<const float4*>(Rb + roff[r][c]);
                const float4 a1 = *reinterpret_cast<const float4*>(Rb + (roff[r][c] ^ 16));
                d[r][c][0] = a0.x; d[r][c][1] = a0.y; d[r][c][2] = a0.z; d[r][c][3] = a0.w;
                d[r][c][4] = a1.x; d[r][c][5] = a1.y; d[r][c][6] = a1.z; d[r][c][7] = a1.w;
            }
        };
        auto rd_u = [&](int eta) {
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                ub[eta * 4 + xi][0] = *reinterpret_cast<const half8*>(Ub + ((eta * 4 + xi) * 2) * 1024 + uoff);
                ub[eta * 4 + xi][1] = *reinterpret_cast<const half8*>(Ub + ((eta * 4 + xi) * 2 + 1) * 1024 + uoff);
            }
        };
        float tr[4][4][8];
        auto row_t = [&](int r) {       // column transform of one raw patch row: t[xi][8 channels]
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                tr[r][0][e] = d[r][0][e] - d[r][2][e];
                tr[r][1][e] = d[r][1][e] + d[r][2][e];
                tr[r][2][e] = d[r][2][e] - d[r][1][e];
                tr[r][3][e] = d[r][1][e] - d[r][3][e];
            }
        };
        // one Winograd position: V (8 channels, fp32) -> (hi, lo) fragments -> three MFMAs against U[pos]
        auto mma = [&](int pos, const float (&v)[8]) {
            unsigned h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ABL & 2) { h[e] = __float_as_uint(v[2 * e]); l[e] = __float_as_uint(v[2 * e + 1]); }
                else if (INT) sed_sf_split2(v[2 * e], v[2 * e + 1], h[e], l[e]);        // staged values are already scaled
                else wsf_split2s(v[2 * e], v[2 * e + 1], sa, h[e], l[e]);
            }
            const uintx4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
            const half8 ah = __builtin_bit_cast(half8, hv), al = __builtin_bit_cast(half8, lv);
            if (ABL & 4) { acc[pos][0] += (float)al[0] * (float)ub[pos][0][0] + (float)ah[1] * (float)ub[pos][1][1]; return; }
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ub[pos][0], acc[pos], 0, 0, 0);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ub[pos][1], acc[pos], 0, 0, 0);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ub[pos][0], acc[pos], 0, 0, 0);
        };
        float v[8];
        rd_row(0); rd_row(2); rd_u(0);
        __builtin_amdgcn_sched_barrier(0);
        row_t(0); row_t(2);
        rd_row(1); rd_u(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tr[0][xi][e] - tr[2][xi][e];
            mma(0 * 4 + xi, v);
        }
        __builtin_amdgcn_sched_barrier(0);
        row_t(1);
        rd_row(3); rd_u(2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tr[1][xi][e] + tr[2][xi][e];
            mma(1 * 4 + xi, v);
        }
        rd_u(3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tr[2][xi][e] - tr[1][xi][e];
            mma(2 * 4 + xi, v);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (INT && ks + 1 < KT) astore(st ^ 1);
        row_t(3);
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tr[1][xi][e] - tr[3][xi][e];
            mma(3 * 4 + xi, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (overflow) {
        if (p.err_host) __hip_atomic_store(p.err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (p.err_dev) __hip_atomic_store(p.err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- output transform Y = A^T M A (lane-local: register r of every accumulator is the same (tile, channel) pair), unscale,
    // store; rows past the image fall outside the descriptor and are dropped
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (long)b * p.H * W * p.N, 0,
                                                                         (int)((unsigned)p.H * W * p.N * 4u), 0x00020000);
    const int col = n0 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ti = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int otx = ti & ((W >> 1) - 1), oty = ti >> (logW - 1);
        float s0[4], s1[4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            s0[xi] = (acc[xi][r] + acc[4 + xi][r]) + acc[8 + xi][r];
            s1[xi] = (acc[4 + xi][r] - acc[8 + xi][r]) - acc[12 + xi][r];
        }
        const float y00 = ((s0[0] + s0[1]) + s0[2]) * inv, y01 = ((s0[1] - s0[2]) - s0[3]) * inv;
        const float y10 = ((s1[0] + s1[1]) + s1[2]) * inv, y11 = ((s1[1] - s1[2]) - s1[3]) * inv;
        const int h = h0 + 2 * oty, w = 2 * otx;
        const int o00 = h < p.H ? ((h * W + w) * p.N + col) * 4 : OOB;
        const int o10 = h + 1 < p.H ? (((h + 1) * W + w) * p.N + col) * 4 : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y00), yrs, o00, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y01), yrs, o00 == OOB ? OOB : o00 + p.N * 4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y10), yrs, o10, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y11), yrs, o10 == OOB ? OOB : o10 + p.N * 4, 0, 0);
    }
}

// =====================================================================================================================
// Second cut (v2, "eta halves"): the same workgroup tile (128 tiles x 32 channels) on EIGHT waves = two per SIMD, so that one
// wave's MFMAs run under the other's transform / split / LDS traffic (what the direct kernel gets from three workgroups per CU).
// A wave owns 32 tiles x 32 channels x EIGHT positions (two eta rows: 8 accumulators = 128 registers); waves 0-3 take eta = 0, 1
// (patch rows 0, 1, 2), waves 4-7 eta = 2, 3 (rows 1, 2, 3); the rows are combined FIRST (w_a = A - B, w_b = B +- C: the eta
// transform commutes with the column transform), four channels at a time, so that a K-step holds at most 48 raw registers; the
// eta halves meet once, in the epilogue, through LDS (half 0 finishes output row 2t, half 1 row 2t + 1).  Raw patch by LDS-DMA
// only (no fused operand transform in this cut).
template <int ABL = 0, int ROT = 0>      // ROT: 0 plain, 1 rotated eta halves, 2 group-b transform interleaved with group-a MFMAs
__global__ __launch_bounds__(512, 1) void conv_wsf16h_kernel(WsfP p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WSF_RAWBYTES + 2 * WSF_USTAGE];
    unsigned char* const Raw = smem;
    unsigned char* const Us = smem + 2 * WSF_RAWBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tbk = wv & 3, hf = wv >> 2;
    const int W = p.W, logW = p.logW, OR = p.OR, PW = p.PW;
    const int KT = p.K >> 4;
    const int nb = p.N >> 5;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    int cb, tb;
    if (nb >= 8) { const int cpx = nb >> 3; cb = xcd * cpx + j % cpx; tb = j / cpx; }
    else { const int share = 8 / nb; cb = xcd % nb; tb = j * share + xcd / nb; }
    if (tb >= p.ntb) return;
    const int n0 = cb * 32;
    const int b = tb / p.nrb, h0 = (tb % p.nrb) * OR;

    const float sa = sed_sf_scale_of(amax_read(p.x_amax)) * 0.25f;
    const float inv = 1.0f / (sa * p.wscale[SED_AMAX_SLOTS]);

    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (long)b * p.H * W * p.K, 0, (int)((unsigned)p.H * W * p.K * 4u), 0x00020000);

    // ---- U by LDS-DMA: 32 slabs of 1 KB per stage, 4 per wave
    const int brow = lane >> 1;
    const int boff = (n0 + brow) * 32 + ((((lane & 1) ^ ((brow >> 3) & 1))) << 4);
    const unsigned us_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Us);
    const long slab_stride = (long)p.N * 32;
    auto udma = [&](int ks, int st) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int slab = wv * 4 + jj;
            const unsigned char* src = reinterpret_cast<const unsigned char*>(p.up) + ((long)ks * 32 + slab) * slab_stride;
            unsigned keep_;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep_)
                         : "v"(boff), "s"(us_base + (unsigned)(st * WSF_USTAGE + slab * 1024)), "s"(src)
                         : "memory");
        }
    };
    // ---- raw patch by LDS-DMA: 16-row blocks, block wv + 8 n of a stage (n < 6)
    constexpr int NRD = 6;
    int voff[NRD];
    const int rows_used = (OR + 2) * 2 * PW;
#pragma unroll
    for (int n = 0; n < NRD; ++n) {
        const int R = 16 * (wv + 8 * n) + (lane >> 2);
        const int prow = R / (2 * PW), rem = R - prow * 2 * PW;
        const int plane = rem >= PW ? 1 : 0, idx = rem - plane * PW;
        const int pc = 2 * idx + plane, h = h0 - 1 + prow;
        const int q = (lane & 3) ^ ((R >> 2) & 3);
        const bool ok = R < rows_used && pc >= 1 && pc <= W && (unsigned)h < (unsigned)p.H;
        voff[n] = ok ? ((h * W + pc - 1) * p.K + q * 4) * 4 : OOB;
    }
    const unsigned raw_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)Raw);
    auto rdma = [&](int ks, int buf) {
#pragma unroll
        for (int n = 0; n < NRD; ++n) {
            if (16 * (wv + 8 * n) < rows_used) {                 // wave-uniform
                unsigned keep_;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep_)
                             : "v"(voff[n]), "s"(raw_base + (unsigned)(buf * WSF_RAWBYTES + (wv + 8 * n) * 1024)), "s"(xrs), "s"(ks * 64)
                             : "memory");
            }
        }
    };
    udma(0, 0);
    rdma(0, 0);
    if (ABL & 1) { udma(1, 1); rdma(1, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- fragment addressing: this lane's tile, its half of the K-step, this wave's three patch rows
    const int kh = lane >> 5;
    const int t = 32 * tbk + (lane & 31);
    const int tx = t & ((W >> 1) - 1), ty = t >> (logW - 1);
    const int rA = hf ? 2 : 0, rB = hf ? 1 : 2, rC = hf ? 3 : 1;
    const float sgn = hf ? -1.0f : 1.0f;             // w_b = B + sgn * C
    int roff[3][4];                                   // chunk 2 kh (channels 8 kh .. + 3); chunk 2 kh + 1 sits at roff ^ 16
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        roff[0][c] = wsf_raw(((2 * ty + rA) * 2 + (c & 1)) * PW + tx + (c >> 1), 2 * kh);
        roff[1][c] = wsf_raw(((2 * ty + rB) * 2 + (c & 1)) * PW + tx + (c >> 1), 2 * kh);
        roff[2][c] = wsf_raw(((2 * ty + rC) * 2 + (c & 1)) * PW + tx + (c >> 1), 2 * kh);
    }
    const int uoff = wsf_usw(lane & 31, kh) + hf * 8 * 2048;     // this half's positions start at 8 hf

    floatx16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 sgn2 = {sgn, sgn};
    // A workgroup barrier per K-step puts all eight waves into the SAME phase: with "transform, then MFMAs" both waves of a SIMD
    // transform together and then queue for the matrix pipe together (ablation: removing the MFMAs removed their whole pipe time --
    // nothing had overlapped them).  ROT = true (measured SLOWER, kept for the record: WSF_V2=2): the two waves of a SIMD are waves
    // w and w + 4 = the two eta halves of one tile block, so the eta-half-1 waves run their loop ROTATED by half a K-step against
    // two barriers per K-step:
    //     first half :  half 0 transforms K-step ks      ||  half 1 issues the MFMAs of K-step ks - 1
    //     second half:  half 0 issues the MFMAs of ks    ||  half 1 transforms ks
    // -- one wave's VALU / LDS work under the other's MFMAs on every SIMD, with no more registers than the un-rotated loop (a
    // wave still transforms, then multiplies; only the barriers moved).  Buffer lifetimes: raw[st] is read until the second half
    // of its K-step, U[st] until the first half of the next one: the raw DMA of ks + 1 goes out at the top of ks, the U DMA of
    // ks + 1 behind the first barrier of ks.
#define WSFH_T(ST)                                                                                                      \
    {                                                                                                                   \
        const unsigned char* const Rb = Raw + (ST) * WSF_RAWBYTES;                                                      \
        _Pragma("unroll") for (int cq = 0; cq < 2; ++cq) {                                                              \
            f2 wa[4][2], wb[4][2];     /* row-combined patch, [column][channel pair]: packed fp32 throughout */            \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                             \
                const float4 A = *reinterpret_cast<const float4*>(Rb + (roff[0][c] ^ (cq << 4)));                       \
                const float4 B = *reinterpret_cast<const float4*>(Rb + (roff[1][c] ^ (cq << 4)));                       \
                const float4 C = *reinterpret_cast<const float4*>(Rb + (roff[2][c] ^ (cq << 4)));                       \
                const f2 A0 = {A.x, A.y}, A1 = {A.z, A.w}, B0 = {B.x, B.y}, B1 = {B.z, B.w}, C0 = {C.x, C.y}, C1 = {C.z, C.w}; \
                wa[c][0] = A0 - B0; wa[c][1] = A1 - B1;                                                                 \
                wb[c][0] = __builtin_elementwise_fma(sgn2, C0, B0); wb[c][1] = __builtin_elementwise_fma(sgn2, C1, B1); \
            }                                                                                                           \
            _Pragma("unroll") for (int ab = 0; ab < 2; ++ab) {                                                          \
                _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                         \
                    const f2 w0 = ab ? wb[0][e] : wa[0][e], w1 = ab ? wb[1][e] : wa[1][e];                              \
                    const f2 w2 = ab ? wb[2][e] : wa[2][e], w3 = ab ? wb[3][e] : wa[3][e];                              \
                    const f2 V0 = w0 - w2, V1 = w1 + w2, V2 = w2 - w1, V3 = w1 - w3;                                    \
                    if (ABL & 2) {                                                                                      \
                        fh[ab * 4 + 0][2 * cq + e] = __float_as_uint(V0[0]); fl[ab * 4 + 0][2 * cq + e] = __float_as_uint(V0[1]); \
                        fh[ab * 4 + 1][2 * cq + e] = __float_as_uint(V1[0]); fl[ab * 4 + 1][2 * cq + e] = __float_as_uint(V1[1]); \
                        fh[ab * 4 + 2][2 * cq + e] = __float_as_uint(V2[0]); fl[ab * 4 + 2][2 * cq + e] = __float_as_uint(V2[1]); \
                        fh[ab * 4 + 3][2 * cq + e] = __float_as_uint(V3[0]); fl[ab * 4 + 3][2 * cq + e] = __float_as_uint(V3[1]); \
                    } else {                                                                                            \
                        wsf_split2s(V0[0], V0[1], sa, fh[ab * 4 + 0][2 * cq + e], fl[ab * 4 + 0][2 * cq + e]);          \
                        wsf_split2s(V1[0], V1[1], sa, fh[ab * 4 + 1][2 * cq + e], fl[ab * 4 + 1][2 * cq + e]);          \
                        wsf_split2s(V2[0], V2[1], sa, fh[ab * 4 + 2][2 * cq + e], fl[ab * 4 + 2][2 * cq + e]);          \
                        wsf_split2s(V3[0], V3[1], sa, fh[ab * 4 + 3][2 * cq + e], fl[ab * 4 + 3][2 * cq + e]);          \
                    }                                                                                                   \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
    }
#define WSFH_M(ST)                                                                                                      \
    {                                                                                                                   \
        const unsigned char* const Ub = Us + (ST) * WSF_USTAGE;                                                         \
        _Pragma("unroll") for (int pos = 0; pos < 8; ++pos) {                                                           \
            const uintx4 hv = {fh[pos][0], fh[pos][1], fh[pos][2], fh[pos][3]}, lv = {fl[pos][0], fl[pos][1], fl[pos][2], fl[pos][3]}; \
            const half8 ah = __builtin_bit_cast(half8, hv), al = __builtin_bit_cast(half8, lv);                         \
            const half8 bh = *reinterpret_cast<const half8*>(Ub + (pos * 2) * 1024 + uoff);                             \
            const half8 bl = *reinterpret_cast<const half8*>(Ub + (pos * 2 + 1) * 1024 + uoff);                         \
            if (ABL & 4) { acc[pos][0] += (float)al[0] * (float)bh[0] + (float)ah[1] * (float)bl[1]; continue; }         \
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[pos], 0, 0, 0);                               \
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[pos], 0, 0, 0);                               \
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[pos], 0, 0, 0);                               \
        }                                                                                                               \
    }
    // K-step KS on stage ST (compile-time stage: every LDS offset is lane base + immediate) as seen by eta half HF (compile-time:
    // the two halves run two SEPARATE loops -- one scalar branch at the top, no control-flow merges inside)
#define WSFH_STEP(KS, ST, HF)                                                                                           \
    {                                                                                                                   \
        if ((KS) + 1 < KT && !(ABL & 1)) rdma((KS) + 1, (ST) ^ 1);                                                      \
        if ((HF) == 0) WSFH_T(ST) else if ((KS) > 0) WSFH_M((ST) ^ 1)                                                   \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
        __syncthreads();                                                                                                \
        if ((KS) + 1 < KT && !(ABL & 1)) udma((KS) + 1, (ST) ^ 1);                                                      \
        if ((HF) == 0) WSFH_M(ST) else WSFH_T(ST)                                                                       \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
        __syncthreads();                                                                                                \
    }
    // ROT = 2: inside a wave, the transform of eta row b runs BETWEEN the MFMAs of eta row a (program order pinned by scheduling
    // barriers: one MFMA, then a twelfth of the row-b work), no loop-carried state: T_a ; [M_a || T_b] ; M_b
#define WSFH_SB() __builtin_amdgcn_sched_barrier(0);
    // group-a transform of channel quad CQ: rows A, B -> w_a -> V -> fragment words 2 CQ, 2 CQ + 1 of positions 0..3
#define WSFH_TA(RB, CQ)                                                                                                 \
    {                                                                                                                   \
        f2 w_[4][2];                                                                                                    \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                 \
            const float4 A = *reinterpret_cast<const float4*>((RB) + (roff[0][c] ^ ((CQ) << 4)));                       \
            const float4 B = *reinterpret_cast<const float4*>((RB) + (roff[1][c] ^ ((CQ) << 4)));                       \
            const f2 A0 = {A.x, A.y}, A1 = {A.z, A.w}, B0 = {B.x, B.y}, B1 = {B.z, B.w};                                \
            w_[c][0] = A0 - B0; w_[c][1] = A1 - B1;                                                                     \
        }                                                                                                               \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                                 \
            const f2 V0 = w_[0][e] - w_[2][e], V1 = w_[1][e] + w_[2][e], V2 = w_[2][e] - w_[1][e], V3 = w_[1][e] - w_[3][e]; \
            wsf_split2s(V0[0], V0[1], sa, fh[0][2 * (CQ) + e], fl[0][2 * (CQ) + e]);                                    \
            wsf_split2s(V1[0], V1[1], sa, fh[1][2 * (CQ) + e], fl[1][2 * (CQ) + e]);                                    \
            wsf_split2s(V2[0], V2[1], sa, fh[2][2 * (CQ) + e], fl[2][2 * (CQ) + e]);                                    \
            wsf_split2s(V3[0], V3[1], sa, fh[3][2 * (CQ) + e], fl[3][2 * (CQ) + e]);                                    \
        }                                                                                                               \
    }
    // one MFMA of position POS: product PR (0: lo*hi, 1: hi*lo, 2: hi*hi) with fragments AH / AL and U fragments BH / BL
#define WSFH_MF(POS, PR, AH, AL, BH, BL)                                                                                \
    acc[POS] = __builtin_amdgcn_mfma_f32_32x32x16_f16((PR) == 0 ? AL : AH, (PR) == 1 ? BL : BH, acc[POS], 0, 0, 0);
#define WSFH_FRAG(POS, AH, AL)                                                                                          \
    const uintx4 hv##POS = {fh[POS][0], fh[POS][1], fh[POS][2], fh[POS][3]}, lv##POS = {fl[POS][0], fl[POS][1], fl[POS][2], fl[POS][3]}; \
    const half8 AH = __builtin_bit_cast(half8, hv##POS), AL = __builtin_bit_cast(half8, lv##POS);
#define WSFH_UB(UB, POS, BH, BL)                                                                                        \
    const half8 BH = *reinterpret_cast<const half8*>((UB) + ((POS) * 2) * 1024 + uoff);                                 \
    const half8 BL = *reinterpret_cast<const half8*>((UB) + ((POS) * 2 + 1) * 1024 + uoff);
    // group-b work of channel quad CQ in six slices, one behind each of six group-a MFMAs
#define WSFH_STEP2(KS, ST)                                                                                              \
    {                                                                                                                   \
        if ((KS) + 1 < KT && !(ABL & 1)) { udma((KS) + 1, (ST) ^ 1); rdma((KS) + 1, (ST) ^ 1); }                        \
        const unsigned char* const Rb = Raw + (ST) * WSF_RAWBYTES;                                                      \
        const unsigned char* const Ub = Us + (ST) * WSF_USTAGE;                                                         \
        WSFH_TA(Rb, 0) WSFH_TA(Rb, 1)                                                                                   \
        WSFH_UB(Ub, 0, bh0, bl0) WSFH_UB(Ub, 1, bh1, bl1)                                                               \
        WSFH_FRAG(0, ah0, al0) WSFH_FRAG(1, ah1, al1) WSFH_FRAG(2, ah2, al2) WSFH_FRAG(3, ah3, al3)                     \
        WSFH_SB()                                                                                                       \
        f2 wq[4][2];                                                                                                    \
        /* ---- channel quad 0 of group b under the MFMAs of positions 0, 1 */                                          \
        float4 Bq[4], Cq[4];                                                                                            \
        WSFH_MF(0, 0, ah0, al0, bh0, bl0)                                                                               \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) { Bq[c] = *reinterpret_cast<const float4*>(Rb + roff[1][c]); Cq[c] = *reinterpret_cast<const float4*>(Rb + roff[2][c]); } \
        WSFH_SB()                                                                                                       \
        WSFH_MF(0, 1, ah0, al0, bh0, bl0)                                                                               \
        WSFH_UB(Ub, 2, bh2, bl2) WSFH_UB(Ub, 3, bh3, bl3)                                                               \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                 \
            const f2 B0 = {Bq[c].x, Bq[c].y}, B1 = {Bq[c].z, Bq[c].w}, C0 = {Cq[c].x, Cq[c].y}, C1 = {Cq[c].z, Cq[c].w}; \
            wq[c][0] = __builtin_elementwise_fma(sgn2, C0, B0); wq[c][1] = __builtin_elementwise_fma(sgn2, C1, B1);     \
        }                                                                                                               \
        WSFH_SB()                                                                                                       \
        WSFH_MF(0, 2, ah0, al0, bh0, bl0)                                                                               \
        { const f2 V0 = wq[0][0] - wq[2][0], V1 = wq[1][0] + wq[2][0];                                                  \
          wsf_split2s(V0[0], V0[1], sa, fh[4][0], fl[4][0]); wsf_split2s(V1[0], V1[1], sa, fh[5][0], fl[5][0]); }       \
        WSFH_SB()                                                                                                       \
        WSFH_MF(1, 0, ah1, al1, bh1, bl1)                                                                               \
        { const f2 V2 = wq[2][0] - wq[1][0], V3 = wq[1][0] - wq[3][0];                                                  \
          wsf_split2s(V2[0], V2[1], sa, fh[6][0], fl[6][0]); wsf_split2s(V3[0], V3[1], sa, fh[7][0], fl[7][0]); }       \
        WSFH_SB()                                                                                                       \
        WSFH_MF(1, 1, ah1, al1, bh1, bl1)                                                                               \
        { const f2 V0 = wq[0][1] - wq[2][1], V1 = wq[1][1] + wq[2][1];                                                  \
          wsf_split2s(V0[0], V0[1], sa, fh[4][1], fl[4][1]); wsf_split2s(V1[0], V1[1], sa, fh[5][1], fl[5][1]); }       \
        WSFH_SB()                                                                                                       \
        WSFH_MF(1, 2, ah1, al1, bh1, bl1)                                                                               \
        { const f2 V2 = wq[2][1] - wq[1][1], V3 = wq[1][1] - wq[3][1];                                                  \
          wsf_split2s(V2[0], V2[1], sa, fh[6][1], fl[6][1]); wsf_split2s(V3[0], V3[1], sa, fh[7][1], fl[7][1]); }       \
        WSFH_SB()                                                                                                       \
        /* ---- channel quad 1 of group b under the MFMAs of positions 2, 3 */                                          \
        WSFH_MF(2, 0, ah2, al2, bh2, bl2)                                                                               \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) { Bq[c] = *reinterpret_cast<const float4*>(Rb + (roff[1][c] ^ 16)); Cq[c] = *reinterpret_cast<const float4*>(Rb + (roff[2][c] ^ 16)); } \
        WSFH_SB()                                                                                                       \
        WSFH_MF(2, 1, ah2, al2, bh2, bl2)                                                                               \
        WSFH_UB(Ub, 4, bh4, bl4) WSFH_UB(Ub, 5, bh5, bl5)                                                               \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                 \
            const f2 B0 = {Bq[c].x, Bq[c].y}, B1 = {Bq[c].z, Bq[c].w}, C0 = {Cq[c].x, Cq[c].y}, C1 = {Cq[c].z, Cq[c].w}; \
            wq[c][0] = __builtin_elementwise_fma(sgn2, C0, B0); wq[c][1] = __builtin_elementwise_fma(sgn2, C1, B1);     \
        }                                                                                                               \
        WSFH_SB()                                                                                                       \
        WSFH_MF(2, 2, ah2, al2, bh2, bl2)                                                                               \
        { const f2 V0 = wq[0][0] - wq[2][0], V1 = wq[1][0] + wq[2][0];                                                  \
          wsf_split2s(V0[0], V0[1], sa, fh[4][2], fl[4][2]); wsf_split2s(V1[0], V1[1], sa, fh[5][2], fl[5][2]); }       \
        WSFH_SB()                                                                                                       \
        WSFH_MF(3, 0, ah3, al3, bh3, bl3)                                                                               \
        { const f2 V2 = wq[2][0] - wq[1][0], V3 = wq[1][0] - wq[3][0];                                                  \
          wsf_split2s(V2[0], V2[1], sa, fh[6][2], fl[6][2]); wsf_split2s(V3[0], V3[1], sa, fh[7][2], fl[7][2]); }       \
        WSFH_SB()                                                                                                       \
        WSFH_MF(3, 1, ah3, al3, bh3, bl3)                                                                               \
        { const f2 V0 = wq[0][1] - wq[2][1], V1 = wq[1][1] + wq[2][1];                                                  \
          wsf_split2s(V0[0], V0[1], sa, fh[4][3], fl[4][3]); wsf_split2s(V1[0], V1[1], sa, fh[5][3], fl[5][3]); }       \
        WSFH_SB()                                                                                                       \
        WSFH_MF(3, 2, ah3, al3, bh3, bl3)                                                                               \
        { const f2 V2 = wq[2][1] - wq[1][1], V3 = wq[1][1] - wq[3][1];                                                  \
          wsf_split2s(V2[0], V2[1], sa, fh[6][3], fl[6][3]); wsf_split2s(V3[0], V3[1], sa, fh[7][3], fl[7][3]); }       \
        WSFH_SB()                                                                                                       \
        /* ---- group b */                                                                                              \
        WSFH_UB(Ub, 6, bh6, bl6) WSFH_UB(Ub, 7, bh7, bl7)                                                               \
        WSFH_FRAG(4, ah4, al4) WSFH_FRAG(5, ah5, al5) WSFH_FRAG(6, ah6, al6) WSFH_FRAG(7, ah7, al7)                     \
        WSFH_MF(4, 0, ah4, al4, bh4, bl4) WSFH_MF(5, 0, ah5, al5, bh5, bl5) WSFH_MF(6, 0, ah6, al6, bh6, bl6) WSFH_MF(7, 0, ah7, al7, bh7, bl7) \
        WSFH_MF(4, 1, ah4, al4, bh4, bl4) WSFH_MF(5, 1, ah5, al5, bh5, bl5) WSFH_MF(6, 1, ah6, al6, bh6, bl6) WSFH_MF(7, 1, ah7, al7, bh7, bl7) \
        WSFH_MF(4, 2, ah4, al4, bh4, bl4) WSFH_MF(5, 2, ah5, al5, bh5, bl5) WSFH_MF(6, 2, ah6, al6, bh6, bl6) WSFH_MF(7, 2, ah7, al7, bh7, bl7) \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
        __syncthreads();                                                                                                \
    }
    // un-rotated K-step (ROT = 0): every wave transforms, then multiplies; one barrier
#define WSFH_STEP1(KS, ST)                                                                                              \
    {                                                                                                                   \
        if ((KS) + 1 < KT && !(ABL & 1)) { udma((KS) + 1, (ST) ^ 1); rdma((KS) + 1, (ST) ^ 1); }                        \
        WSFH_T(ST)                                                                                                      \
        WSFH_M(ST)                                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
        __syncthreads();                                                                                                \
    }
    if (ROT == 2) {
        unsigned fh[8][4], fl[8][4];
        for (int ks = 0; ks < KT; ks += 2) {
            WSFH_STEP2(ks, 0)
            WSFH_STEP2(ks + 1, 1)
        }
    } else if (ROT == 0) {
        unsigned fh[8][4], fl[8][4];                  // (hi, lo) fragment words of the 8 positions: words 0, 1 <- channel quad 0; 2, 3 <- quad 1
        for (int ks = 0; ks < KT; ks += 2) {          // K % 32 == 0 for every layer this cut is run on: two K-steps per trip
            WSFH_STEP1(ks, 0)
            WSFH_STEP1(ks + 1, 1)
        }
    } else if (hf == 0) {
        unsigned fh[8][4], fl[8][4];
        for (int ks = 0; ks < KT; ks += 2) {
            WSFH_STEP(ks, 0, 0)
            WSFH_STEP(ks + 1, 1, 0)
        }
    } else {
        unsigned fh[8][4], fl[8][4];
#pragma unroll
        for (int a_ = 0; a_ < 8; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < 4; ++b_) { fh[a_][b_] = 0u; fl[a_][b_] = 0u; }
        for (int ks = 0; ks < KT; ks += 2) {
            WSFH_STEP(ks, 0, 1)
            WSFH_STEP(ks + 1, 1, 1)
        }
        WSFH_M(1)                                     // the rotated half's last K-step (KT is even: stage 1)
    }
#undef WSFH_STEP
#undef WSFH_STEP1
#undef WSFH_STEP2
#undef WSFH_M
#undef WSFH_T

    // ---- output transform.  Row direction first, per half: half 0 holds M0, M1 -> P = M0 + M1 (its share of output row 0) and
    // Q = M1 (its share of row 1); half 1 holds M2, M3 -> P = M2, Q = -(M2 + M3).  Columns: y[.][0] = v0 + v1 + v2, y[.][1] = v1 - v2 - v3.
    // Half 0 finishes row 0 (needs the other half's P), half 1 row 1 (needs the other half's Q): 32 floats per lane each way, through LDS.
    float* const red = reinterpret_cast<float*>(smem);           // [pair 4][direction 2][32][64 lanes]
    float mine[16][2], give[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float P[4], Q[4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            if (hf == 0) { P[xi] = acc[xi][r] + acc[4 + xi][r]; Q[xi] = acc[4 + xi][r]; }
            else { P[xi] = acc[xi][r]; Q[xi] = -(acc[xi][r] + acc[4 + xi][r]); }
        }
        const float p0 = (P[0] + P[1]) + P[2], p1 = (P[1] - P[2]) - P[3];
        const float q0 = (Q[0] + Q[1]) + Q[2], q1 = (Q[1] - Q[2]) - Q[3];
        if (hf == 0) { mine[r][0] = p0; mine[r][1] = p1; give[r][0] = q0; give[r][1] = q1; }
        else { mine[r][0] = q0; mine[r][1] = q1; give[r][0] = p0; give[r][1] = p1; }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[((tbk * 2 + hf) * 32 + 2 * r) * 64 + lane] = give[r][0];
        red[((tbk * 2 + hf) * 32 + 2 * r + 1) * 64 + lane] = give[r][1];
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (long)b * p.H * W * p.N, 0,
                                                                         (int)((unsigned)p.H * W * p.N * 4u), 0x00020000);
    const int col = n0 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float y0 = (mine[r][0] + red[((tbk * 2 + (hf ^ 1)) * 32 + 2 * r) * 64 + lane]) * inv;
        const float y1 = (mine[r][1] + red[((tbk * 2 + (hf ^ 1)) * 32 + 2 * r + 1) * 64 + lane]) * inv;
        const int ti = 32 * tbk + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int otx = ti & ((W >> 1) - 1), oty = ti >> (logW - 1);
        const int h = h0 + 2 * oty + hf, w = 2 * otx;
        const int o0 = h < p.H ? ((h * W + w) * p.N + col) * 4 : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0), yrs, o0, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y1), yrs, o0 == OOB ? OOB : o0 + p.N * 4, 0, 0);
    }
}

// ---- weights: U = G g G^T per (output channel, input channel) pair; amax of U -> power-of-two scale; (hi, lo) planes.
// dgrad = 1: the operand of the transposed convolution (channel roles swapped, taps flipped).
__device__ __forceinline__ void wsf_u_of(const float* __restrict__ w, int Cout, int Cin, int dgrad, int o, int i, float (&U)[16]) {
    const float* src = w + (dgrad ? ((long)i * Cin + o) : ((long)o * Cin + i)) * 9;
    float g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = src[dgrad ? 8 - k : k];
    float m[4][3];                                    // G g: rows (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) of the 3x3
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        m[0][c] = g[c];
        m[1][c] = 0.5f * ((g[c] + g[3 + c]) + g[6 + c]);
        m[2][c] = 0.5f * ((g[c] - g[3 + c]) + g[6 + c]);
        m[3][c] = g[6 + c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        U[r * 4 + 0] = m[r][0];
        U[r * 4 + 1] = 0.5f * ((m[r][0] + m[r][1]) + m[r][2]);
        U[r * 4 + 2] = 0.5f * ((m[r][0] - m[r][1]) + m[r][2]);
        U[r * 4 + 3] = m[r][2];
    }
}

__global__ __launch_bounds__(256) void wsf_amax_kernel(const float* __restrict__ w, int Cout, int Cin, int dgrad, float* __restrict__ out) {
    const int No = dgrad ? Cin : Cout, Ki = dgrad ? Cout : Cin;
    float mx = 0.f;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < (long)No * Ki; q += (long)gridDim.x * 256) {
        float U[16];
        wsf_u_of(w, Cout, Cin, dgrad, (int)(q / Ki), (int)(q % Ki), U);
#pragma unroll
        for (int k = 0; k < 16; ++k) mx = fmaxf(mx, fabsf(U[k]));
    }
    amax_publish_block(out, mx);
}

// task q = (ks * No + o) * 16 + il (as the direct pack: a wave's stores of one position and plane are contiguous)
__global__ __launch_bounds__(256) void wsf_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int dgrad,
                                                       float* __restrict__ wscale, _Float16* __restrict__ up) {
    const float sw = sed_sf_scale_of(amax_read(wscale));
    if (blockIdx.x == 0 && threadIdx.x == 0) wscale[SED_AMAX_SLOTS] = sw;
    const int No = dgrad ? Cin : Cout, Ki = dgrad ? Cout : Cin;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < (long)No * Ki; q += (long)gridDim.x * 256) {
        const int il = (int)(q & 15);
        const long tt = q >> 4;
        const int o = (int)(tt % No), ks = (int)(tt / No);
        float U[16];
        wsf_u_of(w, Cout, Cin, dgrad, o, ks * 16 + il, U);
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            const float v = U[pos] * sw;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            const long base = (((long)ks * 16 + pos) * 2) * No * 16;
            up[base + (long)o * 16 + il] = hi;
            up[base + (long)No * 16 + (long)o * 16 + il] = lo;
        }
    }
}

int wsf_log2w(int W) { return W == 64 ? 6 : W == 32 ? 5 : W == 16 ? 4 : 3; }
int wsf_pw(int W) { return W == 64 ? 33 : W == 32 ? 20 : W == 16 ? 10 : 5; }

}  // namespace

SED_API int sed_conv3x3_wsf16_supported(int H, int W, int Cin, int Cout) {
    return (W == 8 || W == 16 || W == 32 || W == 64) && H >= 1 && Cin >= 16 && Cin % 16 == 0 && Cout >= 32 && Cout % 32 == 0 &&
           (Cout / 32 >= 8 ? (Cout / 32) % 8 == 0 : 8 % (Cout / 32) == 0);
}

SED_API long sed_conv_wsf16_pack_halfs(int Cin, int Cout) { return 32L * Cin * Cout; }

// w_oihw [Cout][Cin][3][3] -> up (sed_conv_wsf16_pack_halfs halfs) + wscale [SED_AMAX_SLOTS + 1]; dgrad as in sed_pack_conv_weights_sf16
SED_API int sed_pack_conv_weights_wsf16(const float* w_oihw, int Cout, int Cin, int dgrad, void* up, float* wscale, hipStream_t stream) {
    if (!w_oihw || !up || !wscale || Cout <= 0 || Cin <= 0 || (dgrad ? Cout : Cin) % 16) return SED_EINVAL;
    hipError_t e = hipMemsetAsync(wscale, 0, SED_AMAX_SLOTS * sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
    const long pairs = (long)Cout * Cin;
    const int nblk = (int)((pairs + 255) / 256 > 1024 ? 1024 : (pairs + 255) / 256);
    hipLaunchKernelGGL(wsf_amax_kernel, dim3(nblk), dim3(256), 0, stream, w_oihw, Cout, Cin, dgrad ? 1 : 0, wscale);
    hipLaunchKernelGGL(wsf_pack_kernel, dim3(nblk), dim3(256), 0, stream, w_oihw, Cout, Cin, dgrad ? 1 : 0, wscale,
                       reinterpret_cast<_Float16*>(up));
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_conv3x3_wsf16(const float* x, const void* up, const float* wscale, float* y, int B, int H, int W, int Cin, int Cout,
                              const float* in_scale, const float* in_shift, const float* x_amax, int* err_host, int* err_dev,
                              hipStream_t stream) {
    if (!x || !up || !wscale || !y || !x_amax || B <= 0 || !sed_conv3x3_wsf16_supported(H, W, Cin, Cout) ||
        (long)H * W * (Cin > Cout ? Cin : Cout) * 4 >= (1L << 31) || ((in_scale == nullptr) != (in_shift == nullptr)))
        return SED_EINVAL;
    WsfP p{};
    p.x = x; p.up = reinterpret_cast<const _Float16*>(up); p.wscale = wscale; p.x_amax = x_amax; p.y = y;
    p.in_scale = in_scale; p.in_shift = in_shift;
    p.B = B; p.H = H; p.W = W; p.K = Cin; p.N = Cout;
    p.logW = wsf_log2w(W); p.OR = 512 / W; p.nrb = (H + p.OR - 1) / p.OR; p.PW = wsf_pw(W); p.ntb = B * p.nrb;
    p.err_host = err_host; p.err_dev = err_dev;
    { const char* e = getenv("WSF_ABL"); p.abl = e ? atoi(e) : 0; }
    const int nb = Cout / 32;
    const long grid = nb >= 8 ? (long)p.ntb * nb : 8L * ((p.ntb + 8 / nb - 1) / (8 / nb));
    if (grid >= (1L << 31)) return SED_EINVAL;
    { const char* e = getenv("WSF_V2"); if (e && atoi(e) == 3 && !in_scale && Cin % 32 == 0) {
        hipLaunchKernelGGL((conv_wsf16h_kernel<0, 2>), dim3((unsigned)grid), dim3(512), 0, stream, p);
        SED_LAUNCH_CHECK();
        return 0;
    }
    if (e && atoi(e) == 2 && !in_scale && Cin % 32 == 0) {
        hipLaunchKernelGGL((conv_wsf16h_kernel<0, 1>), dim3((unsigned)grid), dim3(512), 0, stream, p);
        SED_LAUNCH_CHECK();
        return 0;
    }
    if (e && atoi(e) == 1 && !in_scale && Cin % 32 == 0) {
        if (p.abl == 1) hipLaunchKernelGGL(conv_wsf16h_kernel<1>, dim3((unsigned)grid), dim3(512), 0, stream, p);
        else if (p.abl == 2) hipLaunchKernelGGL(conv_wsf16h_kernel<2>, dim3((unsigned)grid), dim3(512), 0, stream, p);
        else if (p.abl == 4) hipLaunchKernelGGL(conv_wsf16h_kernel<4>, dim3((unsigned)grid), dim3(512), 0, stream, p);
        else if (p.abl == 6) hipLaunchKernelGGL(conv_wsf16h_kernel<6>, dim3((unsigned)grid), dim3(512), 0, stream, p);
        else if (p.abl == 7) hipLaunchKernelGGL(conv_wsf16h_kernel<7>, dim3((unsigned)grid), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL(conv_wsf16h_kernel<0>, dim3((unsigned)grid), dim3(512), 0, stream, p);
        SED_LAUNCH_CHECK();
        return 0;
    } }
    if (in_scale) hipLaunchKernelGGL((conv_wsf16_kernel<true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else if (p.abl == 1) hipLaunchKernelGGL((conv_wsf16_kernel<false, 1>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else if (p.abl == 2) hipLaunchKernelGGL((conv_wsf16_kernel<false, 2>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else if (p.abl == 4) hipLaunchKernelGGL((conv_wsf16_kernel<false, 4>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else if (p.abl == 3) hipLaunchKernelGGL((conv_wsf16_kernel<false, 3>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else if (p.abl == 6) hipLaunchKernelGGL((conv_wsf16_kernel<false, 6>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else if (p.abl == 7) hipLaunchKernelGGL((conv_wsf16_kernel<false, 7>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((conv_wsf16_kernel<false>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    SED_LAUNCH_CHECK();
    return 0;
}

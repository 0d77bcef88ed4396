// nn.GRU(512, 256, bidirectional) recurrence (reference models.py:529-530, :565-567) as ONE persistent launch per
// pass for BOTH directions: hidden projection h_prev x W_hh^T on fp32 MFMA + the gate math for all T steps, instead of
// T x (a 128x128-tile GEMM launch on 8-24 workgroups, 10-19 us, + a gate launch + two dependent-launch gaps).
//
// The recurrence is latency-bound (125 dependent steps of 0.1 GFLOP per direction), so the kernel is laid out for a
// short per-step critical path, not for MFMA utilisation:
//  * workgroup = 32 batch rows x 32 hidden units (x 3 gates) of one direction -> (H/32) x ceil(B/32) x 2 = 128
//    workgroups at B=256, all co-resident (1 per CU); its 8 waves split the K reduction;
//  * the workgroup's slice of W_hh (96 rows x 256, 96 KB) lives in REGISTERS for the whole sequence (48 VGPRs per lane:
//    the MFMA k index is permuted consistently for both operands, which a dot product allows, so every lane holds
//    contiguous float4 runs of its weight row); per step only the 32 x 256 h_prev block is loaded;
//  * a step of direction d / row block rb depends only on the 8 workgroups (hidden blocks) of the same (d, rb): they
//    synchronise through one monotonic counter in global memory (release: barrier, agent-scope fence, atomic add;
//    acquire: bounded spin on the counter, fence, barrier) - no grid-wide barrier.  The members of a group have
//    linear ids group + 16*jb, i.e. land on one XCD when B = 256;
//  * everything a step needs that does NOT depend on the previous step (gi / g_out / saved gates) is loaded before
//    the wait.
// The spin is bounded: a workgroup that never sees its partners (which cannot happen while all workgroups are
// resident: 128 x 512 threads, 48 KB LDS) gives up after ~1 s and raises the error word instead of hanging the GPU.
#include "common.h"
#include "sed_hip.h"

namespace {

constexpr int GH = 256;                                // hidden size the fused kernels are built for

__device__ __forceinline__ float gru_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// accumulator register r of a 32x32 MFMA tile holds row (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), column lane & 31
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

struct GruStepFwdP {
    const float* gi[2];        // [B] rows (stride ld_gi) x [3H]: input projection incl. b_ih at this direction's time index
    const float* h_prev[2];    // [B][H] or null (h0 = 0)
    const float* w[2];         // W_hh [3H][H]
    const float* bhh[2];       // [3H]
    float* h_out[2];           // [B][H]
    float* out2[2];            // rows (stride ld_out2): the (B,T,2H) output slice
    float* save[2];            // [B][4H] = r, z, n, gh_n
    long ld_gi, ld_out2;
    int B;
};

__global__ __launch_bounds__(512) void gru_step_fwd_kernel(GruStepFwdP p) {
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * 16 * 64];      // 48 KB
    const int d = blockIdx.z, r0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hf = lane >> 5, l31 = lane & 31;
    const float* h_prev = p.h_prev[d];

    floatx16 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    if (h_prev) {
        const int kb = wave * 32 + hf * 16;            // this lane's 16 consecutive k
        const int row = min(r0 + l31, p.B - 1);
        float4 a[4], b[3][4];
        const float4* ap = reinterpret_cast<const float4*>(h_prev + (long)row * GH + kb);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = ap[q];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float4* bp = reinterpret_cast<const float4*>(p.w[d] + (long)(g * GH + j0 + l31) * GH + kb);
#pragma unroll
            for (int q = 0; q < 4; ++q) b[g][q] = bp[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[g][q].x, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[g][q].y, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[g][q].z, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[g][q].w, acc[g], 0, 0, 0);
            }
        // two-phase reduction over the 8 waves: 4..7 -> LDS -> added by 0..3 -> LDS -> summed by the output threads
        if (wave >= 4) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((wave - 4) * 3 + g) * 16 + r) * 64 + lane] = acc[g][r];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] += red[((wave * 3 + g) * 16 + r) * 64 + lane];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave * 3 + g) * 16 + r) * 64 + lane] = acc[g][r];
        }
        __syncthreads();
    }
    // thread -> accumulator register r = tid >> 5 of the lane pair (2q, 2q+1), q = tid & 31: two adjacent hidden units
    const int r = tid >> 5, lp = (tid & 31) * 2;
    const int row = r0 + acc_row(r, lp >> 5);
    if (row >= p.B) return;
    const int j = j0 + (lp & 31);
    float2 gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        float2 s = *reinterpret_cast<const float2*>(p.bhh[d] + g * GH + j);
        if (h_prev) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float2 v = *reinterpret_cast<const float2*>(&red[((w * 3 + g) * 16 + r) * 64 + lp]);
                s.x += v.x; s.y += v.y;
            }
        }
        gh[g] = s;
    }
    const float* gir = p.gi[d] + (long)row * p.ld_gi + j;
    const float2 gr = *reinterpret_cast<const float2*>(gir);
    const float2 gz = *reinterpret_cast<const float2*>(gir + GH);
    const float2 gn = *reinterpret_cast<const float2*>(gir + 2 * GH);
    float2 hp = make_float2(0.f, 0.f);
    if (h_prev) hp = *reinterpret_cast<const float2*>(h_prev + (long)row * GH + j);
    float2 rr, zz, nn, hh;
    rr.x = gru_sigmoid(gr.x + gh[0].x); rr.y = gru_sigmoid(gr.y + gh[0].y);
    zz.x = gru_sigmoid(gz.x + gh[1].x); zz.y = gru_sigmoid(gz.y + gh[1].y);
    nn.x = tanhf(gn.x + rr.x * gh[2].x); nn.y = tanhf(gn.y + rr.y * gh[2].y);
    hh.x = (1.0f - zz.x) * nn.x + zz.x * hp.x; hh.y = (1.0f - zz.y) * nn.y + zz.y * hp.y;
    *reinterpret_cast<float2*>(p.h_out[d] + (long)row * GH + j) = hh;
    *reinterpret_cast<float2*>(p.out2[d] + (long)row * p.ld_out2 + j) = hh;
    float* s = p.save[d] + (long)row * 4 * GH + j;
    *reinterpret_cast<float2*>(s) = rr;
    *reinterpret_cast<float2*>(s + GH) = zz;
    *reinterpret_cast<float2*>(s + 2 * GH) = nn;
    *reinterpret_cast<float2*>(s + 3 * GH) = gh[2];
}

struct GruStepBwdP {
    const float* g_out[2];     // rows (stride ld_go) x [H]: gradient of this direction's output at this time index
    const float* dh_direct[2]; // [B][H] = dh * z of the later step, or null
    const float* dgh_next[2];  // [B][3H] = dgh of the later step (its W_hh path is contracted here), or null
    const float* wt[2];        // W_hh^T [H][3H]
    const float* save[2];      // [B][4H]
    const float* h_prev[2];    // [B][H] or null
    float* dgi[2];             // rows (stride ld_dgi) x [3H]
    float* dgh[2];             // [B][3H]
    float* dh_direct_out[2];   // [B][H]
    long ld_go, ld_dgi;
    int B;
};

__global__ __launch_bounds__(512) void gru_step_bwd_kernel(GruStepBwdP p) {
    __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];          // 16 KB
    const int d = blockIdx.z, r0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hf = lane >> 5, l31 = lane & 31;
    const float* dgh_next = p.dgh_next[d];

    if (dgh_next) {
        // dh_gemm[b][j] = sum_c dgh_next[b][c] * W_hh[c][j], c over 3H = 768: 96 per wave, 48 consecutive per lane
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int kb = wave * 96 + hf * 48;
        const int row = min(r0 + l31, p.B - 1);
        float4 a[12], b[12];
        const float4* ap = reinterpret_cast<const float4*>(dgh_next + (long)row * 3 * GH + kb);
        const float4* bp = reinterpret_cast<const float4*>(p.wt[d] + (long)(j0 + l31) * 3 * GH + kb);
#pragma unroll
        for (int q = 0; q < 12; ++q) { a[q] = ap[q]; b[q] = bp[q]; }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
        }
        if (wave >= 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 4) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[(wave * 16 + r) * 64 + lane];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
    }
    const int r = tid >> 5, lp = (tid & 31) * 2;
    const int row = r0 + acc_row(r, lp >> 5);
    if (row >= p.B) return;
    const int j = j0 + (lp & 31);
    float2 dh = *reinterpret_cast<const float2*>(p.g_out[d] + (long)row * p.ld_go + j);
    if (p.dh_direct[d]) {
        const float2 v = *reinterpret_cast<const float2*>(p.dh_direct[d] + (long)row * GH + j);
        dh.x += v.x; dh.y += v.y;
    }
    if (dgh_next) {
        float2 sum = make_float2(0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float2 v = *reinterpret_cast<const float2*>(&red[(w * 16 + r) * 64 + lp]);
            sum.x += v.x; sum.y += v.y;
        }
        dh.x += sum.x; dh.y += sum.y;
    }
    const float* s = p.save[d] + (long)row * 4 * GH + j;
    const float2 rr = *reinterpret_cast<const float2*>(s), zz = *reinterpret_cast<const float2*>(s + GH);
    const float2 nn = *reinterpret_cast<const float2*>(s + 2 * GH), ghn = *reinterpret_cast<const float2*>(s + 3 * GH);
    float2 hp = make_float2(0.f, 0.f);
    if (p.h_prev[d]) hp = *reinterpret_cast<const float2*>(p.h_prev[d] + (long)row * GH + j);
    float2 dr_pre, dz_pre, dn_pre, dn_r, dhz;
#define SED_GRU_BWD(c)                                                                                          \
    {                                                                                                           \
        const float dn = dh.c * (1.0f - zz.c);                                                                  \
        const float dz = dh.c * (hp.c - nn.c);                                                                  \
        dn_pre.c = dn * (1.0f - nn.c * nn.c);                                                                   \
        const float dr = dn_pre.c * ghn.c;                                                                      \
        dr_pre.c = dr * rr.c * (1.0f - rr.c);                                                                   \
        dz_pre.c = dz * zz.c * (1.0f - zz.c);                                                                   \
        dn_r.c = dn_pre.c * rr.c;                                                                               \
        dhz.c = dh.c * zz.c;                                                                                    \
    }
    SED_GRU_BWD(x) SED_GRU_BWD(y)
#undef SED_GRU_BWD
    float* gi_o = p.dgi[d] + (long)row * p.ld_dgi + j;
    float* gh_o = p.dgh[d] + (long)row * 3 * GH + j;
    *reinterpret_cast<float2*>(gi_o) = dr_pre;
    *reinterpret_cast<float2*>(gi_o + GH) = dz_pre;
    *reinterpret_cast<float2*>(gi_o + 2 * GH) = dn_pre;
    *reinterpret_cast<float2*>(gh_o) = dr_pre;
    *reinterpret_cast<float2*>(gh_o + GH) = dz_pre;
    *reinterpret_cast<float2*>(gh_o + 2 * GH) = dn_r;
    *reinterpret_cast<float2*>(p.dh_direct_out[d] + (long)row * GH + j) = dhz;
}

}  // namespace

SED_API int sed_gru_seq_supported(int Hd) { return Hd == GH; }

// Whole forward recurrence, T launches enqueued from here.  Direction 0 walks t = 0..T-1, direction 1 walks t = T-1..0.
SED_API int sed_gru_seq_fwd(const float* gi, const float* w_hh_f, const float* w_hh_b, const float* b_hh_f,
                            const float* b_hh_b, int B, int T, int Hd, float* hs, float* saves, float* out,
                            hipStream_t stream) {
    if (B <= 0 || T <= 0 || Hd != GH) return SED_EINVAL;
    const long bh = (long)B * GH;
    for (int k = 0; k < T; ++k) {
        const int tf = k, tb = T - 1 - k;
        GruStepFwdP p;
        p.gi[0] = gi + (long)tf * 6 * GH;            p.gi[1] = gi + (long)tb * 6 * GH + 3 * GH;
        p.h_prev[0] = k ? hs + (long)(tf - 1) * bh : nullptr;
        p.h_prev[1] = k ? hs + ((long)T + tb + 1) * bh : nullptr;
        p.w[0] = w_hh_f; p.w[1] = w_hh_b; p.bhh[0] = b_hh_f; p.bhh[1] = b_hh_b;
        p.h_out[0] = hs + (long)tf * bh;             p.h_out[1] = hs + ((long)T + tb) * bh;
        p.out2[0] = out + (long)tf * 2 * GH;         p.out2[1] = out + (long)tb * 2 * GH + GH;
        p.save[0] = saves + (long)tf * 4 * bh;       p.save[1] = saves + ((long)T + tb) * 4 * bh;
        p.ld_gi = (long)T * 6 * GH; p.ld_out2 = (long)T * 2 * GH; p.B = B;
        hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(GH / 32, sed_cdiv(B, 32), 2), dim3(512), 0, stream, p);
    }
    SED_LAUNCH_CHECK();
    return 0;
}

// Whole backward recurrence (reverse processing order).  ws: 4*B*H floats (ping-pong dh*z buffers, both directions).
SED_API int sed_gru_seq_bwd(const float* g_out, const float* wt_f, const float* wt_b, const float* hs, const float* saves,
                            int B, int T, int Hd, float* dgi, float* dgh, float* ws, hipStream_t stream) {
    if (B <= 0 || T <= 0 || Hd != GH) return SED_EINVAL;
    const long bh = (long)B * GH;
    for (int k = T - 1; k >= 0; --k) {
        const int tf = k, tb = T - 1 - k;
        const bool later = k < T - 1;
        float* cur = ws + (long)(k & 1) * 2 * bh;
        const float* prv = ws + (long)((k & 1) ^ 1) * 2 * bh;
        GruStepBwdP p;
        p.g_out[0] = g_out + (long)tf * 2 * GH;      p.g_out[1] = g_out + (long)tb * 2 * GH + GH;
        p.dh_direct[0] = later ? prv : nullptr;      p.dh_direct[1] = later ? prv + bh : nullptr;
        p.dgh_next[0] = later ? dgh + (long)(tf + 1) * 3 * bh : nullptr;
        p.dgh_next[1] = later ? dgh + ((long)T + tb - 1) * 3 * bh : nullptr;
        p.wt[0] = wt_f; p.wt[1] = wt_b;
        p.save[0] = saves + (long)tf * 4 * bh;       p.save[1] = saves + ((long)T + tb) * 4 * bh;
        p.h_prev[0] = k ? hs + (long)(tf - 1) * bh : nullptr;
        p.h_prev[1] = k ? hs + ((long)T + tb + 1) * bh : nullptr;
        p.dgi[0] = dgi + (long)tf * 6 * GH;          p.dgi[1] = dgi + (long)tb * 6 * GH + 3 * GH;
        p.dgh[0] = dgh + (long)tf * 3 * bh;          p.dgh[1] = dgh + ((long)T + tb) * 3 * bh;
        p.dh_direct_out[0] = cur;                    p.dh_direct_out[1] = cur + bh;
        p.ld_go = (long)T * 2 * GH; p.ld_dgi = (long)T * 6 * GH; p.B = B;
        hipLaunchKernelGGL(gru_step_bwd_kernel, dim3(GH / 32, sed_cdiv(B, 32), 2), dim3(512), 0, stream, p);
    }
    SED_LAUNCH_CHECK();
    return 0;
}

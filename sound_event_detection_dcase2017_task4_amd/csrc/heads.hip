// Heads, loss, mixup-of-targets, optimiser and GRU gate kernels.  The dense contractions of the heads
// (fc / AttBlock 1x1 convs / GRU projections) run on the fp32-MFMA GEMMs in conv.hip (sed_gemm_nt / sed_gemm_tn);
// the kernels here are the small per-clip / per-element stages around them.
//
// Reference: FrameAvg head models.py:306-312, FrameMax :221-227, AttBlock :135-143, interpolate :58-69,
// nn.GRU gates :529-530, clip_bce losses.py:5-12, do_mixup pytorch_utils.py:80-93, Adam(amsgrad) main.py:144-145.
#include "common.h"
#include <stdio.h>
#include <string.h>
#include "sed_hip.h"
SED_OBJECT_FLAGS(heads)

namespace {

constexpr int NCLS_MAX = 32;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// column sums of a [n][K] fp32 matrix accumulated in fp64 (bias gradients, per-clip partials)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ parts, long n, int K, long ld,
                                                          float* __restrict__ out, int accumulate) {
    int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= K) return;
    // eight independent fp64 chains (a single dependent chain over the 256 chunk rows of the two-level form took 40 us)
    double s[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    long i = 0;
    for (; i + 8 <= n; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += (double)parts[(i + u) * ld + col];
    }
    for (int u = 0; i < n; ++i, ++u) s[u] += (double)parts[i * ld + col];
    const double t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    out[col] = accumulate ? out[col] + (float)t : (float)t;
}

// two-level variant for long columns: grid.y chunks -> ws, then the kernel above over ws
__global__ __launch_bounds__(256) void reduce_rows_chunk_kernel(const float* __restrict__ parts, long n, int K, long ld,
                                                                long rows_per_chunk, float* __restrict__ ws) {
    int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= K) return;
    long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
    // eight loads in flight and eight independent fp64 chains: one load per dependent iteration made this walk cost a
    // memory round trip per row (50 us for 32000 x 512 at 125 rows per chunk)
    double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    long i = r0;
    for (; i + 8 <= r1; i += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = parts[(i + u) * ld + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] += (double)v[u];
    }
    for (; i < r1; ++i) s8[0] += (double)parts[i * ld + col];
    const double s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    ws[(long)blockIdx.y * K + col] = (float)s;
}

// ---- FrameAvg / FrameMax head around logits = feat x Wfc^T (GEMM), one workgroup per clip --------------------
// logits [B][T][ldn] (first ncls columns used), bias [ncls] -> frame [B][T][ncls] = sigmoid, clip [B][ncls]
// mode 0: mean over frames; mode 1: max over frames (argmax index stored in amax [B][ncls]).
__global__ __launch_bounds__(128) void head_pool_fwd_kernel(const float* __restrict__ logits, int T, int ldn, int ncls,
                                                            const float* __restrict__ bias, int mode,
                                                            float* __restrict__ frame, float* __restrict__ clip,
                                                            int* __restrict__ amax) {
    extern __shared__ float fs[];   // [T][ncls]
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < T * ncls; i += 128) {
        int t = i / ncls, k = i % ncls;
        float s = sigmoidf_(logits[((long)b * T + t) * ldn + k] + bias[k]);
        fs[i] = s;
        frame[(long)b * T * ncls + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < ncls) {
        int k = threadIdx.x;
        if (mode == 0) {
            float s = 0.f;
            for (int t = 0; t < T; ++t) s += fs[t * ncls + k];
            clip[b * ncls + k] = s / (float)T;
        } else {
            float m = fs[k]; int am = 0;
            for (int t = 1; t < T; ++t) { float v = fs[t * ncls + k]; if (v > m) { m = v; am = t; } }
            clip[b * ncls + k] = m;
            amax[b * ncls + k] = am;
        }
    }
}

// g_logits [B][T][ldn] (pad columns zeroed) from g_clip [B][ncls]
__global__ __launch_bounds__(256) void head_pool_bwd_kernel(const float* __restrict__ g_clip,
                                                            const float* __restrict__ frame, const int* __restrict__ amax,
                                                            int B, int T, int ldn, int ncls, int mode,
                                                            float* __restrict__ g_logits) {
    // one element per thread over a flat grid (one 128-thread block per clip walked 62 elements per thread with a div / mod and
    // dependent loads each: 24 us at every batch size)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * T * ldn) return;
    const int k = (int)(i % ldn);
    const long bt = i / ldn;
    const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
    float g = 0.f;
    if (k < ncls) {
        const float s = frame[bt * ncls + k];
        const float gc = g_clip[b * ncls + k];
        const float up = (mode == 0) ? gc / (float)T : ((amax[b * ncls + k] == t) ? gc : 0.f);
        g = up * s * (1.0f - s);
    }
    g_logits[i] = g;
}

// ---- AttBlock pooling around logits = feat x [Watt;Wcla]^T.  logits columns [0,ncls) att, [ncls,2ncls) cla --------
__global__ __launch_bounds__(128) void att_pool_fwd_kernel(const float* __restrict__ logits, int T, int ldn, int ncls,
                                                           const float* __restrict__ b_att,
                                                           const float* __restrict__ b_cla, float* __restrict__ clip,
                                                           float* __restrict__ cla, float* __restrict__ norm_att,
                                                           float* __restrict__ att_sum, int activation, float inv_temp) {
    extern __shared__ float sh[];          // e [T][ncls], c [T][ncls]
    float* es = sh;
    float* cs = sh + T * ncls;
    __shared__ float ssum[NCLS_MAX];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < T * ncls; i += 128) {
        int t = i / ncls, k = i % ncls;
        const float* row = logits + ((long)b * T + t) * ldn;
        float a = row[k] + b_att[k];
        a = fminf(fmaxf(a, -10.0f), 10.0f);
        es[i] = expf(a * inv_temp) + 1e-6f;              // models.py:138-139: clamp, then exp(. / temperature) + 1e-6
        const float z = row[ncls + k] + b_cla[k];
        cs[i] = activation ? sigmoidf_(z) : z;           // nonlinear_transform (models.py:145-149): 'sigmoid' | 'linear'
    }
    __syncthreads();
    if (threadIdx.x < ncls) {
        int k = threadIdx.x;
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += es[t * ncls + k];
        ssum[k] = s;
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += (es[t * ncls + k] / s) * cs[t * ncls + k];
        clip[b * ncls + k] = acc;
        att_sum[b * ncls + k] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T * ncls; i += 128) {
        int k = i % ncls;
        cla[(long)b * T * ncls + i] = cs[i];
        norm_att[(long)b * T * ncls + i] = es[i] / ssum[k];
    }
}

__global__ __launch_bounds__(128) void att_pool_bwd_kernel(const float* __restrict__ g_clip,
                                                           const float* __restrict__ logits,
                                                           const float* __restrict__ b_att, const float* __restrict__ clip,
                                                           const float* __restrict__ cla, const float* __restrict__ norm_att,
                                                           const float* __restrict__ att_sum, int T, int ldn, int ncls,
                                                           float* __restrict__ g_logits, int activation, float inv_temp) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < T * ldn; i += 128) {
        int t = i / ldn, k = i % ldn;
        float g = 0.f;
        if (k < ncls) {
            long o = ((long)b * T + t) * ncls + k;
            float a = logits[((long)b * T + t) * ldn + k] + b_att[k];
            bool pass = (a >= -10.0f) && (a <= 10.0f);
            float n = norm_att[o], S = att_sum[b * ncls + k];
            float c = cla[o];
            g = pass ? g_clip[b * ncls + k] * (c - clip[b * ncls + k]) * (n - 1e-6f / S) * inv_temp : 0.f;
        } else if (k < 2 * ncls) {
            int kk = k - ncls;
            long o = ((long)b * T + t) * ncls + kk;
            float c = cla[o];
            g = g_clip[b * ncls + kk] * norm_att[o] * (activation ? c * (1.0f - c) : 1.0f);
        }
        g_logits[((long)b * T) * ldn + i] = g;
    }
}

// interpolate (models.py:58-69): out[b][t*ratio + r][k] = x[b][t][k]
__global__ __launch_bounds__(256) void interpolate_kernel(const float* __restrict__ x, long BT, int ncls, int ratio,
                                                          float* __restrict__ out) {
    long total = BT * ratio * ncls;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int k = (int)(i % ncls);
        long q = i / ncls;
        long bt = q / ratio;
        out[i] = x[bt * ncls + k];
    }
}

// freq-mean'd features are [B][T][C]; the reference exposes `embedding` as (B, C, T): generic 2-D transpose per batch
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int rows, int cols,
                                                        float* __restrict__ out) {
    __shared__ float tile[32][33];
    const long boff = (long)blockIdx.z * rows * cols;
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < rows && c0 + tx < cols) tile[j][tx] = x[boff + (long)(r0 + j) * cols + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) out[boff + (long)(c0 + j) * rows + r0 + tx] = tile[tx][j];
}

// ---- clip_bce (F.binary_cross_entropy, mean; log terms clamped at -100) + its gradient -------------------------
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ p, const float* __restrict__ y, long n,
                                                  float* __restrict__ loss, float* __restrict__ grad) {
    __shared__ double red[4];
    double acc = 0.0;
    const float inv_n = 1.0f / (float)n;
    for (long i = threadIdx.x; i < n; i += 256) {
        float pi = p[i], yi = y[i];
        // double-precision logs (B x 17 values: free): the fp32 library logf / log1pf compile to a packed-fp32 horizontal add
        // in the fragile operand form of tests/test_isa_audit.py, and the float rounding of an exact log is what torch computes
        float lp = fmaxf((float)log((double)pi), -100.0f), lq = fmaxf((float)log1p(-(double)pi), -100.0f);
        acc += (double)(-(yi * lp + (1.0f - yi) * lq));
        if (grad) grad[i] = (pi - yi) / fmaxf((1.0f - pi) * pi, 1e-12f) * inv_n;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n);
}

// do_mixup on a [2B][D] matrix
__global__ __launch_bounds__(256) void mixup_rows_kernel(const float* __restrict__ x, const float* __restrict__ lam,
                                                         long Bout, long D, float* __restrict__ out) {
    long total = Bout * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long b = i / D, d = i % D;
        out[i] = x[(2 * b) * D + d] * lam[2 * b] + x[(2 * b + 1) * D + d] * lam[2 * b + 1];
    }
}

// ---- Adam(amsgrad=True, weight_decay=0) over one flat buffer ----------------------------------------------------
// found-non-finite guard of the optimiser step: any NaN / inf gradient (after the all-reduce, so every rank of a
// data-parallel job sees what one rank's poisoned step produced) raises the skip flag and the host-mapped error word
// rank_flag (nullable): the word sed_guard_publish left in front of the gradient before the all-reduce -- non-zero (NaN)
// when ANY rank's split-f16 kernels met a non-finite operand this step
__global__ __launch_bounds__(256) void grad_finite_check_kernel(const float* __restrict__ g, long n, int* __restrict__ skip_flag,
                                                                int* __restrict__ err_host, const float* __restrict__ rank_flag) {
    bool bad = false;
    if (rank_flag && blockIdx.x == 0 && threadIdx.x == 0) bad = !(rank_flag[0] == 0.f);
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        bad |= !(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) < __builtin_inff()) ||
               v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; bad |= !(fabsf(v) < __builtin_inff()); }
    if (__any(bad) && (threadIdx.x & 63) == 0) {
        __hip_atomic_store(skip_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (err_host) __hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void adam_amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ vmax, long n, float step_size, float omb1,
                                                           float beta2, float omb2, float eps, float bc2_sqrt,
                                                           float grad_scale, int* __restrict__ skip_flag,
                                                           int* __restrict__ skipped, int* __restrict__ status_host) {
    // found-non-finite skip: a kernel of this step met a NaN / inf operand (skip_flag[0] != 0): leave parameters and
    // moments untouched and count the refused step -- the host learns about it without synchronising
    if (skip_flag && __hip_atomic_load(skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const int before = atomicAdd(skipped ? skipped : skip_flag + 1, 1);
            if (status_host) __hip_atomic_store(status_host, before + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (status_host && skip_flag && blockIdx.x == 0 && threadIdx.x == 0)      // cumulative refused steps as of THIS step
        __hip_atomic_store(status_host, __hip_atomic_load(skipped ? skipped : skip_flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // step_size = lr / (1 - beta1^t), omb1 = 1 - beta1, omb2 = 1 - beta2: formed in DOUBLE on the host and rounded once, as torch
    // does with its Python scalars (1.0f - 0.999f is 1.0000467e-3, not 1e-3: the second moments of rounds 1-5 ran 4.7e-5 high)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float gi = g[i] * grad_scale;
        float mi = fmaf(omb1, gi - m[i], m[i]);                         // exp_avg.lerp_(grad, 1-beta1)
        float vi = fmaf(omb2, gi * gi, v[i] * beta2);                   // mul_(beta2).addcmul_(g, g, 1-beta2)
        float vm = fmaxf(vmax[i], vi);
        float denom = sqrtf(vm) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
        m[i] = mi; v[i] = vi; vmax[i] = vm;
    }
}

// ---- GRU gates (PyTorch order r,z,n; b_hn inside r*(.)), BOTH directions of one recurrence step per launch ------
// (blockIdx.y = direction).  gi [B][3H] (row stride ld_gi) input projection incl. b_ih; gh [B][3H] hidden projection
// incl. b_hh.  h_out = (1-z)*n + z*h_prev.  Saves r,z,n and gh_n for the backward pass.
struct GruFwdP {
    const float* gi[2]; const float* gh[2]; const float* h_prev[2];
    float* h_out[2]; float* h_out2[2]; float* save[2];
    long ld_gi, ld_out, ld_out2;
    int B, Hd;
};
__global__ __launch_bounds__(256) void gru_gate_fwd_kernel(GruFwdP p) {
    const int d = blockIdx.y, Hd = p.Hd;
    const float* gi = p.gi[d]; const float* gh = p.gh[d]; const float* h_prev = p.h_prev[d];
    float* h_out = p.h_out[d]; float* h_out2 = p.h_out2[d]; float* save = p.save[d];
    long total = (long)p.B * Hd;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int b = (int)(i / Hd), j = (int)(i % Hd);
        const float* gir = gi + (long)b * p.ld_gi;
        const float* ghr = gh + (long)b * 3 * Hd;
        float r = sigmoidf_(gir[j] + ghr[j]);
        float z = sigmoidf_(gir[Hd + j] + ghr[Hd + j]);
        float ghn = ghr[2 * Hd + j];
        float n = tanhf(gir[2 * Hd + j] + r * ghn);
        float hp = h_prev ? h_prev[(long)b * Hd + j] : 0.f;
        float h = (1.0f - z) * n + z * hp;
        h_out[(long)b * p.ld_out + j] = h;
        if (h_out2) h_out2[(long)b * p.ld_out2 + j] = h;
        if (save) {
            float* s = save + (long)b * 4 * Hd;
            s[j] = r; s[Hd + j] = z; s[2 * Hd + j] = n; s[3 * Hd + j] = ghn;
        }
    }
}

// dh = g_out (row stride ld_go) + dh_direct (= dh*z of the later step, nullable) + dh_gemm (= dgh x W_hh of the later
// step, nullable).  Produces dgi [B][3H] (row stride ld_dgi), dgh [B][3H] and dh_direct_out [B][H] = dh*z.
struct GruBwdP {
    const float* g_out[2]; const float* dh_direct[2]; const float* dh_gemm[2]; const float* save[2];
    const float* h_prev[2];
    float* dgi[2]; float* dgh[2]; float* dh_direct_out[2];
    long ld_go, ld_dgi;
    int B, Hd;
};
__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(GruBwdP p) {
    const int d = blockIdx.y, Hd = p.Hd;
    const float* g_out = p.g_out[d]; const float* dh1 = p.dh_direct[d]; const float* dh2 = p.dh_gemm[d];
    const float* save = p.save[d]; const float* h_prev = p.h_prev[d];
    float* dgi = p.dgi[d]; float* dgh = p.dgh[d]; float* dh_out = p.dh_direct_out[d];
    long total = (long)p.B * Hd;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int b = (int)(i / Hd), j = (int)(i % Hd);
        const float* s = save + (long)b * 4 * Hd;
        float r = s[j], z = s[Hd + j], n = s[2 * Hd + j], ghn = s[3 * Hd + j];
        float hp = h_prev ? h_prev[(long)b * Hd + j] : 0.f;
        float dh = g_out[(long)b * p.ld_go + j] + (dh1 ? dh1[(long)b * Hd + j] : 0.f) + (dh2 ? dh2[(long)b * Hd + j] : 0.f);
        float dn = dh * (1.0f - z);
        float dz = dh * (hp - n);
        float dn_pre = dn * (1.0f - n * n);
        float dr = dn_pre * ghn;
        float dr_pre = dr * r * (1.0f - r);
        float dz_pre = dz * z * (1.0f - z);
        float* gi_o = dgi + (long)b * p.ld_dgi;
        float* gh_o = dgh + (long)b * 3 * Hd;
        gi_o[j] = dr_pre; gi_o[Hd + j] = dz_pre; gi_o[2 * Hd + j] = dn_pre;
        gh_o[j] = dr_pre; gh_o[Hd + j] = dz_pre; gh_o[2 * Hd + j] = dn_pre * r;
        dh_out[(long)b * Hd + j] = dh * z;
    }
}

// out += a   (accumulate the recurrent path into dh_prev)
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ out, const float* __restrict__ a, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] += a[i];
}

int grid_for(long n) {
    long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------------------

// out[K] (=|+=) column sums of parts [n][K] with row stride ld.  ws: scratch of 256*K floats when n > 512 (else unused).
SED_API int sed_reduce_rows(const float* parts, long n, int K, long ld, float* out, int accumulate, float* ws,
                            hipStream_t stream) {
    if (n <= 0 || K <= 0) return SED_EINVAL;
    if (n > 512 && ws) {
        long rpc = (n + 255) / 256;
        int chunks = (int)((n + rpc - 1) / rpc);
        hipLaunchKernelGGL(reduce_rows_chunk_kernel, dim3(sed_cdiv(K, 256), chunks), dim3(256), 0, stream, parts, n, K, ld, rpc, ws);
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(sed_cdiv(K, 256)), dim3(256), 0, stream, ws, (long)chunks, K, (long)K, out,
                           accumulate);
    } else {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(sed_cdiv(K, 256)), dim3(256), 0, stream, parts, n, K, ld, out, accumulate);
    }
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_head_pool_fwd(const float* logits, int B, int T, int ldn, int ncls, const float* bias, int mode,
                              float* frame, float* clip, int* amax, hipStream_t stream) {
    if (B <= 0 || ncls > NCLS_MAX || ncls > ldn || (size_t)T * ncls * 4 > 60000) return SED_EINVAL;
    hipLaunchKernelGGL(head_pool_fwd_kernel, dim3(B), dim3(128), (size_t)T * ncls * 4, stream, logits, T, ldn, ncls, bias, mode,
                       frame, clip, amax);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_head_pool_bwd(const float* g_clip, const float* frame, const int* amax, int B, int T, int ldn, int ncls,
                              int mode, float* g_logits, hipStream_t stream) {
    if (B <= 0 || ncls > ldn) return SED_EINVAL;
    hipLaunchKernelGGL(head_pool_bwd_kernel, dim3(sed_cdiv((long)B * T * ldn, 256)), dim3(256), 0, stream, g_clip, frame, amax, B, T,
                       ldn, ncls, mode, g_logits);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_att_pool_fwd(const float* logits, int B, int T, int ldn, int ncls, const float* b_att, const float* b_cla,
                             float* clip, float* cla, float* norm_att, float* att_sum, int activation, float temperature,
                             hipStream_t stream) {
    if (B <= 0 || ncls > NCLS_MAX || 2 * ncls > ldn || (size_t)T * ncls * 8 > 60000 || (activation != 0 && activation != 1) ||
        !(temperature > 0.f))
        return SED_EINVAL;
    hipLaunchKernelGGL(att_pool_fwd_kernel, dim3(B), dim3(128), (size_t)T * ncls * 8, stream, logits, T, ldn, ncls, b_att, b_cla,
                       clip, cla, norm_att, att_sum, activation, 1.0f / temperature);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_att_pool_bwd(const float* g_clip, const float* logits, const float* b_att, const float* clip,
                             const float* cla, const float* norm_att, const float* att_sum, int B, int T, int ldn,
                             int ncls, float* g_logits, int activation, float temperature, hipStream_t stream) {
    if (B <= 0 || 2 * ncls > ldn || (activation != 0 && activation != 1) || !(temperature > 0.f)) return SED_EINVAL;
    hipLaunchKernelGGL(att_pool_bwd_kernel, dim3(B), dim3(128), 0, stream, g_clip, logits, b_att, clip, cla, norm_att, att_sum,
                       T, ldn, ncls, g_logits, activation, 1.0f / temperature);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_interpolate(const float* x, long BT, int ncls, int ratio, float* out, hipStream_t stream) {
    if (BT <= 0 || ncls <= 0 || ratio <= 0) return SED_EINVAL;
    hipLaunchKernelGGL(interpolate_kernel, dim3(grid_for(BT * ratio * ncls)), dim3(256), 0, stream, x, BT, ncls, ratio, out);
    SED_LAUNCH_CHECK();
    return 0;
}

// out[b][c][r] = x[b][r][c]
SED_API int sed_transpose(const float* x, int batch, int rows, int cols, float* out, hipStream_t stream) {
    if (batch <= 0 || rows <= 0 || cols <= 0 || batch > 65535) return SED_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3(sed_cdiv(cols, 32), sed_cdiv(rows, 32), batch), dim3(256), 0, stream, x, rows, cols,
                       out);
    SED_LAUNCH_CHECK();
    return 0;
}

// loss[0] = mean BCE; grad (nullable) = d loss / d p
SED_API int sed_clip_bce(const float* p, const float* y, long n, float* loss, float* grad, hipStream_t stream) {
    if (n <= 0) return SED_EINVAL;
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, stream, p, y, n, loss, grad);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_mixup_rows(const float* x, const float* lam, long B2, long D, float* out, hipStream_t stream) {
    if (B2 <= 0 || (B2 & 1) || D <= 0) return SED_EINVAL;
    hipLaunchKernelGGL(mixup_rows_kernel, dim3(grid_for(B2 / 2 * D)), dim3(256), 0, stream, x, lam, B2 / 2, D, out);
    SED_LAUNCH_CHECK();
    return 0;
}

// flag_out[0] = NaN when this rank's found-non-finite word is set, else 0: placed next to the gradient bucket that is
// all-reduced LAST, it tells every rank of a data-parallel job that one of them met a non-finite operand this step
__global__ void guard_publish_kernel(const int* __restrict__ err_dev, float* __restrict__ flag_out) {
    flag_out[0] = __hip_atomic_load(err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? __builtin_nanf("") : 0.f;
}

SED_API int sed_guard_publish(const int* err_dev, float* flag_out, hipStream_t stream) {
    if (!err_dev || !flag_out) return SED_EINVAL;
    hipLaunchKernelGGL(guard_publish_kernel, dim3(1), dim3(1), 0, stream, err_dev, flag_out);
    SED_LAUNCH_CHECK();
    return 0;
}

// One Adam-amsgrad step (step >= 1) over flat buffers; grad_scale multiplies the gradient first (1/world_size).
SED_API int sed_adam_amsgrad(float* p, const float* g, float* m, float* v, float* vmax, long n, int step, double lr,
                             double beta1, double beta2, double eps, float grad_scale, int* skip_flag, int* skipped,
                             int* err_host, int* status_host, const float* rank_flag, hipStream_t stream) {
    if (n <= 0 || step < 1) return SED_EINVAL;
    if ((reinterpret_cast<uintptr_t>(g) & 15) != 0 && skip_flag) return SED_EINVAL;
    if (skip_flag) {
        const long nb = (n / 4 + 255) / 256;
        hipLaunchKernelGGL(grad_finite_check_kernel, dim3((unsigned)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb))), dim3(256), 0, stream, g,
                           n, skip_flag, err_host, rank_flag);
    }
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_amsgrad_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, g, m, v, vmax, n, (float)(lr / bc1),
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)sqrt(bc2), grad_scale, skip_flag,
                       skipped, status_host);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_gru_gate_fwd(const float* gi0, const float* gi1, long ld_gi, const float* gh0, const float* gh1,
                             const float* h_prev0, const float* h_prev1, int B, int Hd, float* h_out0, float* h_out1,
                             long ld_out, float* h_out2_0, float* h_out2_1, long ld_out2, float* save0, float* save1,
                             hipStream_t stream) {
    if (B <= 0 || Hd <= 0) return SED_EINVAL;
    GruFwdP p{{gi0, gi1}, {gh0, gh1}, {h_prev0, h_prev1}, {h_out0, h_out1}, {h_out2_0, h_out2_1}, {save0, save1},
              ld_gi, ld_out, ld_out2, B, Hd};
    hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3(grid_for((long)B * Hd), 2), dim3(256), 0, stream, p);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_gru_gate_bwd(const float* g_out0, const float* g_out1, long ld_go, const float* dh_direct0,
                             const float* dh_direct1, const float* dh_gemm0, const float* dh_gemm1, const float* save0,
                             const float* save1, const float* h_prev0, const float* h_prev1, int B, int Hd, float* dgi0,
                             float* dgi1, long ld_dgi, float* dgh0, float* dgh1, float* dh_direct_out0,
                             float* dh_direct_out1, hipStream_t stream) {
    if (B <= 0 || Hd <= 0) return SED_EINVAL;
    GruBwdP p{{g_out0, g_out1}, {dh_direct0, dh_direct1}, {dh_gemm0, dh_gemm1}, {save0, save1}, {h_prev0, h_prev1},
              {dgi0, dgi1}, {dgh0, dgh1}, {dh_direct_out0, dh_direct_out1}, ld_go, ld_dgi, B, Hd};
    hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(grid_for((long)B * Hd), 2), dim3(256), 0, stream, p);
    SED_LAUNCH_CHECK();
    return 0;
}

// out[i] = x[i] * s[0], s a DEVICE scalar (the upstream gradient of a 0-dim loss: clip_bce's backward)
__global__ __launch_bounds__(256) void scale_by_dev_scalar_kernel(const float* __restrict__ x, const float* __restrict__ s, long n,
                                                                  float* __restrict__ out) {
    const float k = s[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] * k;
}

SED_API int sed_scale_by_scalar(const float* x, const float* scalar_dev, long n, float* out, hipStream_t stream) {
    if (!x || !scalar_dev || !out || n <= 0) return SED_EINVAL;
    hipLaunchKernelGGL(scale_by_dev_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, scalar_dev, n, out);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_axpy(float* out, const float* a, long n, hipStream_t stream) {
    if (n <= 0) return SED_EINVAL;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, stream, out, a, n);
    SED_LAUNCH_CHECK();
    return 0;
}

// "sed-hip <abi> (gfx950) flags:<hash>": the hash identifies the hipcc flags EVERY object of this library was compiled with
// ("mixed:..." when they disagree); build.py defines it, the loader refuses a library whose flags are not the default ones
// unless SED_ALLOW_EXPERIMENT=1.
// SED_OBJECTS (common.h) lists every object of the library; tests/test_capi_and_host.py holds it equal to build.SOURCES.
#define SED_WEAK_FLAGS(name) extern "C" __attribute__((weak, visibility("hidden"))) const char sed_objflags_##name[];
SED_OBJECTS(SED_WEAK_FLAGS)
#define SED_FLAGS_ENTRY(name) sed_objflags_##name,
SED_API const char* sed_version(void) {
    static char buf[768];
    if (!buf[0]) {
        const char* objs[] = {sed_objflags_heads, SED_OBJECTS(SED_FLAGS_ENTRY)};
        bool same = true;
        for (const char* o : objs) same = same && (!o || strcmp(o, objs[0]) == 0);
        if (same) {
            snprintf(buf, sizeof buf, "sed-hip 0.4 (gfx950) flags:%s", objs[0]);
        } else {
            int n = snprintf(buf, sizeof buf, "sed-hip 0.4 (gfx950) flags:mixed");
            for (const char* o : objs)
                if (o && n < (int)sizeof buf - 40) n += snprintf(buf + n, sizeof buf - n, ":%s", o);
        }
    }
    return buf;
}

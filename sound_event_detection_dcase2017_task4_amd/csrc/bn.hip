// BatchNorm statistics / finalisation, bn0 + SpecAugment + mixup, and the BN+ReLU+avg-pool stage.
//
// Replaces (reference pytorch/models.py): `bn0` applied on the mel axis (:287-289), `spec_augmenter` (:291-292),
// `do_mixup` (:295-296, pytorch_utils.py:80-93), and inside ConvBlock (:99-115) the `bnX -> relu_ -> avg_pool2d`
// tail, plus `torch.mean(x, dim=3)` (:303).  All tensors are NHWC ([rows][C], C contiguous); all kernels are
// HBM-bound streaming kernels (float4 per lane, C/4 lanes per pixel).
//
// Statistics are carried as per-tile partials (sum, M2 = sum((x - tile_mean)^2)) in fp32 and merged in fp64
// (raw moments S1, S2 = M2 + sum^2/n), so var = S2/N - mean^2 is immune to cancellation.
#include "common.h"
#include <stdlib.h>
#include <mutex>
#include "sed_hip.h"
SED_OBJECT_FLAGS(bn)

namespace {

// ---------------------------------------------------------------------------------------------------------
// generic per-channel partial statistics of a [N][C] tensor, ROWS rows per workgroup
template <int ROWS>
__global__ __launch_bounds__(256) void chan_stats_kernel(const float* __restrict__ x, long N, int C,
                                                         float* __restrict__ partials /*[nblk][2][C]*/) {
    __shared__ float4 red_s[256], red_q[256];
    const int c4n = C >> 2;                 // float4 columns (16..128); 256 % c4n == 0
    const int rpp = 256 / c4n;              // rows per pass
    const int c4 = threadIdx.x % c4n, r0 = threadIdx.x / c4n;
    const long row_base = (long)blockIdx.x * ROWS;
    const long nrows = min((long)ROWS, N - row_base);
    const float4* xp = reinterpret_cast<const float4*>(x) + row_base * c4n;
    const float4 piv = xp[c4];              // pivot = first row of the tile (shifted sums: no cancellation)
    float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
    for (long r = r0; r < nrows; r += rpp) {
        float4 v = xp[r * c4n + c4];
        float dx = v.x - piv.x, dy = v.y - piv.y, dz = v.z - piv.z, dw = v.w - piv.w;
        s.x += dx; s.y += dy; s.z += dz; s.w += dw;
        q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
    }
    red_s[threadIdx.x] = s; red_q[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < c4n) {
        for (int j = 1; j < rpp; ++j) {
            float4 a = red_s[threadIdx.x + j * c4n], b = red_q[threadIdx.x + j * c4n];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
        }
        const float n = (float)nrows, inv = 1.0f / n;
        float4 sum = make_float4(s.x + n * piv.x, s.y + n * piv.y, s.z + n * piv.z, s.w + n * piv.w);
        float4 m2 = make_float4(fmaxf(q.x - s.x * s.x * inv, 0.f), fmaxf(q.y - s.y * s.y * inv, 0.f),
                                fmaxf(q.z - s.z * s.z * inv, 0.f), fmaxf(q.w - s.w * s.w * inv, 0.f));
        float4* po = reinterpret_cast<float4*>(partials + (long)blockIdx.x * 2 * C);
        po[c4] = sum;
        po[c4n + c4] = m2;
    }
}

// ---------------------------------------------------------------------------------------------------------
// stage 1 of the deterministic fp64 merge: [nparts][K] fp32 -> ws[nchunks][K] fp64.
// MODE 0: plain column sums.  MODE 1: K = 2C laid out [sum | M2] per part -> [S1 | S2] raw moments,
// part i holds n_i = min(rows_per_part, N - i*rows_per_part) rows (rows_per_part < 0: n_i = parts[nparts*K + i]).
template <int MODE>
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ parts, int nparts, int K,
                                                           int parts_per_chunk, long N, int rows_per_part,
                                                           double* __restrict__ ws) {
    const int col = blockIdx.y * 256 + threadIdx.x;
    if (col >= K) return;
    const int p0 = blockIdx.x * parts_per_chunk, p1 = min(nparts, p0 + parts_per_chunk);
    double acc = 0.0;
    if (MODE == 0) {
        for (int p = p0; p < p1; ++p) acc += (double)parts[(long)p * K + col];
    } else {
        const int C = K >> 1;
        if (col < C) {
            for (int p = p0; p < p1; ++p) acc += (double)parts[(long)p * K + col];
        } else {
            for (int p = p0; p < p1; ++p) {
                double n = rows_per_part < 0 ? (double)parts[(long)nparts * K + p]       // explicit per-part counts
                                             : (double)min((long)rows_per_part, N - (long)p * rows_per_part);
                if (n <= 0.0) continue;              // tile rows past the end of the tensor
                double s = (double)parts[(long)p * K + col - C];
                acc += (double)parts[(long)p * K + col] + s * s / n;
            }
        }
    }
    ws[(long)blockIdx.x * K + col] = acc;
}

// stage 2 (training): mean / invstd / folded scale,shift / running-stat update.  nn.BatchNorm2d semantics:
// biased variance normalises, unbiased variance is tracked, momentum 0.1 (reference models.py:87-88, :264).
// Both finalize kernels: one WAVE per channel sums the <= 1024 chunk rows of ws (a serial loop per channel made these
// "tiny" kernels 0.2 ms each; 16 lanes per channel still cost 25 us on the 64-channel layers: 64 dependent fp64 loads
// per lane on 4 workgroups), fixed-order xor-shuffle tree -> deterministic.
__device__ __forceinline__ void chunk_sums16(const double* __restrict__ ws, int nchunks, int C, int c, int sub,
                                             double& s1, double& s2) {
    s1 = 0.0; s2 = 0.0;
    for (int k = sub; k < nchunks; k += 64) { s1 += ws[(long)k * 2 * C + c]; s2 += ws[(long)k * 2 * C + C + c]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
}

// The same sums straight from the partials (no reduce_parts pass: one launch instead of two) for the part counts where one wave
// per channel walks them in a few trips -- blocks 3 and 4 and bn0 at the metric's batch size, every BatchNorm but block 1's at 4
// clips per GPU.  MODE as in reduce_parts_kernel.
struct BnDirectP {
    const float* parts;        // null: the sums come from ws (two-launch form)
    int nparts, rows_per_part;
};
template <int MODE>
__device__ __forceinline__ void direct_sums(const BnDirectP& dp, int C, long N, int c, int sub, double& s1, double& s2) {
    const int K = 2 * C;
    s1 = 0.0; s2 = 0.0;
    for (int p = sub; p < dp.nparts; p += 64) {
        const double a = (double)dp.parts[(long)p * K + c], b = (double)dp.parts[(long)p * K + C + c];
        if (MODE == 0) { s1 += a; s2 += b; }
        else {
            const double n = dp.rows_per_part < 0 ? (double)dp.parts[(long)dp.nparts * K + p]
                                                  : (double)min((long)dp.rows_per_part, N - (long)p * dp.rows_per_part);
            s1 += a;
            if (n > 0.0) s2 += b + a * a / n;          // (tile rows past the end of the tensor: no rows, as in reduce_parts_kernel)
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
}
// per-channel (max, min) of a tensor over its per-part ranges mm [nparts][2][C]: one wave per channel, xor-shuffle tree
__device__ __forceinline__ void range_over_parts(const float* __restrict__ mm, int nparts, int C, int c, int sub, float& mx, float& mn) {
    for (int p = sub; p < nparts; p += 64) {
        mx = fmaxf(mx, mm[((long)p * 2 + 0) * C + c]);
        mn = fminf(mn, mm[((long)p * 2 + 1) * C + c]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
}
// A/B on one box (round 5; 60 steps, twice): two launches 2.12 / 8.068 ms per step (4 clips per GPU under the HIP graph / the
// metric's bs=32), direct up to 512 parts 2.075 / 8.05, up to 2048 parts 2.09 / 8.07, up to 8192 parts 2.085 / 8.19 -- one wave per
// channel walking more than ~8 trips of strided loads is slower than the chunked pass it replaces
constexpr int SED_BN_DIRECT_MAX_PARTS = 512;

__global__ void bn_finalize_kernel(const double* __restrict__ ws, int nchunks, BnDirectP dp, int C, long N,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out,
                                   int* __restrict__ guard_dev, int* __restrict__ guard_host, float* __restrict__ cand,
                                   const float* __restrict__ y_amax, float* __restrict__ act_bound_out,
                                   const float* __restrict__ mm, int mm_parts, float* __restrict__ act_amax_out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), sub = threadIdx.x & 63;    // 256 threads = 4 channels x one wave
    if (c >= C) return;
    double s1, s2;
    if (dp.parts) direct_sums<1>(dp, C, N, c, sub, s1, s2);
    else chunk_sums16(ws, nchunks, C, c, sub, s1, s2);
    const float A = act_bound_out ? amax_read(y_amax) : 0.f;       // (every lane of the wave takes part in the read)
    // by-product (round 6; was the separate act_amax launch): the range of this channel of y over the parts (the conv epilogue's
    // per-part (max, min)), pushed through the affine + ReLU below -- the operand amax of the NEXT convolution
    float ymx = -__builtin_inff(), ymn = __builtin_inff();
    if (act_amax_out) range_over_parts(mm, mm_parts, C, c, sub, ymx, ymn);
    if (sub != 0) return;
    double mean = s1 / (double)N;
    double var = s2 / (double)N - mean * mean;
    if (var < 0.0) var = 0.0;
    double invstd = 1.0 / sqrt(var + (double)eps);
    float meanf = (float)mean, invf = (float)invstd;
    float sc = gamma[c] * invf;
    mean_out[c] = meanf; invstd_out[c] = invf;
    const float shf = fmaf(-meanf, sc, beta[c]);
    scale_out[c] = sc; shift_out[c] = shf;
    if (act_amax_out && ymx >= ymn)          // exact: the affine + ReLU is monotone in y, the extreme sits at a range end
        atomicMax(reinterpret_cast<unsigned*>(act_amax_out) + (c & (SED_AMAX_SLOTS - 1)),
                  __float_as_uint(fmaxf(bn_relu(ymx, sc, shf), bn_relu(ymn, sc, shf))));
    // by-product for the split-f16 path (was the separate act_bound kernel): this channel's bound of the pooled activation
    // relu(scale*y + shift) given |y| <= A, merged over the channels by one atomic per channel on the pre-zeroed slots
    if (act_bound_out)
        atomicMax(reinterpret_cast<unsigned*>(act_bound_out) + (c & (SED_AMAX_SLOTS - 1)),
                  __float_as_uint(fmaxf(fmaf(fabsf(sc), A, shf), 0.f) * 1.0001f));
    if (guard_dev && !(fabs(mean) < 1e300 && var < 1e300)) {
        // found-non-finite guard (the split-f16 path): batch statistics that are NaN / inf raise the error words -- the Adam
        // kernel of this step refuses its update.  (guard_dev null: torch semantics, NaN flows into the buffers.)
        __hip_atomic_store(guard_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (guard_host) __hip_atomic_store(guard_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (running_mean) {
        double unbiased = (N > 1) ? var * (double)N / (double)(N - 1) : var;
        const float nm = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        const float nv = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
        // cand: the new running statistics are only PROPOSED here; sed_bn_commit installs them at the end of the forward
        // pass unless a kernel of that pass raised the found-non-finite word -- a refused step leaves the BatchNorm buffers
        // exactly as intact as the parameters (a ReLU swallows NaN: fmaxf(NaN, 0) = 0, so the layers BEHIND a poisoned one
        // see finite, wrong statistics; only the flag knows)
        if (cand) { cand[c] = nm; cand[C + c] = nv; }
        else { running_mean[c] = nm; running_var[c] = nv; }
    }
}

// eval mode: fold running stats into scale/shift
__global__ void bn_eval_affine_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                      float* __restrict__ scale_out, float* __restrict__ shift_out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invf = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
    float sc = gamma[c] * invf;
    mean_out[c] = rm[c]; invstd_out[c] = invf; scale_out[c] = sc; shift_out[c] = fmaf(-rm[c], sc, beta[c]);
}

// stage 2 (backward): dbeta = sum dy, dgamma = sum dy*xhat; coefficients of g_y = a*dy + b*y + c.
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ ws, int nchunks, BnDirectP dp, int C, long N,
                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                       const float* __restrict__ scale, int batch_stats,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef /*[3][C]*/, const float* __restrict__ y_amax,
                                       const float* __restrict__ g_amax, float ginv, float* __restrict__ bound_out,
                                       const float* __restrict__ mm, int mm_parts) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), sub = threadIdx.x & 63;
    if (c >= C) return;
    double s1, s2;
    if (dp.parts) direct_sums<0>(dp, C, N, c, sub, s1, s2);
    else chunk_sums16(ws, nchunks, C, c, sub, s1, s2);
    const float A = (bound_out && !mm) ? amax_read(y_amax) : 0.f, G = bound_out ? amax_read(g_amax) * ginv : 0.f;
    // mm given (round 6; was the separate grad_bound launch): |y| ranges over this channel's own (min, max) instead of +-A
    float ymx = A, ymn = -A;
    if (bound_out && mm) { ymx = -__builtin_inff(); ymn = __builtin_inff(); range_over_parts(mm, mm_parts, C, c, sub, ymx, ymn); }
    if (sub != 0) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    if (coef) {
        double a = (double)scale[c];
        double b = 0.0, cc = 0.0;                  // eval mode (running statistics): BN is a fixed affine map
        if (batch_stats) {
            b = -a * (double)invstd[c] * s2 / (double)N;
            cc = -a * s1 / (double)N - b * (double)mean[c];
        }
        const float af = (float)a, bf = (float)b, cf = (float)cc;
        coef[c] = af; coef[C + c] = bf; coef[2 * C + c] = cf;
        // by-product (was grad_bound_kernel with the y_amax range): bound of |a*dy + b*y + c| for |dy| <= G, |y| <= A
        if (bound_out && ymx >= ymn)
            atomicMax(reinterpret_cast<unsigned*>(bound_out) + (c & (SED_AMAX_SLOTS - 1)),
                      __float_as_uint((fabsf(af) * G + fmaxf(fabsf(fmaf(bf, ymx, cf)), fabsf(fmaf(bf, ymn, cf)))) * 1.0001f));
    }
}

// ---------------------------------------------------------------------------------------------------------
// bn0 (folded) + SpecAugment stripes + mixup, forward.  logmel [B2][T][64] -> x0 [Bout][T][64]
// stripes [B2][8] = {tb0,td0,tb1,td1,fb0,fd0,fb1,fd1} or null; lam [B2] or null (then Bout = B2).
__device__ __forceinline__ bool specaug_keep(const int* st, int t, int m) {
    bool drop = ((unsigned)(t - st[0]) < (unsigned)st[1]) | ((unsigned)(t - st[2]) < (unsigned)st[3]) |
                ((unsigned)(m - st[4]) < (unsigned)st[5]) | ((unsigned)(m - st[6]) < (unsigned)st[7]);
    return !drop;
}

__global__ __launch_bounds__(256) void bn0_aug_mix_fwd_kernel(const float* __restrict__ lm, int B2, int T,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const int* __restrict__ stripes,
                                                              const float* __restrict__ lam, float* __restrict__ out) {
    const long total4 = (long)(lam ? B2 / 2 : B2) * T * 16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        int m4 = (int)(i & 15);
        long bt = i >> 4;
        int t = (int)(bt % T);
        int bo = (int)(bt / T);
        float4 sc = reinterpret_cast<const float4*>(scale)[m4], sh = reinterpret_cast<const float4*>(shift)[m4];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int nsrc = lam ? 2 : 1;
        for (int j = 0; j < nsrc; ++j) {
            int n = lam ? 2 * bo + j : bo;
            float4 v = reinterpret_cast<const float4*>(lm)[((long)n * T + t) * 16 + m4];
            float l = lam ? lam[n] : 1.0f;
            float y[4] = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
            int st[8];
            if (stripes) {
#pragma unroll
                for (int k = 0; k < 8; ++k) st[k] = stripes[n * 8 + k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bool keep = stripes ? specaug_keep(st, t, m4 * 4 + k) : true;
                float yk = keep ? y[k] : 0.0f;
                acc[k] = lam ? fmaf(l, yk, acc[k]) : yk;
            }
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// backward of the same stage: only dgamma0/dbeta0 are needed (the waveform takes no gradient).
// partials [nblk][2][64]: sum dy, sum dy*xhat with dy = lam*keep*g.
__global__ __launch_bounds__(256) void bn0_aug_mix_bwd_kernel(const float* __restrict__ lm,
                                                              const float* __restrict__ g /*[Bout][T][64]*/, int B2,
                                                              int T, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const int* __restrict__ stripes,
                                                              const float* __restrict__ lam, int rows_per_block,
                                                              float* __restrict__ partials) {
    __shared__ float4 red_a[256], red_b[256];
    const int m4 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    const long nrows = (long)B2 * T;
    const long row_base = (long)blockIdx.x * rows_per_block;
    const long row_end = min(nrows, row_base + rows_per_block);
    float4 mu = reinterpret_cast<const float4*>(mean)[m4], is = reinterpret_cast<const float4*>(invstd)[m4];
    float sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
    // (n, t) of this thread's first row by ONE 64-bit division, then carried along the walk (a div / mod pair per row and eight
    // stripe loads per row made this kernel 57 us at batch 32 and 67 us at batch 256: serial work, not bytes)
    long r = row_base + r0;
    int n = (int)(r / T), t = (int)(r - (long)n * T);
    int n_loaded = -1;
    int st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float l = 1.0f;
    for (; r < row_end; r += 16) {
        if (n != n_loaded) {
            n_loaded = n;
            l = lam ? lam[n] : 1.0f;
            if (stripes) {
#pragma unroll
                for (int k = 0; k < 8; ++k) st[k] = stripes[n * 8 + k];
            }
        }
        const int bo = lam ? n >> 1 : n;
        float4 v = reinterpret_cast<const float4*>(lm)[r * 16 + m4];
        float4 gv = reinterpret_cast<const float4*>(g)[((long)bo * T + t) * 16 + m4];
        float xv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
        float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool keep = stripes ? specaug_keep(st, t, m4 * 4 + k) : true;
            float dy = keep ? l * gg[k] : 0.0f;
            sa[k] += dy;
            sb[k] = fmaf(dy, (xv[k] - muv[k]) * isv[k], sb[k]);
        }
        t += 16;
        while (t >= T) { t -= T; ++n; }
    }
    red_a[threadIdx.x] = make_float4(sa[0], sa[1], sa[2], sa[3]);
    red_b[threadIdx.x] = make_float4(sb[0], sb[1], sb[2], sb[3]);
    __syncthreads();
    if (threadIdx.x < 16) {
        float4 a = red_a[threadIdx.x], b = red_b[threadIdx.x];
        for (int j = 1; j < 16; ++j) {
            float4 a2 = red_a[threadIdx.x + 16 * j], b2 = red_b[threadIdx.x + 16 * j];
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
            b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        float4* po = reinterpret_cast<float4*>(partials + (long)blockIdx.x * 128);
        po[threadIdx.x] = a;
        po[16 + threadIdx.x] = b;
    }
}

// block-wide "some |gamma[c]| < gmin" (every block evaluates it redundantly: C <= 512 loads and one barrier)
__device__ __forceinline__ bool any_small_gamma(const float* __restrict__ gamma, int C, float gmin) {
    bool small = false;
    for (int c = threadIdx.x; c < C; c += 256) small |= fabsf(gamma[c]) < gmin;
    return __syncthreads_or(small ? 1 : 0) != 0;
}

// ---------------------------------------------------------------------------------------------------------
// BN(folded)+ReLU+avg-pool forward.  y [B][H][W][C] -> out [B][H/ph][W/pw][C]  (floor mode: trailing rows dropped)
// CNT: also write, per pooled element, how many of its ph*pw inputs passed the ReLU (one byte per channel, packed 4 to a
// word like the float4 lanes) -- with the pooled output itself that is all backward pass 1 needs (see
// pool_bwd_reduce_win_kernel).
// MODE: 0 = avg_pool2d (every model of the reference), 1 = max_pool2d, 2 = avg + max (models.py:104-111).
template <bool CNT, int MODE = 0>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float* __restrict__ y, int B, int H, int W, int C,
                                                               int ph, int pw, const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               float* __restrict__ out, unsigned* __restrict__ cnt4,
                                                               float* __restrict__ amax_out,
                                                               const float* __restrict__ pair_bound = nullptr) {
    const int Ho = H / ph, Wo = W / pw, c4n = C >> 2;
    const long total = (long)B * Ho * Wo * c4n;
    const float inv = 1.0f / (float)(ph * pw);
    float amax = 0.f;
    // pair_bound: the pooled tensor is written as split-f16 operand pairs (the next block's conv1 and its weight gradient copy
    // them into LDS), scaled by the power of two of that upper bound of its amax
    const float so = pair_bound ? sed_sf_scale_of(amax_read(pair_bound)) : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4 = (int)(i % c4n);
        long p = i / c4n;
        int wo = (int)(p % Wo);
        long q = p / Wo;
        int ho = (int)(q % Ho);
        int b = (int)(q / Ho);
        float4 sc = reinterpret_cast<const float4*>(scale)[c4], sh = reinterpret_cast<const float4*>(shift)[c4];
        float4 acc = make_float4(0, 0, 0, 0), mx = make_float4(0, 0, 0, 0);       // relu(.) >= 0: 0 is the identity of max
        unsigned n4 = 0;
        for (int dh = 0; dh < ph; ++dh)
            for (int dw = 0; dw < pw; ++dw) {
                long src = (((long)b * H + ho * ph + dh) * W + wo * pw + dw) * c4n + c4;
                float4 v = reinterpret_cast<const float4*>(y)[src];
                const float4 a = make_float4(bn_relu(v.x, sc.x, sh.x), bn_relu(v.y, sc.y, sh.y), bn_relu(v.z, sc.z, sh.z),
                                             bn_relu(v.w, sc.w, sh.w));
                acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
                if (MODE != 0) { mx.x = fmaxf(mx.x, a.x); mx.y = fmaxf(mx.y, a.y); mx.z = fmaxf(mx.z, a.z); mx.w = fmaxf(mx.w, a.w); }
                if (CNT)
                    n4 += (bn_relu_active(v.x, sc.x, sh.x) ? 1u : 0u) + (bn_relu_active(v.y, sc.y, sh.y) ? 0x100u : 0u) +
                          (bn_relu_active(v.z, sc.z, sh.z) ? 0x10000u : 0u) + (bn_relu_active(v.w, sc.w, sh.w) ? 0x1000000u : 0u);
            }
        float4 o = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        if (MODE == 1) o = mx;
        if (MODE == 2) { o.x += mx.x; o.y += mx.y; o.z += mx.z; o.w += mx.w; }
        if (pair_bound) sed_store_pairs4(out, i, make_float4(o.x * so, o.y * so, o.z * so, o.w * so));
        else store_nt4(out, i, o);
        amax = fmaxf(fmaxf(amax, fmaxf(o.x, o.y)), fmaxf(o.z, o.w));          // o >= 0
        if (CNT) cnt4[i] = n4;
    }
    if (amax_out) amax_publish_block(amax_out, amax);   // max of the pooled tensor = the split-f16 scale of the next block's conv1
}

// amax of relu(scale*y + shift) over a tensor that is never materialised, from per-part per-channel (max, min) of y
// [nparts][2][C] (conv epilogues): the affine + ReLU is monotone in y, so the extreme of every channel is taken at one of
// its two range ends -- exact, no pass over the tensor.
__global__ __launch_bounds__(256) void act_amax_kernel(const float* __restrict__ mm, int nparts, int C,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int parts_per_block, float* __restrict__ amax_out) {
    __shared__ float red[256];
    const int p0 = blockIdx.x * parts_per_block, p1 = min(nparts, p0 + parts_per_block);
    float best = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float mx = -__builtin_inff(), mn = __builtin_inff();
        for (int q = p0; q < p1; ++q) {
            mx = fmaxf(mx, mm[((long)q * 2 + 0) * C + c]);
            mn = fminf(mn, mm[((long)q * 2 + 1) * C + c]);
        }
        if (mx >= mn) {
            const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
            best = fmaxf(best, scale ? fmaxf(bn_relu(mx, sc, sh), bn_relu(mn, sc, sh)) : fmaxf(fabsf(mx), fabsf(mn)));
        }
    }
    red[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned*>(amax_out) + (blockIdx.x & (SED_AMAX_SLOTS - 1)), __float_as_uint(red[0]));
}

// the same amax by a pass over y [nrows][C] (producers that leave no range partials)
__global__ __launch_bounds__(256) void act_amax_full_kernel(const float* __restrict__ y, long nrows, int C,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ amax_out) {
    const int c4n = C >> 2;
    const long total = nrows * c4n;
    float amax = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        const float4 sc = reinterpret_cast<const float4*>(scale)[c4], sh = reinterpret_cast<const float4*>(shift)[c4];
        amax = fmaxf(fmaxf(amax, fmaxf(bn_relu(v.x, sc.x, sh.x), bn_relu(v.y, sc.y, sh.y))),
                     fmaxf(bn_relu(v.z, sc.z, sh.z), bn_relu(v.w, sc.w, sh.w)));
    }
    amax_publish_block(amax_out, amax);
}

// backward pass 1 / pass 2 of the same stage.  dy = g_out[pooled pos]/(ph*pw) * relu-mask (0 on dropped rows).
// PASS 1: per-block partial sums (sum dy, sum dy*xhat) -> partials[nblk][2][C]
// PASS 2: g_y = a*dy + b*y + c  -> gy [B][H][W][C]
// MODE (0 avg / 1 max / 2 avg + max): what reaches an input element from the pooled gradient g: g/n (avg), g if the element is the
// FIRST maximum of its window in scan order (torch's max_pool2d keeps the first of equal maxima), or both.
template <int PASS, int MODE = 0>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_kernel(const float* __restrict__ y,
                                                               const float* __restrict__ gout, int B, int H, int W,
                                                               int C, int ph, int pw, const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ coef, int rows_per_block,
                                                               float* __restrict__ partials, float* __restrict__ gy,
                                                               const float* __restrict__ gamma, float gmin,
                                                               float* __restrict__ amax_out,
                                                               const float* __restrict__ pair_bound = nullptr) {
    __shared__ float4 red_a[256], red_b[256];
    float amax = 0.f;
    // PASS 2 with pair_bound: gy is written as split-f16 operand pairs scaled by the power of two of that BOUND of its amax
    const float sg = (PASS == 2 && pair_bound) ? sed_sf_scale_of(amax_read(pair_bound)) : 1.f;
    // PASS 1 with a gamma pointer is the exact fallback of the windowed pass: it runs only when that one declines
    if (PASS == 1 && gamma && !any_small_gamma(gamma, C, gmin)) return;
    const int Ho = H / ph, Wo = W / pw, c4n = C >> 2;
    const int rpp = 256 / c4n;
    const int c4 = threadIdx.x % c4n, r0 = threadIdx.x / c4n;
    const long nrows = (long)B * H * W;
    const long row_base = (long)blockIdx.x * rows_per_block;
    const long row_end = min(nrows, row_base + rows_per_block);
    const float inv = 1.0f / (float)(ph * pw);
    const float4 sc = reinterpret_cast<const float4*>(scale)[c4], sh = reinterpret_cast<const float4*>(shift)[c4];
    float4 mu, is, ca, cb, cc;
    if (PASS == 1) {
        mu = reinterpret_cast<const float4*>(mean)[c4]; is = reinterpret_cast<const float4*>(invstd)[c4];
    } else {
        ca = reinterpret_cast<const float4*>(coef)[c4]; cb = reinterpret_cast<const float4*>(coef)[c4n + c4];
        cc = reinterpret_cast<const float4*>(coef)[2 * c4n + c4];
    }
    float4 sa = make_float4(0, 0, 0, 0), sb = make_float4(0, 0, 0, 0);
    for (long r = row_base + r0; r < row_end; r += rpp) {
        int w = (int)(r % W);
        long q = r / W;
        int h = (int)(q % H);
        int b = (int)(q / H);
        int ho = h / ph, wo = w / pw;
        float4 v = load_nt4(y, r * c4n + c4);
        float4 g = make_float4(0, 0, 0, 0);
        if (ho < Ho && wo < Wo) g = reinterpret_cast<const float4*>(gout)[(((long)b * Ho + ho) * Wo + wo) * c4n + c4];
        float4 wgt = make_float4(inv, inv, inv, inv);       // share of the pooled gradient this element receives
        if (MODE != 0 && ho < Ho && wo < Wo) {
            const float4 self = make_float4(bn_relu(v.x, sc.x, sh.x), bn_relu(v.y, sc.y, sh.y), bn_relu(v.z, sc.z, sh.z),
                                            bn_relu(v.w, sc.w, sh.w));
            // first maximum of the window: no earlier element (scan order) is >= self, no later element is > self
            bool fx = true, fy = true, fz = true, fw = true;
            const int my = (h - ho * ph) * pw + (w - wo * pw);
            for (int dh = 0; dh < ph; ++dh)
                for (int dw = 0; dw < pw; ++dw) {
                    const int k = dh * pw + dw;
                    if (k == my) continue;
                    const float4 u = reinterpret_cast<const float4*>(y)[(((long)b * H + ho * ph + dh) * W + wo * pw + dw) * c4n + c4];
                    const float4 a = make_float4(bn_relu(u.x, sc.x, sh.x), bn_relu(u.y, sc.y, sh.y), bn_relu(u.z, sc.z, sh.z),
                                                 bn_relu(u.w, sc.w, sh.w));
                    if (k < my) { fx &= a.x < self.x; fy &= a.y < self.y; fz &= a.z < self.z; fw &= a.w < self.w; }
                    else { fx &= a.x <= self.x; fy &= a.y <= self.y; fz &= a.z <= self.z; fw &= a.w <= self.w; }
                }
            const float base = MODE == 2 ? inv : 0.f;
            wgt = make_float4(base + (fx ? 1.f : 0.f), base + (fy ? 1.f : 0.f), base + (fz ? 1.f : 0.f), base + (fw ? 1.f : 0.f));
        }
        float4 dy;
        dy.x = bn_relu_active(v.x, sc.x, sh.x) ? g.x * wgt.x : 0.f;
        dy.y = bn_relu_active(v.y, sc.y, sh.y) ? g.y * wgt.y : 0.f;
        dy.z = bn_relu_active(v.z, sc.z, sh.z) ? g.z * wgt.z : 0.f;
        dy.w = bn_relu_active(v.w, sc.w, sh.w) ? g.w * wgt.w : 0.f;
        if (PASS == 1) {
            sa.x += dy.x; sa.y += dy.y; sa.z += dy.z; sa.w += dy.w;
            sb.x = fmaf(dy.x, (v.x - mu.x) * is.x, sb.x); sb.y = fmaf(dy.y, (v.y - mu.y) * is.y, sb.y);
            sb.z = fmaf(dy.z, (v.z - mu.z) * is.z, sb.z); sb.w = fmaf(dy.w, (v.w - mu.w) * is.w, sb.w);
        } else {
            float4 o;
            o.x = fmaf(ca.x, dy.x, fmaf(cb.x, v.x, cc.x)); o.y = fmaf(ca.y, dy.y, fmaf(cb.y, v.y, cc.y));
            o.z = fmaf(ca.z, dy.z, fmaf(cb.z, v.z, cc.z)); o.w = fmaf(ca.w, dy.w, fmaf(cb.w, v.w, cc.w));
            if (pair_bound) sed_store_pairs4(gy, r * c4n + c4, make_float4(o.x * sg, o.y * sg, o.z * sg, o.w * sg));
            else store_nt4(gy, r * c4n + c4, o);
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (PASS == 2 && amax_out) {        // amax of the tensor just written, for the split-f16 consumers (conv_sf16.hip)
        amax_publish_block(amax_out, amax);
    }
    if (PASS == 1) {
        red_a[threadIdx.x] = sa; red_b[threadIdx.x] = sb;
        __syncthreads();
        if (threadIdx.x < c4n) {
            for (int j = 1; j < rpp; ++j) {
                float4 a2 = red_a[threadIdx.x + j * c4n], b2 = red_b[threadIdx.x + j * c4n];
                sa.x += a2.x; sa.y += a2.y; sa.z += a2.z; sa.w += a2.w;
                sb.x += b2.x; sb.y += b2.y; sb.z += b2.z; sb.w += b2.w;
            }
            float4* po = reinterpret_cast<float4*>(partials + (long)blockIdx.x * 2 * C);
            po[c4] = sa;
            po[c4n + c4] = sb;
        }
    }
}

// Backward pass 1 at POOLED resolution.  The pool gradient is constant over a window, so with n = ph*pw,
// cnt = number of active inputs of the window and p = the pooled output (n*p = sum over the window of mask*(gamma*xhat +
// beta)):   sum dy = sum_windows g*cnt/n,   sum dy*xhat = sum_windows g*(p - beta*cnt/n)/gamma.
// Reads g, p and one byte per element instead of the full-resolution y: 0.52 instead of 1.25 tensor sizes for 2x2.
// The division by gamma amplifies p's rounding by 1/|gamma|: callers use the exact full-resolution pass when any
// |gamma| is small (ops.py keeps a per-layer guard).
__global__ __launch_bounds__(256) void pool_bwd_reduce_win_kernel(const float* __restrict__ gout,
                                                                  const float* __restrict__ pooled,
                                                                  const unsigned* __restrict__ cnt4, long nrows, int C,
                                                                  float inv, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, int rows_per_block,
                                                                  float* __restrict__ partials, float gmin,
                                                                  const float* __restrict__ pair_bound = nullptr) {
    __shared__ float4 red_a[256], red_b[256];
    if (gmin > 0.f && any_small_gamma(gamma, C, gmin)) return;     // the exact pass takes over (partials pre-zeroed)
    const float pinv = pair_bound ? 1.0f / sed_sf_scale_of(amax_read(pair_bound)) : 1.f;     // pooled stored as pairs: decode
    const int c4n = C >> 2;
    const int rpp = 256 / c4n;
    const int c4 = threadIdx.x % c4n, r0 = threadIdx.x / c4n;
    const long row_base = (long)blockIdx.x * rows_per_block;
    const long row_end = min(nrows, row_base + rows_per_block);
    const float4 ga = reinterpret_cast<const float4*>(gamma)[c4], be = reinterpret_cast<const float4*>(beta)[c4];
    const float4 gi = make_float4(1.0f / ga.x, 1.0f / ga.y, 1.0f / ga.z, 1.0f / ga.w);
    float4 sa = make_float4(0, 0, 0, 0), sb = make_float4(0, 0, 0, 0);
    for (long r = row_base + r0; r < row_end; r += rpp) {
        const float4 g = reinterpret_cast<const float4*>(gout)[r * c4n + c4];
        float4 p;
        if (pair_bound) {
            p = sed_load_pairs4(pooled, r * c4n + c4);
            p.x *= pinv; p.y *= pinv; p.z *= pinv; p.w *= pinv;
        } else {
            p = reinterpret_cast<const float4*>(pooled)[r * c4n + c4];
        }
        const unsigned n4 = cnt4[r * c4n + c4];
        float4 gc;
        gc.x = g.x * ((float)(n4 & 255u) * inv); gc.y = g.y * ((float)((n4 >> 8) & 255u) * inv);
        gc.z = g.z * ((float)((n4 >> 16) & 255u) * inv); gc.w = g.w * ((float)(n4 >> 24) * inv);
        sa.x += gc.x; sa.y += gc.y; sa.z += gc.z; sa.w += gc.w;
        sb.x = fmaf(fmaf(g.x, p.x, -gc.x * be.x), gi.x, sb.x); sb.y = fmaf(fmaf(g.y, p.y, -gc.y * be.y), gi.y, sb.y);
        sb.z = fmaf(fmaf(g.z, p.z, -gc.z * be.z), gi.z, sb.z); sb.w = fmaf(fmaf(g.w, p.w, -gc.w * be.w), gi.w, sb.w);
    }
    red_a[threadIdx.x] = sa; red_b[threadIdx.x] = sb;
    __syncthreads();
    if (threadIdx.x < c4n) {
        for (int j = 1; j < rpp; ++j) {
            float4 a2 = red_a[threadIdx.x + j * c4n], b2 = red_b[threadIdx.x + j * c4n];
            sa.x += a2.x; sa.y += a2.y; sa.z += a2.z; sa.w += a2.w;
            sb.x += b2.x; sb.y += b2.y; sb.z += b2.z; sb.w += b2.w;
        }
        float4* po = reinterpret_cast<float4*>(partials + (long)blockIdx.x * 2 * C);
        po[c4] = sa;
        po[c4n + c4] = sb;
    }
}

// g_y = a*dy + b*y + c, in place on dy (the conv1-side BN backward, dy produced by the dgrad epilogue)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ dy, const float* __restrict__ y,
                                                           long nrows, int C, const float* __restrict__ coef,
                                                           float* __restrict__ amax_out, const float* __restrict__ pair_bound) {
    const int c4n = C >> 2;
    const long total = nrows * c4n;
    float amax = 0.f;
    const float sg = pair_bound ? sed_sf_scale_of(amax_read(pair_bound)) : 1.f;     // pairs out: see bn_relu_pool_bwd_kernel
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4 = (int)(i % c4n);
        float4 ca = reinterpret_cast<const float4*>(coef)[c4], cb = reinterpret_cast<const float4*>(coef)[c4n + c4],
               cc = reinterpret_cast<const float4*>(coef)[2 * c4n + c4];
        float4 d = reinterpret_cast<float4*>(dy)[i], v = reinterpret_cast<const float4*>(y)[i];
        float4 o;
        o.x = fmaf(ca.x, d.x, fmaf(cb.x, v.x, cc.x)); o.y = fmaf(ca.y, d.y, fmaf(cb.y, v.y, cc.y));
        o.z = fmaf(ca.z, d.z, fmaf(cb.z, v.z, cc.z)); o.w = fmaf(ca.w, d.w, fmaf(cb.w, v.w, cc.w));
        if (pair_bound) sed_store_pairs4(dy, i, make_float4(o.x * sg, o.y * sg, o.z * sg, o.w * sg));
        else store_nt4(dy, i, o);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (amax_out) {
        amax_publish_block(amax_out, amax);
    }
}

}  // namespace

// Address ranges whose amax buffers arrive zeroed (slices of a pool the caller zeroes with one fill): the entry points skip
// their 256-byte memset launches (33 per training step) for pointers inside such a range ONLY -- a per-buffer property, not a
// process-wide mode: every other caller of the library in the same process keeps the self-zeroing contract.
namespace {
struct PrezeroRange { const float* lo; const float* hi; };
constexpr int SED_MAX_PREZERO = 64;
PrezeroRange prezero_ranges__[SED_MAX_PREZERO];
int n_prezero__ = 0;
std::mutex prezero_mu__;
}  // namespace
bool sed_amax_is_prezeroed__(const float* p) {
    if (n_prezero__ == 0) return false;
    std::lock_guard<std::mutex> g(prezero_mu__);
    for (int i = 0; i < n_prezero__; ++i)
        if (p >= prezero_ranges__[i].lo && p < prezero_ranges__[i].hi) return true;
    return false;
}
SED_API int sed_amax_prezeroed_range(const float* base, long nfloats, int on) {
    if (!base || nfloats <= 0) return SED_EINVAL;
    std::lock_guard<std::mutex> g(prezero_mu__);
    for (int i = 0; i < n_prezero__; ++i)
        if (prezero_ranges__[i].lo == base) {              // re-register / unregister
            prezero_ranges__[i] = prezero_ranges__[--n_prezero__];
            break;
        }
    if (on) {
        if (n_prezero__ == SED_MAX_PREZERO) return SED_EINVAL;
        prezero_ranges__[n_prezero__++] = PrezeroRange{base, base + nfloats};
    }
    return 0;
}
SED_API int sed_amax_slots(void) { return SED_AMAX_SLOTS; }

namespace {

int stream_grid(long work_items) {
    long blocks = (work_items + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int reduce_chunks(int nparts) {          // parts per chunk: <= 1024 chunks (ws holds 1024 * 2C doubles)
    int ppc = sed_cdiv(nparts, 1024);
    if (ppc < 8) ppc = 8;
    return ppc;
}

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------------

// Partial statistics of x [N][C] in tiles of sed_stats_rows_per_part() rows.  partials must hold ceil(N/rows)*2*C floats.
SED_API int sed_chan_stats(const float* x, long N, int C, float* partials, hipStream_t stream) {
    if (N <= 0 || C < 64 || C > 512 || (256 % (C / 4)) != 0) return SED_EINVAL;
    // (256-row parts would fill the chip at the metric's batch size -- 63 workgroups today, 21 -> 8 us -- but re-partitioning
    // bn0's statistics moves their last bits, and with them a ReLU flip somewhere downstream: the bn0.weight gradient of the
    // 6-clip fixtures went from < 2e-3 to 2.7e-3 of float64.  Not worth 0.15 % of the step.)
    hipLaunchKernelGGL(chan_stats_kernel<1024>, dim3(sed_cdiv(N, 1024)), dim3(256), 0, stream, x, N, C, partials);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_stats_rows_per_part(void) { return 1024; }

// ws: at least 1024*2*C doubles.
SED_API int sed_bn_finalize(const float* partials, int nparts, int rows_per_part, long N, int C, const float* gamma,
                            const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                            float* mean_out, float* invstd_out, float* scale_out, float* shift_out, double* ws,
                            int* guard_dev, int* guard_host, float* cand, const float* y_amax, float* act_bound_out,
                            const float* minmax, float* act_amax_out, hipStream_t stream) {
    if (nparts <= 0 || C <= 0 || N <= 0 || (act_bound_out && !y_amax) || (act_amax_out && !minmax)) return SED_EINVAL;
    if (act_bound_out) {
        hipError_t e = sed_amax_clear(act_bound_out, stream);
        if (e != hipSuccess) return (int)e;
    }
    if (act_amax_out) {
        hipError_t e = sed_amax_clear(act_amax_out, stream);
        if (e != hipSuccess) return (int)e;
    }
    int ppc = reduce_chunks(nparts), nchunks = sed_cdiv(nparts, ppc), K = 2 * C;
    const bool direct = nparts <= SED_BN_DIRECT_MAX_PARTS;
    const BnDirectP dp{direct ? partials : nullptr, nparts, rows_per_part};
    if (!direct)
        hipLaunchKernelGGL(reduce_parts_kernel<1>, dim3(nchunks, sed_cdiv(K, 256)), dim3(256), 0, stream, partials, nparts, K,
                           ppc, N, rows_per_part, ws);
    // the operand amax of the next convolution rides on the finalize launch while one wave per channel walks the parts in a few
    // trips (the same limit as the direct sums); beyond it the chunked act_amax launch follows as before
    const bool fold = act_amax_out && direct;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(sed_cdiv(C, 4)), dim3(256), 0, stream, ws, nchunks, dp, C, N, gamma, beta, eps,
                       momentum, running_mean, running_var, mean_out, invstd_out, scale_out, shift_out, guard_dev, guard_host, cand,
                       y_amax, act_bound_out, fold ? minmax : nullptr, nparts, fold ? act_amax_out : nullptr);
    if (act_amax_out && !fold) {
        int ppb = sed_cdiv(nparts, 1024);
        if (ppb < 8) ppb = 8;
        hipLaunchKernelGGL(act_amax_kernel, dim3(sed_cdiv(nparts, ppb)), dim3(256), 0, stream, minmax, nparts, C,
                           (const float*)scale_out, (const float*)shift_out, ppb, act_amax_out);
    }
    SED_LAUNCH_CHECK();
    return 0;
}

namespace {
constexpr int SED_BN_COMMIT_MAX = 16;
struct BnCommitP {
    float* cand[SED_BN_COMMIT_MAX];
    float* rm[SED_BN_COMMIT_MAX];
    float* rv[SED_BN_COMMIT_MAX];
    int C[SED_BN_COMMIT_MAX];
};
// RESTORE = false (end of the forward pass): install the proposed statistics and keep the ones they replace in `cand` --
// unless the found-non-finite word is already up: then nothing is installed and `cand` takes a copy of the untouched buffers.
// RESTORE = true (behind the optimiser step, whose finite check has folded in the backward pass, the all-reduced gradient and
// the other ranks' flags): a refused step puts the old statistics back.  Either way `cand` holds "before this step".
template <bool RESTORE>
__global__ __launch_bounds__(256) void bn_commit_kernel(BnCommitP p, const int* __restrict__ guard_dev) {
    const bool up = guard_dev && __hip_atomic_load(guard_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (RESTORE && !up) return;
    const int t = blockIdx.y, C = p.C[t];
    for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
        const float om = p.rm[t][c], ov = p.rv[t][c];
        if (RESTORE) { p.rm[t][c] = p.cand[t][c]; p.rv[t][c] = p.cand[t][C + c]; }
        else {
            if (!up) { p.rm[t][c] = p.cand[t][c]; p.rv[t][c] = p.cand[t][C + c]; }
            p.cand[t][c] = om; p.cand[t][C + c] = ov;
        }
    }
}

int bn_commit_launch(bool restore, int n, float* const* cand, float* const* running_mean, float* const* running_var, const int* C,
                     const int* guard_dev, hipStream_t stream) {
    if (n <= 0 || n > SED_BN_COMMIT_MAX || !cand || !running_mean || !running_var || !C || (restore && !guard_dev)) return SED_EINVAL;
    BnCommitP p;
    int most = 0;
    for (int t = 0; t < SED_BN_COMMIT_MAX; ++t) {
        const int s = t < n ? t : 0;
        if (!cand[s] || !running_mean[s] || !running_var[s] || C[s] <= 0) return SED_EINVAL;
        p.cand[t] = cand[s]; p.rm[t] = running_mean[s]; p.rv[t] = running_var[s]; p.C[t] = C[s];
        if (t < n && C[s] > most) most = C[s];
    }
    if (restore) hipLaunchKernelGGL(bn_commit_kernel<true>, dim3(sed_cdiv(most, 256), n), dim3(256), 0, stream, p, guard_dev);
    else hipLaunchKernelGGL(bn_commit_kernel<false>, dim3(sed_cdiv(most, 256), n), dim3(256), 0, stream, p, guard_dev);
    SED_LAUNCH_CHECK();
    return 0;
}
}  // namespace

// Install the running statistics proposed by n <= 16 sed_bn_finalize(..., cand) calls -- unless *guard_dev != 0 -- and leave
// the statistics from before the step in `cand`; sed_bn_restore puts those back when *guard_dev != 0 (a step refused later:
// non-finite backward pass, non-finite all-reduced gradient, another rank's flag).
SED_API int sed_bn_commit(int n, float* const* cand, float* const* running_mean, float* const* running_var, const int* C,
                          const int* guard_dev, hipStream_t stream) {
    return bn_commit_launch(false, n, cand, running_mean, running_var, C, guard_dev, stream);
}

SED_API int sed_bn_restore(int n, float* const* cand, float* const* running_mean, float* const* running_var, const int* C,
                           const int* guard_dev, hipStream_t stream) {
    return bn_commit_launch(true, n, cand, running_mean, running_var, C, guard_dev, stream);
}

SED_API int sed_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                               const float* running_var, float eps, float* mean_out, float* invstd_out,
                               float* scale_out, float* shift_out, hipStream_t stream) {
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(sed_cdiv(C, 64)), dim3(64), 0, stream, C, gamma, beta, running_mean,
                       running_var, eps, mean_out, invstd_out, scale_out, shift_out);
    SED_LAUNCH_CHECK();
    return 0;
}

// partials [nparts][2][C] = (sum dy, sum dy*xhat).  coef may be null (bn0: only dgamma/dbeta wanted).
// (defined here, ahead of its two users: sed_bn_bwd_finalize's chunked follow-up launch and sed_grad_bound)
namespace {
__global__ __launch_bounds__(256) void grad_bound_kernel(const float* __restrict__ mm, int nparts, int C,
                                                         const float* __restrict__ coef, const float* __restrict__ g_amax,
                                                         float ginv, int parts_per_block, float* __restrict__ bound_out,
                                                         const float* __restrict__ y_amax) {
    __shared__ float red[256];
    const int p0 = blockIdx.x * parts_per_block, p1 = min(nparts, p0 + parts_per_block);
    const float G = amax_read(g_amax) * ginv;
    const float A = y_amax ? amax_read(y_amax) : 0.f;        // mm null: |y| <= A for every channel (the looser, cheaper range)
    float best = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float mx = mm ? -__builtin_inff() : A, mn = mm ? __builtin_inff() : -A;
        for (int q = p0; mm && q < p1; ++q) {
            mx = fmaxf(mx, mm[((long)q * 2 + 0) * C + c]);
            mn = fminf(mn, mm[((long)q * 2 + 1) * C + c]);
        }
        if (mx >= mn) {
            const float a = coef[c], b = coef[C + c], cc = coef[2 * C + c];
            // 1.0001: the kernels evaluate a*dy + (b*y + c) with two roundings; the bound must never fall below the value
            best = fmaxf(best, (fabsf(a) * G + fmaxf(fabsf(fmaf(b, mx, cc)), fabsf(fmaf(b, mn, cc)))) * 1.0001f);
        }
    }
    red[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned*>(bound_out) + (blockIdx.x & (SED_AMAX_SLOTS - 1)), __float_as_uint(red[0]));
}
}  // namespace

SED_API int sed_bn_bwd_finalize(const float* partials, int nparts, long N, int C, const float* mean,
                                const float* invstd, const float* scale, int batch_stats, float* dgamma, float* dbeta,
                                float* coef, double* ws, const float* y_amax, const float* g_amax, float ginv, float* bound_out,
                                const float* minmax, hipStream_t stream) {
    if (nparts <= 0 || C <= 0 || N <= 0 || (bound_out && (!coef || !g_amax || ((y_amax == nullptr) == (minmax == nullptr)))))
        return SED_EINVAL;
    if (bound_out) {
        hipError_t e = sed_amax_clear(bound_out, stream);
        if (e != hipSuccess) return (int)e;
    }
    int ppc = reduce_chunks(nparts), nchunks = sed_cdiv(nparts, ppc), K = 2 * C;
    const bool direct = nparts <= SED_BN_DIRECT_MAX_PARTS;
    const BnDirectP dp{direct ? partials : nullptr, nparts, 0};
    if (!direct)
        hipLaunchKernelGGL(reduce_parts_kernel<0>, dim3(nchunks, sed_cdiv(K, 256)), dim3(256), 0, stream, partials, nparts, K,
                           ppc, N, 0, ws);
    // minmax [nparts][2][C] (the per-part range of y the forward convolution left; same parts as `partials`): the bound uses each
    // channel's own range -- inside this launch while one wave per channel walks the parts in a few trips, else by the chunked
    // grad_bound launch behind it, as before round 6
    const bool late = bound_out && minmax && !direct;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(sed_cdiv(C, 4)), dim3(256), 0, stream, ws, nchunks, dp, C, N, mean, invstd,
                       scale, batch_stats, dgamma, dbeta, coef, y_amax, g_amax, ginv, late ? nullptr : bound_out,
                       late ? nullptr : minmax, nparts);
    if (late) {
        int ppb = sed_cdiv(nparts, 1024);
        if (ppb < 8) ppb = 8;
        hipLaunchKernelGGL(grad_bound_kernel, dim3(sed_cdiv(nparts, ppb)), dim3(256), 0, stream, minmax, nparts, C, (const float*)coef,
                           g_amax, ginv, ppb, bound_out, (const float*)nullptr);
    }
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_bn0_aug_mix_fwd(const float* logmel, int B2, int T, const float* scale, const float* shift,
                                const int* stripes, const float* lam, float* out, hipStream_t stream) {
    if (B2 <= 0 || T <= 0 || (lam && (B2 & 1))) return SED_EINVAL;
    long total4 = (long)(lam ? B2 / 2 : B2) * T * 16;
    hipLaunchKernelGGL(bn0_aug_mix_fwd_kernel, dim3(stream_grid(total4)), dim3(256), 0, stream, logmel, B2, T, scale, shift,
                       stripes, lam, out);
    SED_LAUNCH_CHECK();
    return 0;
}

// partials must hold ceil(B2*T/1024)*128 floats; returns the number of parts through *nparts_out (host int).
SED_API int sed_bn0_aug_mix_bwd(const float* logmel, const float* g_out, int B2, int T, const float* mean,
                                const float* invstd, const int* stripes, const float* lam, float* partials,
                                int* nparts_out, hipStream_t stream) {
    if (B2 <= 0 || T <= 0 || (lam && (B2 & 1))) return SED_EINVAL;
    int nblk = sed_cdiv((long)B2 * T, 256);            // 256 rows per workgroup: partials must hold ceil(B2*T / 256) * 128 floats
    hipLaunchKernelGGL(bn0_aug_mix_bwd_kernel, dim3(nblk), dim3(256), 0, stream, logmel, g_out, B2, T, mean, invstd, stripes,
                       lam, 256, partials);
    if (nparts_out) *nparts_out = nblk;
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_bn_relu_pool_fwd(const float* y, int B, int H, int W, int C, int ph, int pw, const float* scale,
                                 const float* shift, float* out, float* amax_out, hipStream_t stream) {
    if (B <= 0 || (C & 3) || ph <= 0 || pw <= 0 || H / ph <= 0 || W / pw <= 0) return SED_EINVAL;
    if (amax_out) {
        hipError_t e = sed_amax_clear(amax_out, stream);
        if (e != hipSuccess) return (int)e;
    }
    long total = (long)B * (H / ph) * (W / pw) * (C / 4);
    hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<false>, dim3(stream_grid(total)), dim3(256), 0, stream, y, B, H, W, C, ph, pw,
                       scale, shift, out, (unsigned*)nullptr, amax_out);
    SED_LAUNCH_CHECK();
    return 0;
}

// Same, also writing cnt [B][H/ph][W/pw][C] bytes = number of window inputs that passed the ReLU (ph*pw <= 255).
SED_API int sed_bn_relu_pool_fwd_cnt(const float* y, int B, int H, int W, int C, int ph, int pw, const float* scale,
                                     const float* shift, float* out, unsigned char* cnt, float* amax_out,
                                     hipStream_t stream) {
    if (B <= 0 || (C & 3) || ph <= 0 || pw <= 0 || H / ph <= 0 || W / pw <= 0 || ph * pw > 255 || !cnt) return SED_EINVAL;
    if (amax_out) {
        hipError_t e = sed_amax_clear(amax_out, stream);
        if (e != hipSuccess) return (int)e;
    }
    long total = (long)B * (H / ph) * (W / pw) * (C / 4);
    hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<true>, dim3(stream_grid(total)), dim3(256), 0, stream, y, B, H, W, C, ph, pw,
                       scale, shift, out, reinterpret_cast<unsigned*>(cnt), amax_out);
    SED_LAUNCH_CHECK();
    return 0;
}

// ---- the pooled block output as split-f16 operand pairs (round 4): read only by the next block's conv1 (forward and weight
// gradient: plain-copy staging) and by this block's windowed backward pass 1 (decodes).  The scale needs an upper bound of the
// pooled amax BEFORE the pass: the average of relu(scale*y + shift) over a window is <= max_c (|scale_c| * amax|y| + |shift_c|).
namespace {
__global__ __launch_bounds__(256) void act_bound_kernel(const float* __restrict__ y_amax, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int C, float* __restrict__ bound_out) {
    __shared__ float red[256];
    const float A = amax_read(y_amax);
    float best = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) best = fmaxf(best, fmaxf(fmaf(fabsf(scale[c]), A, shift[c]), 0.f) * 1.0001f);
    red[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) bound_out[0] = red[0];
}
}  // namespace

SED_API int sed_act_bound(const float* y_amax, const float* scale, const float* shift, int C, float* bound_out, hipStream_t stream) {
    if (!y_amax || !scale || !shift || !bound_out || C <= 0) return SED_EINVAL;
    hipError_t e = sed_amax_clear(bound_out, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(act_bound_kernel, dim3(1), dim3(256), 0, stream, y_amax, scale, shift, C, bound_out);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_bn_relu_pool_fwd_cnt_pairs(const float* y, int B, int H, int W, int C, int ph, int pw, const float* scale,
                                           const float* shift, void* out_pairs, unsigned char* cnt, const float* bound,
                                           hipStream_t stream) {
    if (!y || !out_pairs || !bound || B <= 0 || (C & 3) || ph <= 0 || pw <= 0 || H / ph <= 0 || W / pw <= 0 || ph * pw > 255 || !cnt)
        return SED_EINVAL;
    long total = (long)B * (H / ph) * (W / pw) * (C / 4);
    hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<true>, dim3(stream_grid(total)), dim3(256), 0, stream, y, B, H, W, C, ph, pw,
                       scale, shift, (float*)out_pairs, reinterpret_cast<unsigned*>(cnt), (float*)nullptr, bound);
    SED_LAUNCH_CHECK();
    return 0;
}

// Backward pass 1 from the pooled tensors: g_out, pooled, cnt [B*Ho*Wo][C]; gamma / beta = the BatchNorm weight / bias.
// partials must hold ceil(M' / sed_pool_bwd_rows_per_block(M')) * 2*C floats, M' = B*Ho*Wo.
SED_API int sed_bn_relu_pool_bwd_reduce_win(const float* g_out, const float* pooled, const unsigned char* cnt, long Mp,
                                            int C, int window, const float* gamma, const float* beta, float* partials,
                                            int* nparts_out, hipStream_t stream) {
    if (Mp <= 0 || C < 64 || C > 512 || (256 % (C / 4)) != 0 || window <= 0 || window > 255) return SED_EINVAL;
    const int rpb = sed_pool_bwd_rows_per_block(Mp);
    int nblk = sed_cdiv(Mp, rpb);
    hipLaunchKernelGGL(pool_bwd_reduce_win_kernel, dim3(nblk), dim3(256), 0, stream, g_out, pooled,
                       reinterpret_cast<const unsigned*>(cnt), Mp, C, 1.0f / (float)window, gamma, beta, rpb, partials, 0.f);
    if (nparts_out) *nparts_out = nblk;
    SED_LAUNCH_CHECK();
    return 0;
}

// Pass 1 with the choice made ON THE DEVICE, in stream order, from this step's gamma: the windowed kernel runs when every
// |gamma[c]| >= gamma_min, otherwise it returns at once and the full-resolution kernel (which returns at once in the
// common case) produces the sums.  partials: max(parts of either) * 2*C floats, zeroed here; *nparts_out = that maximum.
SED_API long sed_bn_relu_pool_bwd_reduce_auto_parts(int B, int H, int W, int ph, int pw) {
    if (B <= 0 || ph <= 0 || pw <= 0 || H / ph <= 0 || W / pw <= 0) return SED_EINVAL;
    const long M = (long)B * H * W, Mp = (long)B * (H / ph) * (W / pw);
    const long a = sed_cdiv(M, sed_pool_bwd_rows_per_block(M)), b = sed_cdiv(Mp, sed_pool_bwd_rows_per_block(Mp));
    return a > b ? a : b;
}
SED_API int sed_bn_relu_pool_bwd_reduce_auto(const float* y, const float* g_out, const float* pooled,
                                             const unsigned char* cnt, int B, int H, int W, int C, int ph, int pw,
                                             const float* scale, const float* shift, const float* mean,
                                             const float* invstd, const float* gamma, const float* beta,
                                             float gamma_min, float* partials, int* nparts_out, const float* pooled_bound,
                                             hipStream_t stream) {
    if (B <= 0 || C < 64 || C > 512 || (256 % (C / 4)) != 0 || ph * pw > 255 || !(gamma_min > 0.f)) return SED_EINVAL;
    const long nparts = sed_bn_relu_pool_bwd_reduce_auto_parts(B, H, W, ph, pw);
    if (nparts <= 0) return SED_EINVAL;
    hipError_t e = hipMemsetAsync(partials, 0, (size_t)nparts * 2 * C * sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
    const long M = (long)B * H * W, Mp = (long)B * (H / ph) * (W / pw);
    const int rpb = sed_pool_bwd_rows_per_block(M), rpbw = sed_pool_bwd_rows_per_block(Mp);
    hipLaunchKernelGGL(pool_bwd_reduce_win_kernel, dim3(sed_cdiv(Mp, rpbw)), dim3(256), 0, stream, g_out, pooled,
                       reinterpret_cast<const unsigned*>(cnt), Mp, C, 1.0f / (float)(ph * pw), gamma, beta, rpbw, partials,
                       gamma_min, pooled_bound);
    hipLaunchKernelGGL(bn_relu_pool_bwd_kernel<1>, dim3(sed_cdiv(M, rpb)), dim3(256), 0, stream, y, g_out, B, H, W, C, ph, pw,
                       scale, shift, mean, invstd, (const float*)nullptr, rpb, partials, (float*)nullptr, gamma, gamma_min, (float*)nullptr);
    if (nparts_out) *nparts_out = (int)nparts;
    SED_LAUNCH_CHECK();
    return 0;
}

// rows of y per workgroup of the two pool-backward passes: 1024 for big tensors, fewer (>= 64) for small ones so that the
// grid still holds ~2048 workgroups (at batch 32 a fixed 1024 left block 4 with 31 workgroups on 256 CUs)
SED_API int sed_pool_bwd_rows_per_block(long M) {
    long r = (M + 2047) / 2048;
    r = (r + 63) / 64 * 64;
    return (int)(r < 64 ? 64 : (r > 1024 ? 1024 : r));
}

// pass 1: partials must hold ceil(B*H*W / sed_pool_bwd_rows_per_block(B*H*W)) * 2*C floats
SED_API int sed_bn_relu_pool_bwd_reduce(const float* y, const float* g_out, int B, int H, int W, int C, int ph, int pw,
                                        const float* scale, const float* shift, const float* mean,
                                        const float* invstd, float* partials, int* nparts_out, hipStream_t stream) {
    if (B <= 0 || C < 64 || C > 512 || (256 % (C / 4)) != 0) return SED_EINVAL;
    const int rpb = sed_pool_bwd_rows_per_block((long)B * H * W);
    int nblk = sed_cdiv((long)B * H * W, rpb);
    hipLaunchKernelGGL(bn_relu_pool_bwd_kernel<1>, dim3(nblk), dim3(256), 0, stream, y, g_out, B, H, W, C, ph, pw, scale, shift,
                       mean, invstd, (const float*)nullptr, rpb, partials, (float*)nullptr, (const float*)nullptr, 0.f, (float*)nullptr);
    if (nparts_out) *nparts_out = nblk;
    SED_LAUNCH_CHECK();
    return 0;
}

// pass 2: gy [B][H][W][C]
SED_API int sed_bn_relu_pool_bwd_apply(const float* y, const float* g_out, int B, int H, int W, int C, int ph, int pw,
                                       const float* scale, const float* shift, const float* coef, float* gy,
                                       float* amax_out, hipStream_t stream) {
    if (B <= 0 || C < 64 || C > 512 || (256 % (C / 4)) != 0) return SED_EINVAL;
    if (amax_out) {
        hipError_t e = sed_amax_clear(amax_out, stream);
        if (e != hipSuccess) return (int)e;
    }
    const int rpb = sed_pool_bwd_rows_per_block((long)B * H * W);
    int nblk = sed_cdiv((long)B * H * W, rpb);
    hipLaunchKernelGGL(bn_relu_pool_bwd_kernel<2>, dim3(nblk), dim3(256), 0, stream, y, g_out, B, H, W, C, ph, pw, scale, shift,
                       (const float*)nullptr, (const float*)nullptr, coef, rpb, (float*)nullptr, gy, (const float*)nullptr, 0.f, amax_out);
    SED_LAUNCH_CHECK();
    return 0;
}

// ---- the same three stages for any pool_type of ConvBlock (models.py:104-111): pool_mode 0 = 'avg', 1 = 'max', 2 = 'avg+max'.
// No model of the reference selects 1 or 2; they run the exact full-resolution backward pass 1 (the windowed shortcut is an
// identity of the average) and re-read each element's window to find the first maximum.
SED_API int sed_bn_relu_pool_fwd_mode(const float* y, int B, int H, int W, int C, int ph, int pw, int pool_mode, const float* scale,
                                      const float* shift, float* out, float* amax_out, hipStream_t stream) {
    if (pool_mode == 0) return sed_bn_relu_pool_fwd(y, B, H, W, C, ph, pw, scale, shift, out, amax_out, stream);
    if (B <= 0 || (C & 3) || ph <= 0 || pw <= 0 || H / ph <= 0 || W / pw <= 0 || pool_mode < 0 || pool_mode > 2) return SED_EINVAL;
    if (amax_out) {
        hipError_t e = sed_amax_clear(amax_out, stream);
        if (e != hipSuccess) return (int)e;
    }
    long total = (long)B * (H / ph) * (W / pw) * (C / 4);
    if (pool_mode == 1)
        hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<false, 1>), dim3(stream_grid(total)), dim3(256), 0, stream, y, B, H, W, C, ph, pw,
                           scale, shift, out, (unsigned*)nullptr, amax_out);
    else
        hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<false, 2>), dim3(stream_grid(total)), dim3(256), 0, stream, y, B, H, W, C, ph, pw,
                           scale, shift, out, (unsigned*)nullptr, amax_out);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_bn_relu_pool_bwd_reduce_mode(const float* y, const float* g_out, int B, int H, int W, int C, int ph, int pw,
                                             int pool_mode, const float* scale, const float* shift, const float* mean,
                                             const float* invstd, float* partials, int* nparts_out, hipStream_t stream) {
    if (pool_mode == 0)
        return sed_bn_relu_pool_bwd_reduce(y, g_out, B, H, W, C, ph, pw, scale, shift, mean, invstd, partials, nparts_out, stream);
    if (B <= 0 || C < 64 || C > 512 || (256 % (C / 4)) != 0 || pool_mode < 0 || pool_mode > 2) return SED_EINVAL;
    const int rpb = sed_pool_bwd_rows_per_block((long)B * H * W);
    int nblk = sed_cdiv((long)B * H * W, rpb);
    if (pool_mode == 1)
        hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<1, 1>), dim3(nblk), dim3(256), 0, stream, y, g_out, B, H, W, C, ph, pw, scale, shift,
                           mean, invstd, (const float*)nullptr, rpb, partials, (float*)nullptr, (const float*)nullptr, 0.f, (float*)nullptr);
    else
        hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<1, 2>), dim3(nblk), dim3(256), 0, stream, y, g_out, B, H, W, C, ph, pw, scale, shift,
                           mean, invstd, (const float*)nullptr, rpb, partials, (float*)nullptr, (const float*)nullptr, 0.f, (float*)nullptr);
    if (nparts_out) *nparts_out = nblk;
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_bn_relu_pool_bwd_apply_mode(const float* y, const float* g_out, int B, int H, int W, int C, int ph, int pw,
                                            int pool_mode, const float* scale, const float* shift, const float* coef, float* gy,
                                            float* amax_out, hipStream_t stream) {
    if (pool_mode == 0) return sed_bn_relu_pool_bwd_apply(y, g_out, B, H, W, C, ph, pw, scale, shift, coef, gy, amax_out, stream);
    if (B <= 0 || C < 64 || C > 512 || (256 % (C / 4)) != 0 || pool_mode < 0 || pool_mode > 2) return SED_EINVAL;
    if (amax_out) {
        hipError_t e = sed_amax_clear(amax_out, stream);
        if (e != hipSuccess) return (int)e;
    }
    const int rpb = sed_pool_bwd_rows_per_block((long)B * H * W);
    int nblk = sed_cdiv((long)B * H * W, rpb);
    if (pool_mode == 1)
        hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<2, 1>), dim3(nblk), dim3(256), 0, stream, y, g_out, B, H, W, C, ph, pw, scale, shift,
                           (const float*)nullptr, (const float*)nullptr, coef, rpb, (float*)nullptr, gy, (const float*)nullptr, 0.f, amax_out);
    else
        hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<2, 2>), dim3(nblk), dim3(256), 0, stream, y, g_out, B, H, W, C, ph, pw, scale, shift,
                           (const float*)nullptr, (const float*)nullptr, coef, rpb, (float*)nullptr, gy, (const float*)nullptr, 0.f, amax_out);
    SED_LAUNCH_CHECK();
    return 0;
}

// amax of relu(scale*y + shift) (scale / shift null: of |y|) from range partials minmax [nparts][2][C] = per-part
// (max, min) per channel, as left by sed_conv1_fwd / sed_conv3x3_sf16.
SED_API int sed_act_amax(const float* minmax, int nparts, int C, const float* scale, const float* shift, float* amax_out,
                         hipStream_t stream) {
    if (!minmax || !amax_out || nparts <= 0 || C <= 0 || ((scale == nullptr) != (shift == nullptr))) return SED_EINVAL;
    hipError_t e = sed_amax_clear(amax_out, stream);
    if (e != hipSuccess) return (int)e;
    int ppb = sed_cdiv(nparts, 1024);
    if (ppb < 8) ppb = 8;
    hipLaunchKernelGGL(act_amax_kernel, dim3(sed_cdiv(nparts, ppb)), dim3(256), 0, stream, minmax, nparts, C, scale, shift, ppb,
                       amax_out);
    SED_LAUNCH_CHECK();
    return 0;
}

// The same quantity by one pass over y [nrows][C] (for producers without range partials).
SED_API int sed_act_amax_full(const float* y, long nrows, int C, const float* scale, const float* shift, float* amax_out,
                              hipStream_t stream) {
    if (!y || !amax_out || !scale || !shift || nrows <= 0 || (C & 3)) return SED_EINVAL;
    hipError_t e = sed_amax_clear(amax_out, stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(act_amax_full_kernel, dim3(stream_grid(nrows * (C / 4))), dim3(256), 0, stream, y, nrows, C, scale, shift,
                       amax_out);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_bn_bwd_apply(float* dy_inout, const float* y, long nrows, int C, const float* coef, float* amax_out,
                             hipStream_t stream) {
    if (nrows <= 0 || (C & 3)) return SED_EINVAL;
    if (amax_out) {
        hipError_t e = sed_amax_clear(amax_out, stream);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(nrows * (C / 4))), dim3(256), 0, stream, dy_inout, y, nrows, C,
                       coef, amax_out, (const float*)nullptr);
    SED_LAUNCH_CHECK();
    return 0;
}

// ---- gradients written as split-f16 operand pairs (round 4).  The BatchNorm-backward apply kernels produce the tensors that ONLY
// the split-f16 dgrad / weight-gradient kernels read; written as pairs (same bytes), every consumer tile copies them into
// LDS instead of re-deriving (hi, lo) -- 64/32 of them re-derive the same values today.  The scale must be known BEFORE the
// pass: sed_grad_bound gives an upper bound of max |a*dy + b*y + c| from what is known per channel -- the coefficients, the
// range of y (the conv epilogue's minmax partials) and the amax of the incoming gradient: |a|*G*ginv + max(|b*ymax + c|,
// |b*ymin + c|).  A bound that is a few times too large costs nothing: hi and lo are floating-point numbers.

SED_API int sed_grad_bound(const float* minmax, int nparts, int C, const float* coef, const float* g_amax, float ginv,
                           float* bound_out, const float* y_amax, hipStream_t stream) {
    if ((minmax == nullptr) == (y_amax == nullptr) || !coef || !g_amax || !bound_out || C <= 0 || (minmax && nparts <= 0))
        return SED_EINVAL;
    hipError_t e = sed_amax_clear(bound_out, stream);
    if (e != hipSuccess) return (int)e;
    if (!minmax) nparts = 1;
    int ppb = sed_cdiv(nparts, 1024);
    if (ppb < 8) ppb = 8;
    hipLaunchKernelGGL(grad_bound_kernel, dim3(sed_cdiv(nparts, ppb)), dim3(256), 0, stream, minmax, nparts, C, coef, g_amax, ginv,
                       ppb, bound_out, y_amax);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_bn_bwd_apply_pairs(float* dy_inout, const float* y, long nrows, int C, const float* coef, const float* bound,
                                   hipStream_t stream) {
    if (!dy_inout || !y || !coef || !bound || nrows <= 0 || (C & 3)) return SED_EINVAL;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(nrows * (C / 4))), dim3(256), 0, stream, dy_inout, y, nrows, C,
                       coef, (float*)nullptr, bound);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_bn_relu_pool_bwd_apply_pairs(const float* y, const float* g_out, int B, int H, int W, int C, int ph, int pw,
                                             const float* scale, const float* shift, const float* coef, void* gy_pairs,
                                             const float* bound, hipStream_t stream) {
    if (!y || !g_out || !coef || !gy_pairs || !bound || B <= 0 || C < 64 || C > 512 || (256 % (C / 4)) != 0) return SED_EINVAL;
    const int rpb = sed_pool_bwd_rows_per_block((long)B * H * W);
    int nblk = sed_cdiv((long)B * H * W, rpb);
    hipLaunchKernelGGL(bn_relu_pool_bwd_kernel<2>, dim3(nblk), dim3(256), 0, stream, y, g_out, B, H, W, C, ph, pw, scale, shift,
                       (const float*)nullptr, (const float*)nullptr, coef, rpb, (float*)nullptr, (float*)gy_pairs,
                       (const float*)nullptr, 0.f, (float*)nullptr, bound);
    SED_LAUNCH_CHECK();
    return 0;
}

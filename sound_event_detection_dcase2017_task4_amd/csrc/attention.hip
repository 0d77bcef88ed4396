// Multi-head self-attention core of the Transformer heads (reference pytorch/models.py:587-665): 8 heads x 64,
// T = 125 frames (10 s clips).  4 MFLOP per (clip, head) = 0.02 % of the step's arithmetic, so these are plain
// fp32 vector kernels (one thread per query / key row, K/V or Q/dO chunks broadcast from LDS), not MFMA tiles.
//
//   S = Q K^T / sqrt(64),  P = softmax_j(S),  Pd = P * keep / (1 - p) (training),  O = Pd V
//   backward:  D_i = dO_i . O_i (= sum_j P_ij dP_ij),  dP = (dO V^T) * keep / (1 - p),  dS = P (dP - D) / sqrt(64),
//              dQ = dS K,  dK = dS^T Q,  dV = Pd^T dO
// q, k, v, o: [B*T][8*64] fp32 with head h in columns 64h .. 64h+63 (the Linear outputs, no permutes);
// keep mask: bytes [8*B][T][T] with row index h*B + b (the (n*b) layout of models.py:651-657), or null in eval mode;
// stats [B][8][T][4] = (row max m, row sum l, D, unused).
#include "common.h"
#include "sed_hip.h"
SED_OBJECT_FLAGS(attention)

namespace {

constexpr int MHA_H = 8, MHA_D = 64, MHA_LD = MHA_H * MHA_D, MHA_CH = 64;   // keys / queries staged per chunk

// ---- forward: one workgroup per (b, h); thread r handles query rows r, r+128, ...
__global__ __launch_bounds__(128) void mha_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, const unsigned char* __restrict__ keep,
                                                      float inv_keep, int B, int T, float* __restrict__ o,
                                                      float* __restrict__ stats) {
    __shared__ float Ks[MHA_CH][MHA_D], Vs[MHA_CH][MHA_D];
    const int b = blockIdx.x / MHA_H, h = blockIdx.x % MHA_H;
    const long row0 = (long)b * T;
    const float scale = 0.125f;                                  // 1 / sqrt(64)
    for (int i0 = 0; i0 < T; i0 += 128) {
        const int i = i0 + threadIdx.x;
        const bool act = i < T;
        float qi[MHA_D], acc[MHA_D];
#pragma unroll
        for (int d = 0; d < MHA_D; d += 4) {
            float4 t4 = act ? *reinterpret_cast<const float4*>(q + (row0 + i) * MHA_LD + h * MHA_D + d) : make_float4(0, 0, 0, 0);
            qi[d] = t4.x * scale; qi[d + 1] = t4.y * scale; qi[d + 2] = t4.z * scale; qi[d + 3] = t4.w * scale;
            acc[d] = acc[d + 1] = acc[d + 2] = acc[d + 3] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        const unsigned char* kp = keep ? keep + (((long)h * B + b) * T + (act ? i : 0)) * T : nullptr;
        for (int j0 = 0; j0 < T; j0 += MHA_CH) {
            const int nj = min(MHA_CH, T - j0);
            __syncthreads();
            for (int e = threadIdx.x; e < MHA_CH * MHA_D / 4; e += 128) {
                const int jr = e / (MHA_D / 4), c4 = e % (MHA_D / 4);
                float4 kk = make_float4(0, 0, 0, 0), vv = kk;
                if (jr < nj) {
                    kk = *reinterpret_cast<const float4*>(k + (row0 + j0 + jr) * MHA_LD + h * MHA_D + c4 * 4);
                    vv = *reinterpret_cast<const float4*>(v + (row0 + j0 + jr) * MHA_LD + h * MHA_D + c4 * 4);
                }
                *reinterpret_cast<float4*>(&Ks[jr][c4 * 4]) = kk;
                *reinterpret_cast<float4*>(&Vs[jr][c4 * 4]) = vv;
            }
            __syncthreads();
            for (int jr = 0; jr < nj; ++jr) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < MHA_D; ++d) s = fmaf(qi[d], Ks[jr][d], s);
                const float mn = fmaxf(m, s);
                const float corr = __expf(m - mn), e = __expf(s - mn);
                l = l * corr + e;
                const float w = (kp && !kp[j0 + jr]) ? 0.f : e;
#pragma unroll
                for (int d = 0; d < MHA_D; ++d) acc[d] = fmaf(acc[d], corr, w * Vs[jr][d]);
                m = mn;
            }
        }
        if (act) {
            const float r = inv_keep / l;
#pragma unroll
            for (int d = 0; d < MHA_D; d += 4)
                *reinterpret_cast<float4*>(o + (row0 + i) * MHA_LD + h * MHA_D + d) =
                    make_float4(acc[d] * r, acc[d + 1] * r, acc[d + 2] * r, acc[d + 3] * r);
            if (stats) {
                float* st = stats + (((long)b * MHA_H + h) * T + i) * 4;
                st[0] = m; st[1] = l;
            }
        }
    }
}

// ---- backward, query side: D_i and dQ_i (thread per query row, K / V chunks from LDS)
__global__ __launch_bounds__(128) void mha_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ o,
                                                        const float* __restrict__ go, const unsigned char* __restrict__ keep,
                                                        float inv_keep, int B, int T, float* __restrict__ stats,
                                                        float* __restrict__ gq) {
    __shared__ float Ks[MHA_CH][MHA_D], Vs[MHA_CH][MHA_D];
    const int b = blockIdx.x / MHA_H, h = blockIdx.x % MHA_H;
    const long row0 = (long)b * T;
    const float scale = 0.125f;
    for (int i0 = 0; i0 < T; i0 += 128) {
        const int i = i0 + threadIdx.x;
        const bool act = i < T;
        float qi[MHA_D], gi[MHA_D], acc[MHA_D];
        float Di = 0.f;
#pragma unroll
        for (int d = 0; d < MHA_D; d += 4) {
            const long off = (row0 + (act ? i : 0)) * MHA_LD + h * MHA_D + d;
            const float4 t4 = *reinterpret_cast<const float4*>(q + off), g4 = *reinterpret_cast<const float4*>(go + off),
                         o4 = *reinterpret_cast<const float4*>(o + off);
            qi[d] = t4.x * scale; qi[d + 1] = t4.y * scale; qi[d + 2] = t4.z * scale; qi[d + 3] = t4.w * scale;
            gi[d] = g4.x; gi[d + 1] = g4.y; gi[d + 2] = g4.z; gi[d + 3] = g4.w;
            Di += g4.x * o4.x + g4.y * o4.y + g4.z * o4.z + g4.w * o4.w;
            acc[d] = acc[d + 1] = acc[d + 2] = acc[d + 3] = 0.f;
        }
        float* st = stats + (((long)b * MHA_H + h) * T + (act ? i : 0)) * 4;
        const float m = st[0], rl = 1.f / st[1];
        if (act) st[2] = Di;
        const unsigned char* kp = keep ? keep + (((long)h * B + b) * T + (act ? i : 0)) * T : nullptr;
        for (int j0 = 0; j0 < T; j0 += MHA_CH) {
            const int nj = min(MHA_CH, T - j0);
            __syncthreads();
            for (int e = threadIdx.x; e < MHA_CH * MHA_D / 4; e += 128) {
                const int jr = e / (MHA_D / 4), c4 = e % (MHA_D / 4);
                float4 kk = make_float4(0, 0, 0, 0), vv = kk;
                if (jr < nj) {
                    kk = *reinterpret_cast<const float4*>(k + (row0 + j0 + jr) * MHA_LD + h * MHA_D + c4 * 4);
                    vv = *reinterpret_cast<const float4*>(v + (row0 + j0 + jr) * MHA_LD + h * MHA_D + c4 * 4);
                }
                *reinterpret_cast<float4*>(&Ks[jr][c4 * 4]) = kk;
                *reinterpret_cast<float4*>(&Vs[jr][c4 * 4]) = vv;
            }
            __syncthreads();
            for (int jr = 0; jr < nj; ++jr) {
                float s = 0.f, dpd = 0.f;
#pragma unroll
                for (int d = 0; d < MHA_D; ++d) { s = fmaf(qi[d], Ks[jr][d], s); dpd = fmaf(gi[d], Vs[jr][d], dpd); }
                const float p = __expf(s - m) * rl;
                const float dp = (kp && !kp[j0 + jr]) ? 0.f : dpd * inv_keep;
                const float ds = p * (dp - Di) * scale;
#pragma unroll
                for (int d = 0; d < MHA_D; ++d) acc[d] = fmaf(ds, Ks[jr][d], acc[d]);
            }
        }
        if (act) {
#pragma unroll
            for (int d = 0; d < MHA_D; d += 4)
                *reinterpret_cast<float4*>(gq + (row0 + i) * MHA_LD + h * MHA_D + d) = make_float4(acc[d], acc[d + 1], acc[d + 2], acc[d + 3]);
        }
    }
}

// ---- backward, key side: dK_j and dV_j (thread per key row; Q / dO chunks and their row statistics from LDS)
__global__ __launch_bounds__(128) void mha_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ go,
                                                         const unsigned char* __restrict__ keep, float inv_keep, int B, int T,
                                                         const float* __restrict__ stats, float* __restrict__ gk,
                                                         float* __restrict__ gv) {
    __shared__ float Qs[MHA_CH][MHA_D], Gs[MHA_CH][MHA_D];
    __shared__ float Ms[MHA_CH], Ls[MHA_CH], Ds[MHA_CH];
    const int b = blockIdx.x / MHA_H, h = blockIdx.x % MHA_H;
    const long row0 = (long)b * T;
    const float scale = 0.125f;
    for (int j0 = 0; j0 < T; j0 += 128) {
        const int j = j0 + threadIdx.x;
        const bool act = j < T;
        float kj[MHA_D], vj[MHA_D], ak[MHA_D], av[MHA_D];
#pragma unroll
        for (int d = 0; d < MHA_D; d += 4) {
            const long off = (row0 + (act ? j : 0)) * MHA_LD + h * MHA_D + d;
            const float4 k4 = *reinterpret_cast<const float4*>(k + off), v4 = *reinterpret_cast<const float4*>(v + off);
            kj[d] = k4.x * scale; kj[d + 1] = k4.y * scale; kj[d + 2] = k4.z * scale; kj[d + 3] = k4.w * scale;
            vj[d] = v4.x; vj[d + 1] = v4.y; vj[d + 2] = v4.z; vj[d + 3] = v4.w;
            ak[d] = ak[d + 1] = ak[d + 2] = ak[d + 3] = 0.f;
            av[d] = av[d + 1] = av[d + 2] = av[d + 3] = 0.f;
        }
        for (int i0 = 0; i0 < T; i0 += MHA_CH) {
            const int ni = min(MHA_CH, T - i0);
            __syncthreads();
            for (int e = threadIdx.x; e < MHA_CH * MHA_D / 4; e += 128) {
                const int ir = e / (MHA_D / 4), c4 = e % (MHA_D / 4);
                float4 qq = make_float4(0, 0, 0, 0), gg = qq;
                if (ir < ni) {
                    qq = *reinterpret_cast<const float4*>(q + (row0 + i0 + ir) * MHA_LD + h * MHA_D + c4 * 4);
                    gg = *reinterpret_cast<const float4*>(go + (row0 + i0 + ir) * MHA_LD + h * MHA_D + c4 * 4);
                }
                *reinterpret_cast<float4*>(&Qs[ir][c4 * 4]) = qq;
                *reinterpret_cast<float4*>(&Gs[ir][c4 * 4]) = gg;
            }
            if (threadIdx.x < ni) {
                const float* st = stats + (((long)b * MHA_H + h) * T + i0 + threadIdx.x) * 4;
                Ms[threadIdx.x] = st[0]; Ls[threadIdx.x] = 1.f / st[1]; Ds[threadIdx.x] = st[2];
            }
            __syncthreads();
            for (int ir = 0; ir < ni; ++ir) {
                float s = 0.f, dpd = 0.f;
#pragma unroll
                for (int d = 0; d < MHA_D; ++d) { s = fmaf(Qs[ir][d], kj[d], s); dpd = fmaf(Gs[ir][d], vj[d], dpd); }
                const float p = __expf(s - Ms[ir]) * Ls[ir];
                const bool kept = !(keep && !keep[(((long)h * B + b) * T + i0 + ir) * T + (act ? j : 0)]);
                const float pd = kept ? p * inv_keep : 0.f;                 // dropped attention weight
                const float ds = p * ((kept ? dpd * inv_keep : 0.f) - Ds[ir]) * scale;   // d(q_i . k_j), q unscaled
#pragma unroll
                for (int d = 0; d < MHA_D; ++d) {
                    av[d] = fmaf(pd, Gs[ir][d], av[d]);
                    ak[d] = fmaf(ds, Qs[ir][d], ak[d]);
                }
            }
        }
        if (act) {
#pragma unroll
            for (int d = 0; d < MHA_D; d += 4) {
                const long off = (row0 + j) * MHA_LD + h * MHA_D + d;
                *reinterpret_cast<float4*>(gk + off) = make_float4(ak[d], ak[d + 1], ak[d + 2], ak[d + 3]);
                *reinterpret_cast<float4*>(gv + off) = make_float4(av[d], av[d + 1], av[d + 2], av[d + 3]);
            }
        }
    }
}

// y = relu(x * keep / (1 - p)) (keep null: y = relu(x));  backward g_x = g_y * keep / (1 - p) where y > 0
__global__ __launch_bounds__(256) void drop_relu_fwd_kernel(const float* __restrict__ x, const unsigned char* __restrict__ keep,
                                                            float inv_keep, long n, float* __restrict__ y) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = (keep && !keep[i]) ? 0.f : x[i] * inv_keep;
        y[i] = fmaxf(v, 0.f);
    }
}
__global__ __launch_bounds__(256) void drop_relu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                            const unsigned char* __restrict__ keep, float inv_keep, long n,
                                                            float* __restrict__ gx) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        gx[i] = (y[i] > 0.f && !(keep && !keep[i])) ? gy[i] * inv_keep : 0.f;
}

}  // namespace

SED_API int sed_mha_fwd(const float* q, const float* k, const float* v, const unsigned char* keep, float p_drop, int B, int T,
                        float* o, float* stats, hipStream_t stream) {
    if (B <= 0 || T <= 0 || p_drop < 0.f || p_drop >= 1.f || (long)B * MHA_H >= (1L << 31)) return SED_EINVAL;
    hipLaunchKernelGGL(mha_fwd_kernel, dim3(B * MHA_H), dim3(128), 0, stream, q, k, v, keep, keep ? 1.f / (1.f - p_drop) : 1.f, B, T,
                       o, stats);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_mha_bwd(const float* q, const float* k, const float* v, const float* o, const float* g_o,
                        const unsigned char* keep, float p_drop, int B, int T, float* stats, float* g_q, float* g_k, float* g_v,
                        hipStream_t stream) {
    if (B <= 0 || T <= 0 || !stats || p_drop < 0.f || p_drop >= 1.f || (long)B * MHA_H >= (1L << 31)) return SED_EINVAL;
    const float ik = keep ? 1.f / (1.f - p_drop) : 1.f;
    hipLaunchKernelGGL(mha_bwd_q_kernel, dim3(B * MHA_H), dim3(128), 0, stream, q, k, v, o, g_o, keep, ik, B, T, stats, g_q);
    hipLaunchKernelGGL(mha_bwd_kv_kernel, dim3(B * MHA_H), dim3(128), 0, stream, q, k, v, g_o, keep, ik, B, T, stats, g_k, g_v);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_drop_relu_fwd(const float* x, const unsigned char* keep, float p_drop, long n, float* y, hipStream_t stream) {
    if (n <= 0 || p_drop < 0.f || p_drop >= 1.f) return SED_EINVAL;
    long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(drop_relu_fwd_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, x, keep,
                       keep ? 1.f / (1.f - p_drop) : 1.f, n, y);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_drop_relu_bwd(const float* g_y, const float* y, const unsigned char* keep, float p_drop, long n, float* g_x,
                              hipStream_t stream) {
    if (n <= 0 || p_drop < 0.f || p_drop >= 1.f) return SED_EINVAL;
    long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(drop_relu_bwd_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, g_y, y, keep,
                       keep ? 1.f / (1.f - p_drop) : 1.f, n, g_x);
    SED_LAUNCH_CHECK();
    return 0;
}

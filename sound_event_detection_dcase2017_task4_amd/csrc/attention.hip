// Multi-head self-attention core of the Transformer heads (reference pytorch/models.py:587-665): 8 heads x 64,
// T = 125 frames (10 s clips).  Two sets of kernels: fp32-MFMA kernels for T <= 128 (the whole score tile of a (clip, head)
// at once; second half of this file) and plain fp32 vector kernels for any T (one thread per query / key row, K/V or Q/dO
// chunks broadcast from LDS).  The vector kernels were the only ones until round 4: 4 MFLOP per (clip, head) is 0.02 % of the
// step's arithmetic, but their 128 LDS broadcasts per key and thread made them 1.87 ms of a 64 ms step at B = 256.
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

// ==== T <= 128 (10 s clips: T = 125): the whole 128 x 128 score tile of a (clip, head) at once, on the fp32 MFMA pipe ==========
// One workgroup = 4 waves per (b, h); a wave owns 32 queries (forward, query-side backward) or 32 keys (key-side backward).
// v_mfma_f32_32x32x2_f32: A lane = (row l % 32, k l / 32), B lane = (k l / 32, col l % 32), accumulator register r of lane l =
// (row 8 (r / 4) + r % 4 + 4 (l / 32), col l % 32).  Two things make the kernels short:
//  * the reduction index of a dot product may be permuted: lane half 0 takes d = 0..31, half 1 takes d = 32..63, so a lane's
//    operand run is 32 CONSECUTIVE floats -- registers for the wave's own rows (8 float4 loads), ds_read_b128 for the others;
//  * scores are computed TRANSPOSED, S^T[key][query]: a lane then holds ONE query's logits (64 of them, the other 64 in lane
//    l ^ 32), so the softmax is in-lane work plus one DPP exchange, and accumulator register r is exactly the A operand
//    (row = query, k = lane half) of the k-step that multiplies keys 8 (r / 4) + r % 4 + {0, 4} of the tile into P V: the
//    probabilities never leave the registers.  The same holds for dS -> dQ and, with the roles of queries and keys swapped,
//    for P^T dO -> dV and dS^T Q -> dK.
// fp32 operands and fp32 accumulation throughout: no operand scales, the rounding of a plain fp32 dot product.
// LDS: two 128 x 64 fp32 tiles, 16-byte slots XOR-swizzled by the row (conflict-free for the column-run reads of 16
// consecutive rows and for the row reads of 32 consecutive columns).
constexpr int MT = 128;                                // padded sequence length of the MFMA kernels
__device__ __forceinline__ int mha_slot(int row, int c4) { return row * MHA_D + ((c4 ^ (row & 15)) << 2); }
__device__ __forceinline__ int mha_at(int row, int col) { return mha_slot(row, col >> 2) + (col & 3); }
__device__ __forceinline__ int mha_acc_row(int r, int half) { return 8 * (r >> 2) + (r & 3) + 4 * half; }

// rows [0, T) of head h of clip b -> a swizzled LDS tile (rows >= T zero)
__device__ __forceinline__ void mha_stage(float* __restrict__ dst, const float* __restrict__ src, long row0, int h, int T) {
#pragma unroll
    for (int i = 0; i < MT * MHA_D / 4 / 256; ++i) {
        const int e = threadIdx.x + 256 * i, row = e >> 4, c4 = e & 15;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < T) t = *reinterpret_cast<const float4*>(src + (row0 + row) * MHA_LD + h * MHA_D + c4 * 4);
        *reinterpret_cast<float4*>(dst + mha_slot(row, c4)) = t;
    }
}
// this lane's operand run of its own row: src[row][32 half .. 32 half + 31] * mul (row >= T: zeros)
__device__ __forceinline__ void mha_own(float (&f)[32], const float* __restrict__ src, long row0, int h, int row, int half, int T, float mul) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < T) t = *reinterpret_cast<const float4*>(src + (row0 + row) * MHA_LD + h * MHA_D + 32 * half + 4 * c);
        f[4 * c] = t.x * mul; f[4 * c + 1] = t.y * mul; f[4 * c + 2] = t.z * mul; f[4 * c + 3] = t.w * mul;
    }
}
// acc += tile[trow0 + l % 32][.] . own[.]  (M = the tile's 32 rows, N = the wave's 32 own rows, K = 64)
__device__ __forceinline__ void mha_dot(floatx16& acc, const float* __restrict__ tile, int trow, int half, const float (&own)[32]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(tile + mha_slot(trow, 8 * half + c));
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, own[4 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, own[4 * c + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, own[4 * c + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, own[4 * c + 3], acc, 0, 0, 0);
    }
}
// out[dt] += w^T-as-A . tile rows:  out[own row][d] += sum over the 32 tile rows of w[r] * tile[trow0 + row(r, half)][d];
// the two output tiles INTERLEAVE the columns (lane l31 of tile dt owns column 2 l31 + dt), so one 8-byte LDS read feeds both
// MFMAs of a k-step and the epilogues store float2
__device__ __forceinline__ void mha_apply(floatx16 (&out)[2], const floatx16& w, const float* __restrict__ tile, int trow0, int half, int l31) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = trow0 + mha_acc_row(r, half);
        const float2 b = *reinterpret_cast<const float2*>(tile + mha_at(row, 2 * l31));      // columns 2 l31 and 2 l31 + 1: one read for both tiles
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            out[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], dt ? b.y : b.x, out[dt], 0, 0, 0);
    }
}
__device__ __forceinline__ float mha_xor32(float v) { return __shfl_xor(v, 32, 64); }
// The attention-dropout mask reaches the MFMA kernels as BITS: mha_pack_mask_kernel turns the bytes [8 B][T][T] into
// rows[hb][q] = 4 words of DROPPED bits over the keys (bit j of word w = key 32 w + j) and cols[hb][key] = the same over the
// queries, once per forward pass (one coalesced sweep of the 32 MB at B = 256); a lane fetches the 16 bytes of its own query
// (key) and reads the bit of accumulator register r of tile t with a constant shift.  Byte loads from the kernels themselves
// (64 per lane, one query row per lane: 64 cache lines per instruction) cost 70 us per forward launch and 170 us per backward.
__global__ __launch_bounds__(256) void mha_pack_mask_kernel(const unsigned char* __restrict__ keep, int T, unsigned* __restrict__ bits,
                                                            long nhb) {
    __shared__ __attribute__((aligned(16))) unsigned mk32[MT * MT / 4 + 2];
    unsigned char* const mk8 = reinterpret_cast<unsigned char*>(mk32);
    const long hb = blockIdx.x;
    const unsigned char* src = keep + hb * T * T;
    // the T*T bytes of this (head, clip) start at an arbitrary byte: whole dwords from the first aligned address on, the
    // (<= 3) bytes in front of it and behind the last whole dword one by one; LDS keeps the global byte phase
    const int n = T * T, off = (int)(reinterpret_cast<unsigned long long>(src) & 3), head = (4 - off) & 3;
    const int nd = (n - head) >> 2;
    const unsigned* src32 = reinterpret_cast<const unsigned*>(src + head);
    unsigned* dst32 = mk32 + ((off + head) >> 2);
#pragma unroll 8
    for (int e = threadIdx.x; e < nd; e += 256) dst32[e] = src32[e];
    if ((int)threadIdx.x < head) mk8[off + threadIdx.x] = src[threadIdx.x];
    const int tail0 = head + 4 * nd;
    if (tail0 + (int)threadIdx.x < n) mk8[off + tail0 + threadIdx.x] = src[tail0 + threadIdx.x];
    __syncthreads();
    const unsigned char* mk = mk8 + off;
    // thread t < T: the key bits of query t; thread 128 + t: the query bits of key t
    const int t = threadIdx.x & 127, cols = threadIdx.x >> 7;
    if (t >= T) return;
    const int sj = cols ? T : 1, st = cols ? 1 : T;
    unsigned w[4];
#pragma unroll
    for (int wi = 0; wi < 4; ++wi) {
        unsigned acc = 0u;
#pragma unroll 8
        for (int jj = 0; jj < 32; ++jj) {
            const int j = 32 * wi + jj;
            if (j < T) acc |= (mk[t * st + j * sj] ? 0u : 1u) << jj;
        }
        w[wi] = acc;
    }
    *reinterpret_cast<uint4*>(bits + ((cols ? nhb + hb : hb) * T + t) * 4) = make_uint4(w[0], w[1], w[2], w[3]);
}
// this lane's four words, pre-shifted by its accumulator half: bit (8 (r / 4) + r % 4) of word t = "position (t, r) is dropped"
__device__ __forceinline__ void mha_mask_words(unsigned (&w)[4], const unsigned* __restrict__ bits, long row, int half) {
    w[0] = w[1] = w[2] = w[3] = 0u;
    if (!bits) return;
    const uint4 v = *reinterpret_cast<const uint4*>(bits + row * 4);
    w[0] = v.x >> (4 * half); w[1] = v.y >> (4 * half); w[2] = v.z >> (4 * half); w[3] = v.w >> (4 * half);
}
__device__ __forceinline__ bool mha_is_dropped(const unsigned (&w)[4], int t, int r) { return (w[t] >> (8 * (r >> 2) + (r & 3))) & 1u; }

__global__ __launch_bounds__(256, 2) void mha_fwd_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const unsigned* __restrict__ bits,
                                                           float inv_keep, int B, int T, float* __restrict__ o,
                                                           float* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float Ks[MT * MHA_D], Vs[MT * MHA_D];
    const int b = blockIdx.x / MHA_H, h = blockIdx.x % MHA_H;
    const long row0 = (long)b * T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int qi = 32 * wave + l31;                    // this lane's query
    mha_stage(Ks, k, row0, h, T);
    mha_stage(Vs, v, row0, h, T);
    float qf[32];
    mha_own(qf, q, row0, h, qi, half, T, 0.125f);      // 1 / sqrt(64)
    unsigned drop[4];
    mha_mask_words(drop, bits, ((long)h * B + b) * T + min(qi, T - 1), half);
    __syncthreads();
    if (32 * wave >= T) return;                        // a wave without queries (after the only barrier)
    floatx16 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
        mha_dot(s[kt], Ks, 32 * kt + l31, half, qf);
    }
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (32 * kt + mha_acc_row(r, half) >= T) s[kt][r] = -INFINITY;
            m = fmaxf(m, s[kt][r]);
        }
    m = fmaxf(m, mha_xor32(m));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __expf(s[kt][r] - m);
            l += e;
            s[kt][r] = mha_is_dropped(drop, kt, r) ? 0.f : e;
        }
    l += mha_xor32(l);
    floatx16 acc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) mha_apply(acc, s[kt], Vs, 32 * kt, half, l31);
    // accumulator rows are queries 32 wave + row(r, half): their 1 / l sits in the lane of that index
    const float rl = inv_keep / l;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qr = mha_acc_row(r, half);
        const float sc = __shfl(rl, qr, 64);
        if (32 * wave + qr < T) {
            *reinterpret_cast<float2*>(o + (row0 + 32 * wave + qr) * MHA_LD + h * MHA_D + 2 * l31) = make_float2(acc[0][r] * sc, acc[1][r] * sc);
        }
    }
    if (stats && half == 0 && qi < T) {
        float* st = stats + (((long)b * MHA_H + h) * T + qi) * 4;
        st[0] = m; st[1] = l;
    }
}

// backward, query side: D_i and dQ_i
__global__ __launch_bounds__(256, 2) void mha_bwd_q_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, const float* __restrict__ o,
                                                             const float* __restrict__ go, const unsigned* __restrict__ bits,
                                                             float inv_keep, int B, int T, float* __restrict__ stats,
                                                             float* __restrict__ gq) {
    __shared__ __attribute__((aligned(16))) float Ks[MT * MHA_D], Vs[MT * MHA_D];
    const int b = blockIdx.x / MHA_H, h = blockIdx.x % MHA_H;
    const long row0 = (long)b * T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int qi = 32 * wave + l31;
    mha_stage(Ks, k, row0, h, T);
    mha_stage(Vs, v, row0, h, T);
    float qf[32], gf[32];
    mha_own(qf, q, row0, h, qi, half, T, 0.125f);
    mha_own(gf, go, row0, h, qi, half, T, 1.0f);
    float Di = 0.f;
    if (qi < T) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 t = *reinterpret_cast<const float4*>(o + (row0 + qi) * MHA_LD + h * MHA_D + 32 * half + 4 * c);
            Di += gf[4 * c] * t.x + gf[4 * c + 1] * t.y + gf[4 * c + 2] * t.z + gf[4 * c + 3] * t.w;
        }
    }
    Di += mha_xor32(Di);
    float m = 0.f, rl = 0.f;
    if (qi < T) {
        float* st = stats + (((long)b * MHA_H + h) * T + qi) * 4;
        m = st[0]; rl = 1.f / st[1];
        if (half == 0) st[2] = Di;
    }
    unsigned drop[4];
    mha_mask_words(drop, bits, ((long)h * B + b) * T + min(qi, T - 1), half);
    __syncthreads();
    if (32 * wave >= T) return;
    floatx16 acc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        floatx16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
        mha_dot(s, Ks, 32 * kt + l31, half, qf);
        mha_dot(dp, Vs, 32 * kt + l31, half, gf);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(s[r] - m) * rl;
            const float dpk = mha_is_dropped(drop, kt, r) ? 0.f : dp[r] * inv_keep;
            s[r] = p * (dpk - Di) * 0.125f;            // keys >= T: K rows are zero, the product adds nothing
        }
        mha_apply(acc, s, Ks, 32 * kt, half, l31);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qr = 32 * wave + mha_acc_row(r, half);
        if (qr < T) {
            *reinterpret_cast<float2*>(gq + (row0 + qr) * MHA_LD + h * MHA_D + 2 * l31) = make_float2(acc[0][r], acc[1][r]);
        }
    }
}

// backward, key side: dK_j and dV_j (a wave owns 32 keys; Q and dO tiles in LDS, row statistics of the queries beside them)
__global__ __launch_bounds__(256, 2) void mha_bwd_kv_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ go,
                                                              const unsigned* __restrict__ bits, float inv_keep, int B, int T,
                                                              const float* __restrict__ stats, float* __restrict__ gk,
                                                              float* __restrict__ gv) {
    extern __shared__ __attribute__((aligned(16))) float mha_dyn[];
    float* const Qs = mha_dyn;
    float* const Gs = mha_dyn + MT * MHA_D;
    float* const Ms = mha_dyn + 2 * MT * MHA_D;        // [3][MT]: m, 1 / l (0 for rows >= T), D
    const int b = blockIdx.x / MHA_H, h = blockIdx.x % MHA_H;
    const long row0 = (long)b * T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int kj = 32 * wave + l31;                    // this lane's key
    mha_stage(Qs, q, row0, h, T);
    mha_stage(Gs, go, row0, h, T);
    if (threadIdx.x < MT) {
        float m = 0.f, rl = 0.f, D = 0.f;
        if ((int)threadIdx.x < T) {
            const float* st = stats + (((long)b * MHA_H + h) * T + threadIdx.x) * 4;
            m = st[0]; rl = 1.f / st[1]; D = st[2];
        }
        Ms[threadIdx.x] = m; Ms[MT + threadIdx.x] = rl; Ms[2 * MT + threadIdx.x] = D;
    }
    float kf[32], vf[32];
    mha_own(kf, k, row0, h, kj, half, T, 0.125f);
    mha_own(vf, v, row0, h, kj, half, T, 1.0f);
    unsigned drop[4];
    mha_mask_words(drop, bits, ((long)MHA_H * B + (long)h * B + b) * T + min(kj, T - 1), half);      // the query bits of this key
    __syncthreads();
    if (32 * wave >= T) return;
    floatx16 ak[2], av[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ak[dt][r] = av[dt][r] = 0.f;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        if (32 * qt >= T) break;                       // wave-uniform
        floatx16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
        mha_dot(s, Qs, 32 * qt + l31, half, kf);       // rows = queries, column = this lane's key
        mha_dot(dp, Gs, 32 * qt + l31, half, vf);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = 32 * qt + mha_acc_row(r, half);
            const float p = __expf(s[r] - Ms[qr]) * Ms[MT + qr];             // 0 for query rows >= T
            const bool kept = !mha_is_dropped(drop, qt, r);
            s[r] = p * ((kept ? dp[r] * inv_keep : 0.f) - Ms[2 * MT + qr]) * 0.125f;   // d(q_i . k_j), q unscaled
            dp[r] = kept ? p * inv_keep : 0.f;                              // dropped attention weight
        }
        mha_apply(av, dp, Gs, 32 * qt, half, l31);
        mha_apply(ak, s, Qs, 32 * qt, half, l31);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int kr = 32 * wave + mha_acc_row(r, half);
        if (kr < T) {
            const long off = (row0 + kr) * MHA_LD + h * MHA_D + 2 * l31;
            *reinterpret_cast<float2*>(gk + off) = make_float2(ak[0][r], ak[1][r]);
            *reinterpret_cast<float2*>(gv + off) = make_float2(av[0][r], av[1][r]);
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

SED_API long sed_mha_mask_words(int B, int T) { return (B > 0 && T > 0 && T <= MT) ? 2L * MHA_H * B * T * 4 : 0; }

SED_API int sed_mha_fwd(const float* q, const float* k, const float* v, const unsigned char* keep, float p_drop, int B, int T,
                        float* o, float* stats, unsigned* keep_bits, hipStream_t stream) {
    if (B <= 0 || T <= 0 || p_drop < 0.f || p_drop >= 1.f || (long)B * MHA_H >= (1L << 31)) return SED_EINVAL;
    const float ik = keep ? 1.f / (1.f - p_drop) : 1.f;
    if (T <= MT) {
        if (keep && !keep_bits) return SED_EINVAL;
        if (keep) hipLaunchKernelGGL(mha_pack_mask_kernel, dim3(B * MHA_H), dim3(256), 0, stream, keep, T, keep_bits, (long)B * MHA_H);
        hipLaunchKernelGGL(mha_fwd_mfma_kernel, dim3(B * MHA_H), dim3(256), 0, stream, q, k, v, keep ? keep_bits : nullptr, ik, B, T, o,
                           stats);
    } else hipLaunchKernelGGL(mha_fwd_kernel, dim3(B * MHA_H), dim3(128), 0, stream, q, k, v, keep, ik, B, T, o, stats);
    SED_LAUNCH_CHECK();
    return 0;
}

SED_API int sed_mha_bwd(const float* q, const float* k, const float* v, const float* o, const float* g_o,
                        const unsigned char* keep, float p_drop, int B, int T, float* stats, float* g_q, float* g_k, float* g_v,
                        const unsigned* keep_bits, hipStream_t stream) {
    if (B <= 0 || T <= 0 || !stats || p_drop < 0.f || p_drop >= 1.f || (long)B * MHA_H >= (1L << 31)) return SED_EINVAL;
    const float ik = keep ? 1.f / (1.f - p_drop) : 1.f;
    if (T <= MT) {
        if (keep && !keep_bits) return SED_EINVAL;
        const unsigned* bits = keep ? keep_bits : nullptr;
        constexpr int kv_lds = (2 * MT * MHA_D + 3 * MT) * (int)sizeof(float);      // 65.5 KB: above the static limit
        static int raised_dev = -1;                    // the attribute belongs to (function, device)
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return SED_EINVAL;
        if (dev != raised_dev) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mha_bwd_kv_mfma_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kv_lds);
            if (e != hipSuccess) return (int)e;
            raised_dev = dev;
        }
        hipLaunchKernelGGL(mha_bwd_q_mfma_kernel, dim3(B * MHA_H), dim3(256), 0, stream, q, k, v, o, g_o, bits, ik, B, T, stats, g_q);
        hipLaunchKernelGGL(mha_bwd_kv_mfma_kernel, dim3(B * MHA_H), dim3(256), kv_lds, stream, q, k, v, g_o, bits, ik, B, T, stats,
                           g_k, g_v);
    } else {
        hipLaunchKernelGGL(mha_bwd_q_kernel, dim3(B * MHA_H), dim3(128), 0, stream, q, k, v, o, g_o, keep, ik, B, T, stats, g_q);
        hipLaunchKernelGGL(mha_bwd_kv_kernel, dim3(B * MHA_H), dim3(128), 0, stream, q, k, v, g_o, keep, ik, B, T, stats, g_k, g_v);
    }
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

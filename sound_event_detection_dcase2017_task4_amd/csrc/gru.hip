// nn.GRU(512, 256, bidirectional) recurrence (reference models.py:529-530, :565-567) as ONE persistent launch per
// pass for BOTH directions: hidden projection h_prev x W_hh^T + the gate math for all T steps, instead of
// T x (a 128x128-tile GEMM launch on 8-24 workgroups, 10-19 us, + a gate launch + two dependent-launch gaps).
//
// The recurrence is latency-bound (125 dependent steps of 0.1 GFLOP per direction), so the kernel is laid out for a
// short per-step critical path, not for MFMA utilisation:
//  * workgroup = 32 batch rows x 32 hidden units (x 3 gates) of one direction -> (H/32) x ceil(B/32) x 2 = 128
//    workgroups at B=256, all co-resident (1 per CU); its 8 waves split the K reduction;
//  * the workgroup's slice of W_hh (96 rows x 256) lives in REGISTERS for the whole sequence (48 VGPRs per lane) as
//    split-f16 operands (round 4): w = (hi + lo) / s with a power-of-two scale per (gate, hidden unit, wave) taken from the
//    lane pair's own 32 values; per step the 32 x 256 h_prev block is loaded, split the same way (|h| <= 1: fixed scale 2^13)
//    and multiplied as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 -- exact products, fp32 accumulation, the error of an
//    fp32 dot product (as in csrc/conv_sf16.hip) at 18 MFMAs of 32 cycles per wave and step instead of 48 fp32 MFMAs of 64
//    (the MFMA k index is permuted consistently for both operands, which a dot product allows, so every lane reads
//    contiguous float4 runs);
//  * a step of direction d / row block rb depends only on the 8 workgroups (hidden blocks) of the same (d, rb).  They hand
//    over THROUGH THE DATA (round 4): the exchange buffers (h_t forward, dgh_t backward) are pre-filled with a sentinel NaN
//    by the launcher, producers write their values with agent-scope atomic stores, and every consumer wave polls its own
//    operand block with `sc1` (agent-scope) loads until no sentinel is left.  Per step that is one store-to-load trip to the coherence
//    point; the counter protocol of rounds 1-3 (stores -> wait -> barrier -> atomic add -> spin on the counter -> barrier ->
//    loads) was three such trips in a row.  The members of a group have workgroup ids x + 8 m (gru_group_of): the round-robin
//    dispatcher puts them on ONE XCD, whose L2 is then the coherence point of the exchange: producers use PLAIN stores (they land in that L2, dirty) and the consumers' `sc1` loads -- which bypass the
//    vector L1 -- find them there: 1.46 us per step of a bare 8 x 4 KB all-gather instead of 2.39 us with agent-scope stores
//    that write through to the memory side (tools/xcd_exchange_probe.hip).  Dirty lines of one XCD's L2 are invisible to the
//    other seven (the same probe with a group spread over the XCDs never completes), so the kernel does not ASSUME the
//    placement: every workgroup publishes the XCC_ID hardware register it runs on, and a group whose eight members do not all
//    report the same XCD falls back to the agent-scope stores;
//  * everything a step needs that does NOT depend on the previous step (gi / g_out / saved gates) is loaded before
//    the wait.
// The spin is bounded: a wave that never sees its operands (which cannot happen while all workgroups are resident:
// 128 x 512 threads, 48 KB LDS) gives up after ~1 s and raises the error word instead of hanging the GPU.
#include "common.h"
#include "sed_hip.h"
SED_OBJECT_FLAGS(gru)

namespace {

constexpr int GH = 256;                                // hidden size the kernels are built for
constexpr int GRU_FLAG_INTS = 1024;                    // workspace ints: the error word lives at [GRU_MAX_GROUPS]
constexpr int GRU_MAX_GROUPS = GRU_FLAG_INTS - 1;

__device__ __forceinline__ float gru_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// accumulator register r of a 32x32 MFMA tile holds row (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), column lane & 31
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Data exchanged between workgroups DURING the kernel (h_t forward, dgh_t backward) is written and read with
// agent-scope relaxed atomics (64-bit): they go to the coherence point, so the step synchronisation needs no L2
// write-back / invalidate (an agent-scope fence pair cost ~10 us per step on the 8-XCD part: measured).
__device__ __forceinline__ void st_coherent(float* ptr, float2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(ptr), __builtin_bit_cast(unsigned long long, v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_xcd_local(float* ptr, float2 v) {       // lands in this XCD's L2; see gru_group_on_one_xcd
    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ void st_exchange(float* ptr, float2 v, bool one_xcd) {
    if (one_xcd) st_xcd_local(ptr, v);
    else st_coherent(ptr, v);
}
// 16-byte coherent load (agent-scope cache policy: `sc1`, 3 % faster than `sc0 sc1` here; atomicity is not needed: the data was completed before the
// sentinel disappeared -- each float is checked on its own).  The result is valid only after ld_coherent_wait on the same registers.
typedef float f4r __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ld_coherent4(f4r& dst, const float* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(ptr) : "memory");
}
template <int N>
__device__ __forceinline__ void ld_coherent_wait(f4r (&v)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// Hand-over through the data: every exchange buffer is pre-filled with this quiet NaN (a payload no arithmetic produces; a NaN
// that training itself produces carries the default payload 0x7fc00000) and a consumer's operand block is complete when
// none of its words is the sentinel.
constexpr unsigned GRU_SENTINEL = 0x7fc0deadu;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

// 8 fp32 values (already scaled) -> their (hi, lo) f16 halves as MFMA operands; element i of the operand = value i
__device__ __forceinline__ void gru_split8(const f4r& u, const f4r& v, float s, half8& hi, half8& lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    sed_sf_split2(u[0] * s, u[1] * s, h0, l0);
    sed_sf_split2(u[2] * s, u[3] * s, h1, l1);
    sed_sf_split2(v[0] * s, v[1] * s, h2, l2);
    sed_sf_split2(v[2] * s, v[3] * s, h3, l3);
    const uintx4 h = {h0, h1, h2, h3}, l = {l0, l1, l2, l3};
    hi = __builtin_bit_cast(half8, h);
    lo = __builtin_bit_cast(half8, l);
}
__device__ __forceinline__ float gru_amax4(const f4r& v) { return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))); }
__device__ __forceinline__ bool gru_has_sentinel(const f4r& v) {
    return __float_as_uint(v[0]) == GRU_SENTINEL || __float_as_uint(v[1]) == GRU_SENTINEL ||
           __float_as_uint(v[2]) == GRU_SENTINEL || __float_as_uint(v[3]) == GRU_SENTINEL;
}
// the three exact partial products of one K = 16 slab, small terms first
__device__ __forceinline__ void gru_mfma3(floatx16& acc, const half8& ahi, const half8& alo, const half8& bhi, const half8& blo) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
}

// Poll this wave's operand block (N x 16 bytes per lane at `ap`) until it holds no sentinel.  Bounded: after `limit` polls,
// or when another wave has given up, the error word is raised and the block is taken as it is (the launcher's check kernel
// then overwrites the pass's output with NaN).  `dead` is sticky per wave: a pass that failed stops polling.
template <int N>
__device__ __forceinline__ void gru_poll(f4r (&a)[N], const float* ap, int* err, long limit, bool& dead) {
    long polls = 0;
    for (;;) {
        if (N > 4) {                                   // a wide block: watch ONE piece until it arrives, then fetch the block
            for (;;) {
                f4r one[1];
                ld_coherent4(one[0], ap + 4 * (N - 1));
                ld_coherent_wait(one);
                if (dead || __builtin_amdgcn_ballot_w64(gru_has_sentinel(one[0])) == 0) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++polls & 63) == 0 || polls >= limit) {
                    if (polls >= limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        dead = true;
                        break;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < N; ++q) ld_coherent4(a[q], ap + 4 * q);
        ld_coherent_wait(a);
        bool bad = false;
#pragma unroll
        for (int q = 0; q < N; ++q) bad |= gru_has_sentinel(a[q]);
        if (dead || __builtin_amdgcn_ballot_w64(bad) == 0) return;
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 63) == 0 || polls >= limit) {
            if (polls >= limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // give up everywhere, never hang
                dead = true;
                return;
            }
        }
    }
}

// Do the `members` workgroups of this group (linear ids group + ngroups * m) run on ONE XCD?  Each publishes the XCC_ID it reads
// from the hardware register (+ 1; the launcher zeroed the table) with an agent-scope store and reads the others' the same way,
// so every member sees the same eight values and takes the same decision.  Bounded like every wait of these kernels; a group
// that cannot tell (or a failed pass) uses the agent-scope stores, which are right for any placement.
// Workgroup id -> (group, member): the dispatcher deals workgroup i to XCD i % 8, so the 8 members of a group are the ids
// x + 8 m (m = 0..7) of one block of 64 ids: group = x + 8 * (id / 64).  The grid is padded to whole blocks of 64; workgroups of
// groups that do not exist leave at once.
__host__ __device__ __forceinline__ int gru_group_of(int id) { return (id & 7) + 8 * (id >> 6); }
__host__ __device__ __forceinline__ int gru_member_of(int id) { return (id >> 3) & 7; }
__host__ __device__ __forceinline__ int gru_grid(int ngroups) { return 64 * ((ngroups + 7) / 8); }
static_assert(GH / 32 == 8, "a group is 8 hidden blocks");
constexpr int GRU_XCC_TABLE = 256;                     // flags[GRU_XCC_TABLE + workgroup id]
__device__ __forceinline__ bool gru_group_on_one_xcd(int* flags, int group, long limit, int* lds_word) {
    constexpr int members = GH / 32;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) {
            unsigned id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
            __hip_atomic_store(flags + GRU_XCC_TABLE + blockIdx.x, (int)(id & 0xf) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int v = 0;
        bool ok = false;
        for (long polls = 0; polls < limit; ++polls) {
            if (lane < members) v = __hip_atomic_load(flags + GRU_XCC_TABLE + (group & 7) + 8 * lane + 64 * (group >> 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = __builtin_amdgcn_ballot_w64(lane < members && v == 0) == 0;
            if (ok) break;
            __builtin_amdgcn_s_sleep(1);
        }
        const int v0 = __shfl(v, 0, 64);
        const bool same = ok && __builtin_amdgcn_ballot_w64(lane < members && v != v0) == 0;
        if (lane == 0) *lds_word = same ? 1 : 0;
    }
    __syncthreads();
    const bool r = *lds_word != 0;
    __syncthreads();
    return r;
}

struct GruSeqFwdP {
    const float* gi;           // [B][T][6H]: input projections incl. b_ih, forward gates then reverse gates
    const float* w[2];         // W_hh [3H][H]
    const float* bhh[2];       // [3H]
    float* hs;                 // [2][T][B][H]
    float* saves;              // [2][T][B][4H] = r, z, n, gh_n
    float* out;                // [B][T][2H]
    int* flags;                // workspace (zeroed by the launcher): the error word at [GRU_MAX_GROUPS]
    int B, T, ngroups;
    long spin_limit;
};

__global__ __launch_bounds__(512) void gru_seq_fwd_kernel(GruSeqFwdP p) {
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * 16 * 64];      // 48 KB
    const int group = gru_group_of(blockIdx.x), jb = gru_member_of(blockIdx.x);
    if (group >= p.ngroups) return;                    // filler workgroups of the XCD-aligned grid (whole workgroup, before any barrier)
    const int d = group & 1, r0 = (group >> 1) * 32, j0 = jb * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hf = lane >> 5, l31 = lane & 31;
    const int B = p.B, T = p.T;
    const long bh = (long)B * GH;
    int* const err = p.flags + GRU_MAX_GROUPS;
    const bool one_xcd = gru_group_on_one_xcd(p.flags, group, p.spin_limit > 8 ? (p.spin_limit >> 3) : 1, reinterpret_cast<int*>(red));

    // this lane's part of the weight slice, resident for the whole sequence: gate g, hidden unit j0 + l31, 16 consecutive k,
    // as split-f16 operands scaled by a power of two of the lane pair's own amax (the pair feeds one accumulator column)
    const int kb = wave * 32 + hf * 16;
    half8 whi[3][2], wlo[3][2];
    float winv[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const f4r* bp = reinterpret_cast<const f4r*>(p.w[d] + (long)(g * GH + j0 + l31) * GH + kb);
        const f4r b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        float am = fmaxf(fmaxf(gru_amax4(b0), gru_amax4(b1)), fmaxf(gru_amax4(b2), gru_amax4(b3)));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        const float sw = sed_sf_scale_of(am);
        gru_split8(b0, b1, sw, whi[g][0], wlo[g][0]);
        gru_split8(b2, b3, sw, whi[g][1], wlo[g][1]);
        winv[g] = (1.0f / sw) * (1.0f / 8192.0f);       // h_prev is scaled by 2^13 (|h| <= 1)
    }
    const int arow = min(r0 + l31, B - 1);             // operand row of this lane (clamped: ragged last row block)
    // output role: accumulator register r = tid >> 5 of the lane pair (2q, 2q+1), q = tid & 31: two adjacent hidden units
    const int r = tid >> 5, lp = (tid & 31) * 2;
    const int row = r0 + acc_row(r, lp >> 5);
    const bool live = row < B;
    const int rowc = live ? row : B - 1;
    const int j = j0 + (lp & 31);
    float2 bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = *reinterpret_cast<const float2*>(p.bhh[d] + g * GH + j);

    float2 hlast = make_float2(0.f, 0.f);
    bool dead = false;
    const long poll_limit = p.spin_limit > 8 ? (p.spin_limit >> 3) : 1;
    for (int k = 0; k < T; ++k) {
        const int t = d ? T - 1 - k : k;
        const float* h_prev = p.hs + ((long)d * T + (d ? t + 1 : t - 1)) * bh;
        // independent of the previous step: this thread's input-projection values
        const float* gir = p.gi + ((long)rowc * T + t) * 6 * GH + d * 3 * GH + j;
        const float2 gr = *reinterpret_cast<const float2*>(gir);
        const float2 gz = *reinterpret_cast<const float2*>(gir + GH);
        const float2 gn = *reinterpret_cast<const float2*>(gir + 2 * GH);
        float2 gh[3] = {bias[0], bias[1], bias[2]};
        float2 hp = make_float2(0.f, 0.f);
        if (k > 0) {
            f4r a[4];
            gru_poll(a, h_prev + (long)arow * GH + kb, err, poll_limit, dead);
            hp = hlast;                                // this thread wrote h_{t-1}[row][j, j+1] itself
            half8 ahi[2], alo[2];
            gru_split8(a[0], a[1], 8192.0f, ahi[0], alo[0]);
            gru_split8(a[2], a[3], 8192.0f, ahi[1], alo[1]);
            floatx16 acc[3];
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 3; ++g) gru_mfma3(acc[g], ahi[m], alo[m], whi[g][m], wlo[g][m]);
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[g][i] *= winv[g];
            // two-phase reduction over the 8 waves: 4..7 -> LDS -> added by 0..3 -> LDS -> summed by the output threads
            if (wave >= 4) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) red[(((wave - 4) * 3 + g) * 16 + i) * 64 + lane] = acc[g][i];
            }
            __syncthreads();
            if (wave < 4) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[g][i] += red[((wave * 3 + g) * 16 + i) * 64 + lane];
            }
            __syncthreads();
            if (wave < 4) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) red[((wave * 3 + g) * 16 + i) * 64 + lane] = acc[g][i];
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float2 v = *reinterpret_cast<const float2*>(&red[((w * 3 + g) * 16 + r) * 64 + lp]);
                    gh[g].x += v.x; gh[g].y += v.y;
                }
        }
        if (live) {
            float2 rr, zz, nn, hh;
            rr.x = gru_sigmoid(gr.x + gh[0].x); rr.y = gru_sigmoid(gr.y + gh[0].y);
            zz.x = gru_sigmoid(gz.x + gh[1].x); zz.y = gru_sigmoid(gz.y + gh[1].y);
            nn.x = tanhf(gn.x + rr.x * gh[2].x); nn.y = tanhf(gn.y + rr.y * gh[2].y);
            hh.x = (1.0f - zz.x) * nn.x + zz.x * hp.x; hh.y = (1.0f - zz.y) * nn.y + zz.y * hp.y;
            st_exchange(p.hs + ((long)d * T + t) * bh + (long)row * GH + j, hh, one_xcd);      // the hand-over: replaces the sentinel
            hlast = hh;
            *reinterpret_cast<float2*>(p.out + ((long)row * T + t) * 2 * GH + d * GH + j) = hh;
            float* s = p.saves + ((long)d * T + t) * 4 * bh + (long)row * 4 * GH + j;
            *reinterpret_cast<float2*>(s) = rr;
            *reinterpret_cast<float2*>(s + GH) = zz;
            *reinterpret_cast<float2*>(s + 2 * GH) = nn;
            *reinterpret_cast<float2*>(s + 3 * GH) = gh[2];
        }
        if (k > 0 && k + 1 < T) __syncthreads();       // `red` is rewritten by the next step
    }
}

struct GruSeqBwdP {
    const float* g_out;        // [B][T][2H]
    const float* wt[2];        // W_hh^T [H][3H]
    const float* hs;           // [2][T][B][H]
    const float* saves;        // [2][T][B][4H]
    float* dgi;                // [B][T][6H]
    float* dgh;                // [2][T][B][3H]
    float* dbias;              // nullable: [2 directions][row blocks][4: dr, dz, dn, dn*r][H] sums over (t, the block's rows)
    float* dgi_amax;           // nullable: amax slots of |dgi| (the operand scale of the split-f16 input-gradient GEMM)
    int* flags;
    int B, T, ngroups;
    long spin_limit;
};

__global__ __launch_bounds__(512) void gru_seq_bwd_kernel(GruSeqBwdP p) {
    __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];          // 16 KB
    const int group = gru_group_of(blockIdx.x), jb = gru_member_of(blockIdx.x);
    if (group >= p.ngroups) return;
    const int d = group & 1, r0 = (group >> 1) * 32, j0 = jb * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hf = lane >> 5, l31 = lane & 31;
    const int B = p.B, T = p.T;
    const long bh = (long)B * GH;
    int* const err = p.flags + GRU_MAX_GROUPS;

    // dh_gemm[b][j] = sum_c dgh_later[b][c] * W_hh[c][j], c over 3H = 768: 96 per wave, 48 consecutive per lane;
    // this lane's run of row j0 + l31 of W_hh^T is resident for the whole sequence as split-f16 operands (scale: a power of two
    // of the lane pair's own amax -- the pair feeds one accumulator column)
    const int kb = wave * 96 + hf * 48;
    half8 whi[6], wlo[6];
    float winv;
    {
        const f4r* bp = reinterpret_cast<const f4r*>(p.wt[d] + (long)(j0 + l31) * 3 * GH + kb);
        f4r b[12];
        float am = 0.f;
#pragma unroll
        for (int q = 0; q < 12; ++q) { b[q] = bp[q]; am = fmaxf(am, gru_amax4(b[q])); }
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        const float sw = sed_sf_scale_of(am);
#pragma unroll
        for (int m = 0; m < 6; ++m) gru_split8(b[2 * m], b[2 * m + 1], sw, whi[m], wlo[m]);
        winv = 1.0f / sw;
    }
    const int arow = min(r0 + l31, B - 1);
    const int r = tid >> 5, lp = (tid & 31) * 2;
    const int row = r0 + acc_row(r, lp >> 5);
    const bool live = row < B;
    const int rowc = live ? row : B - 1;
    const int j = j0 + (lp & 31);
    float2 dhz = make_float2(0.f, 0.f);                // dh * z of the step processed before (the later time step)
    // bias gradients (db_ih = column sums of dgi, db_hh = of dgh) accumulate here over the time steps: the two column-sum
    // passes over dgi / dgh (0.44 ms per step at B = 256) disappear
    float2 sb_r = make_float2(0.f, 0.f), sb_z = sb_r, sb_n = sb_r, sb_nr = sb_r;
    float gmax = 0.f;                                  // max |dgi| this thread wrote
    bool dead = false;
    const long poll_limit = p.spin_limit > 8 ? (p.spin_limit >> 3) : 1;

    for (int k = T - 1, done = 0; k >= 0; --k, ++done) {
        const int t = d ? T - 1 - k : k;
        // independent of the later step: output gradient, saved gates, previous hidden state
        float2 dh = *reinterpret_cast<const float2*>(p.g_out + ((long)rowc * T + t) * 2 * GH + d * GH + j);
        const float* s = p.saves + ((long)d * T + t) * 4 * bh + (long)rowc * 4 * GH + j;
        const float2 rr = *reinterpret_cast<const float2*>(s), zz = *reinterpret_cast<const float2*>(s + GH);
        const float2 nn = *reinterpret_cast<const float2*>(s + 2 * GH), ghn = *reinterpret_cast<const float2*>(s + 3 * GH);
        float2 hp = make_float2(0.f, 0.f);
        if (k > 0) hp = *reinterpret_cast<const float2*>(p.hs + ((long)d * T + (d ? t + 1 : t - 1)) * bh + (long)rowc * GH + j);
        dh.x += dhz.x; dh.y += dhz.y;
        if (done > 0) {
            const float* dgh_later = p.dgh + ((long)d * T + (d ? t - 1 : t + 1)) * 3 * bh;
            f4r a[12];
            gru_poll(a, dgh_later + (long)arow * 3 * GH + kb, err, poll_limit, dead);
            // gradients have no fixed range: one power-of-two scale per wave and step from the amax of its 32 x 96 block
            float am = 0.f;
#pragma unroll
            for (int q = 0; q < 12; ++q) am = fmaxf(am, gru_amax4(a[q]));
            const float sa = sed_sf_scale_of(wave_max(am));
            floatx16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                half8 ahi, alo;
                gru_split8(a[2 * m], a[2 * m + 1], sa, ahi, alo);
                gru_mfma3(acc, ahi, alo, whi[m], wlo[m]);
            }
            const float unscale = winv * (1.0f / sa);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= unscale;
            if (wave >= 4) {
#pragma unroll
                for (int i = 0; i < 16; ++i) red[((wave - 4) * 16 + i) * 64 + lane] = acc[i];
            }
            __syncthreads();
            if (wave < 4) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] += red[(wave * 16 + i) * 64 + lane];
            }
            __syncthreads();
            if (wave < 4) {
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float2 v = *reinterpret_cast<const float2*>(&red[(w * 16 + r) * 64 + lp]);
                dh.x += v.x; dh.y += v.y;
            }
        }
    float2 dr_pre, dz_pre, dn_pre, dn_r;
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
        if (live) {
            float* gi_o = p.dgi + ((long)row * T + t) * 6 * GH + d * 3 * GH + j;
            float* gh_o = p.dgh + ((long)d * T + t) * 3 * bh + (long)row * 3 * GH + j;
            // the hand-over first: nothing else is queued in front of it.  Agent-scope stores here although the group shares an XCD:
            // with three exchanged rows per thread and 48 KB polled per workgroup the XCD-local form measured 7 % SLOWER
            // (1.04 vs 0.98 ms at B = 256), the forward pass 15 % faster (0.60 vs 0.71 ms)
            st_coherent(gh_o, dr_pre);
            st_coherent(gh_o + GH, dz_pre);
            st_coherent(gh_o + 2 * GH, dn_r);
            *reinterpret_cast<float2*>(gi_o) = dr_pre;
            *reinterpret_cast<float2*>(gi_o + GH) = dz_pre;
            *reinterpret_cast<float2*>(gi_o + 2 * GH) = dn_pre;
            sb_r.x += dr_pre.x; sb_r.y += dr_pre.y; sb_z.x += dz_pre.x; sb_z.y += dz_pre.y;
            sb_n.x += dn_pre.x; sb_n.y += dn_pre.y; sb_nr.x += dn_r.x; sb_nr.y += dn_r.y;
            gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(dr_pre.x), fabsf(dr_pre.y))),
                         fmaxf(fmaxf(fabsf(dz_pre.x), fabsf(dz_pre.y)), fmaxf(fabsf(dn_pre.x), fabsf(dn_pre.y))));
        }
        if (done > 0 && k > 0) __syncthreads();       // `red` is rewritten by the next step
    }
    if (p.dgi_amax) amax_publish_block(p.dgi_amax, gmax);
    if (p.dbias) {                                     // rows of the block: 16 r x 2 halves per hidden pair -> LDS -> 128 sums
        __syncthreads();
        float* o = red + tid * 8;
        o[0] = sb_r.x; o[1] = sb_r.y; o[2] = sb_z.x; o[3] = sb_z.y; o[4] = sb_n.x; o[5] = sb_n.y; o[6] = sb_nr.x; o[7] = sb_nr.y;
        __syncthreads();
        if (tid < 128) {
            const int kind = tid >> 5, jj = tid & 31;              // hidden unit j0 + jj lives in the threads with (q & 15) == jj / 2
            float acc = 0.f;
#pragma unroll 4
            for (int rr = 0; rr < 16; ++rr)
#pragma unroll
                for (int half = 0; half < 2; ++half)
                    acc += red[(rr * 32 + half * 16 + (jj >> 1)) * 8 + kind * 2 + (jj & 1)];
            p.dbias[(((long)d * (p.ngroups >> 1) + (group >> 1)) * 4 + kind) * GH + j0 + jj] = acc;
        }
    }
}


// ---- run-time failure handling --------------------------------------------------------------------------------------
// The persistent kernels need all their workgroups resident.  If a bounded spin gives up (CU mask, partitioned GPU, a
// co-tenant kernel holding CUs for longer than the bound) the error word in `flags` is set and the results are garbage.
// This follow-up kernel, enqueued right behind the recurrence, makes that LOUD without a host synchronisation: it
// overwrites the pass's output with NaN and raises a flag in host-mapped memory that the host polls at its next call.
__global__ __launch_bounds__(256) void gru_seq_check_kernel(const int* flags, int* err_host, int code, float* out, long n) {
    if (__hip_atomic_load(flags + GRU_MAX_GROUPS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const float nan = __builtin_nanf("");
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = nan;
    if (blockIdx.x == 0 && threadIdx.x == 0 && err_host)
        __hip_atomic_store(err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// test hook: occupies `blocks` whole CUs (one workgroup with all of the CU's LDS each) for `microseconds`
__global__ __launch_bounds__(256) void occupy_kernel(long ticks) {
    extern __shared__ float hog[];
    hog[threadIdx.x] = 0.f;
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

long g_spin_limit = 1L << 23;          // ~1 s of polling

// all workgroups of the persistent kernels fit on the current device at once?  (cached per device)
bool gru_device_fits(int grid) {
    static int cached_dev = -1, cached_cap = 0;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return false;
    if (dev != cached_dev) {
        int cus = 0, per_f = 0, per_b = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_f, gru_seq_fwd_kernel, 512, 0) != hipSuccess) return false;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_b, gru_seq_bwd_kernel, 512, 0) != hipSuccess) return false;
        // one workgroup per CU is what the kernels are laid out for; the occupancy API may over-report by one block per CU
        // (MI355X_MICROARCH.md), so only its ">= 1" answer is used
        cached_cap = (per_f >= 1 && per_b >= 1) ? cus : 0;
        cached_dev = dev;
    }
    return grid <= cached_cap;
}

}  // namespace

SED_API int sed_gru_seq_supported(int B, int Hd) {
    if (!(Hd == GH && B > 0 && 2 * sed_cdiv(B, 32) * (GH / 32) <= 256)) return 0;
    return gru_device_fits(gru_grid(2 * sed_cdiv(B, 32))) ? 1 : 0;
}
SED_API long sed_gru_seq_ws_floats(void) { return GRU_FLAG_INTS; }
SED_API int sed_gru_set_spin_limit(long spins) {
    g_spin_limit = spins > 0 ? spins : (1L << 23);
    return 0;
}
SED_API int sed_debug_occupy(int blocks, int lds_bytes, long microseconds, hipStream_t stream) {
    if (blocks <= 0 || lds_bytes < 1024 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 2000000) return SED_EINVAL;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), lds_bytes, stream, microseconds * 100L);   // 100 MHz wall clock
    SED_LAUNCH_CHECK();
    return 0;
}

// Whole forward recurrence in one launch.  Direction 0 walks t = 0..T-1, direction 1 walks t = T-1..0.
SED_API int sed_gru_seq_fwd(const float* gi, const float* w_hh_f, const float* w_hh_b, const float* b_hh_f,
                            const float* b_hh_b, int B, int T, int Hd, float* hs, float* saves, float* out, float* ws,
                            int* err_host, hipStream_t stream) {
    const int ngroups = 2 * sed_cdiv(B, 32);
    if (B <= 0 || T <= 0 || Hd != GH || ngroups > GRU_MAX_GROUPS || ngroups * (GH / 32) > 256) return SED_EINVAL;
    hipError_t e = hipMemsetAsync(ws, 0, GRU_FLAG_INTS * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    // the hand-over buffer starts out as sentinels; every word of it is replaced by the kernel
    e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(hs), (int)GRU_SENTINEL, (size_t)2 * T * B * GH, stream);
    if (e != hipSuccess) return (int)e;
    GruSeqFwdP p{gi, {w_hh_f, w_hh_b}, {b_hh_f, b_hh_b}, hs, saves, out, reinterpret_cast<int*>(ws), B, T, ngroups, g_spin_limit};
    hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3(gru_grid(ngroups)), dim3(512), 0, stream, p);
    SED_LAUNCH_CHECK();
    hipLaunchKernelGGL(gru_seq_check_kernel, dim3(256), dim3(256), 0, stream, reinterpret_cast<const int*>(ws), err_host, 1, out,
                       (long)B * T * 2 * GH);
    SED_LAUNCH_CHECK();
    return 0;
}

// Whole backward recurrence in one launch (reverse processing order).
SED_API int sed_gru_seq_bwd(const float* g_out, const float* wt_f, const float* wt_b, const float* hs, const float* saves,
                            int B, int T, int Hd, float* dgi, float* dgh, float* dbias_parts, float* ws, int* err_host,
                            float* dgi_amax, hipStream_t stream) {
    const int ngroups = 2 * sed_cdiv(B, 32);
    if (B <= 0 || T <= 0 || Hd != GH || ngroups > GRU_MAX_GROUPS || ngroups * (GH / 32) > 256) return SED_EINVAL;
    hipError_t e = hipMemsetAsync(ws, 0, GRU_FLAG_INTS * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dgh), (int)GRU_SENTINEL, (size_t)2 * T * B * 3 * GH, stream);
    if (e != hipSuccess) return (int)e;
    if (dgi_amax) {
        e = sed_amax_clear(dgi_amax, stream);
        if (e != hipSuccess) return (int)e;
    }
    GruSeqBwdP p{g_out, {wt_f, wt_b}, hs, saves, dgi, dgh, dbias_parts, dgi_amax, reinterpret_cast<int*>(ws), B, T, ngroups, g_spin_limit};
    hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3(gru_grid(ngroups)), dim3(512), 0, stream, p);
    SED_LAUNCH_CHECK();
    hipLaunchKernelGGL(gru_seq_check_kernel, dim3(256), dim3(256), 0, stream, reinterpret_cast<const int*>(ws), err_host, 2, dgi,
                       (long)B * T * 6 * GH);
    SED_LAUNCH_CHECK();
    return 0;
}

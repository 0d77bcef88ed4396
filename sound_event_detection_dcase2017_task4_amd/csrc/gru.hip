// nn.GRU(512, 256, bidirectional) recurrence (reference models.py:529-530, :565-567) as ONE persistent launch per
// pass for BOTH directions: hidden projection h_prev x W_hh^T + the gate math for all T steps, instead of
// T x (a 128x128-tile GEMM launch on 8-24 workgroups, 10-19 us, + a gate launch + two dependent-launch gaps).
//
// The recurrence is latency-bound (125 dependent steps of 0.1 GFLOP per direction), so the kernel is laid out for a
// short per-step critical path, not for MFMA utilisation:
//  * workgroup = 16 batch rows x 64 hidden units (x 3 gates) of one direction -> (H/64) x ceil(B/16) x 2 = 128
//    workgroups at B=256, all co-resident (1 per CU), 4 waves = 16 units each with the whole K reduction (see "the
//    recurrences" below: no cross-wave sum, ONE barrier per step: the shared operand block is double-buffered);
//  * the wave's slice of W_hh (48 rows x 256) lives in REGISTERS for the whole sequence (192 VGPRs per lane) as
//    split-f16 operands (round 4): w = (hi + lo) / s with a power-of-two scale per (gate, hidden unit) taken from the
//    row's own amax; per step the 16 x 256 h_prev block is fetched once per workgroup, split the same way (|h| <= 1: fixed
//    scale 2^13), shared through LDS and multiplied as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 -- exact products,
//    fp32 accumulation, the error of an fp32 dot product (as in csrc/conv_sf16.hip) at 72 MFMAs of 16 cycles per wave and step
//    (the K index is permuted consistently for both operands, which a dot product allows, so the polling loads are 64
//    contiguous bytes per four lanes);
//  * a step of direction d / row block rb depends only on the 4 workgroups (hidden blocks) of the same (d, rb).  They hand
//    over THROUGH THE DATA (round 4): the exchange buffers (h_t forward, dgh_t backward) are pre-filled with a sentinel NaN
//    by the launcher, producers write their values, and every consumer wave polls its own quarter of the
//    operand block with `sc1` (agent-scope) loads until no sentinel is left.  Per step that is one store-to-load trip to the coherence
//    point; the counter protocol of rounds 1-3 (stores -> wait -> barrier -> atomic add -> spin on the counter -> barrier ->
//    loads) was three such trips in a row.  The members of a group have workgroup ids x + 8 m (gru_group_of): the round-robin
//    dispatcher puts them on ONE XCD, whose L2 is then the coherence point of the exchange:
//    producers use PLAIN stores (they land in that L2, dirty) and the consumers' `sc1` loads -- which bypass the
//    vector L1 -- find them there: 1.46 us per step of a bare 8 x 4 KB all-gather among 8 workgroups instead of 2.39 us with agent-scope stores
//    that write through to the memory side (tools/xcd_exchange_probe.hip).  Dirty lines of one XCD's L2 are invisible to the
//    other seven (the same probe with a group spread over the XCDs never completes), so the kernel does not ASSUME the
//    placement: every workgroup publishes the XCC_ID hardware register it runs on, and a group whose four members do not all
//    report the same XCD falls back to the agent-scope stores;
//  * everything a step needs that does NOT depend on the previous step (gi / g_out / saved gates) is loaded before
//    the wait.
// The spin is bounded: a wave that never sees its operands (which cannot happen while all workgroups are resident:
// 128 x 256 threads, 34 / 98 KB LDS) gives up after ~1 s and raises the error word instead of hanging the GPU.
#include "common.h"
#include "sed_hip.h"
#include "sed_hip_test.h"
SED_OBJECT_FLAGS(gru)

namespace {

constexpr int GH = 256;                                // hidden size the kernels are built for
constexpr int GRU_FLAG_INTS = 1024;                    // workspace ints: the error word lives at [GRU_MAX_GROUPS]
constexpr int GRU_MAX_GROUPS = GRU_FLAG_INTS - 1;

__device__ __forceinline__ float gru_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// The gate math sits on the critical path of every step of the persistent forward kernel (0.7 us of 4 per step with expf / tanhf /
// IEEE division for the lane's four elements), so there it runs on the hardware exp2 and reciprocal (1 ulp each): absolute
// error <= 1.5e-7 on values in [-1, 1], exact limits (exp -> inf gives rcp -> 0), NaN propagates.
__device__ __forceinline__ float gru_sigmoid_hw(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float gru_tanh_hw(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

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
// Poll this wave's operand block (N x 16 bytes per lane at `ap`) until it holds no sentinel.  Bounded: after `limit` polls,
// or when another wave has given up, the error word is raised and the block is taken as it is (the launcher's check kernel
// then overwrites the pass's output with NaN).  `dead` is sticky per wave: a pass that failed stops polling.
// watch_one: poll ONE piece of a wide block until it arrives and fetch the block then -- pays behind agent-scope stores (0.84 ->
// 0.81 ms for the backward pass at B = 256), costs behind XCD-local ones (0.71 -> 0.75 ms): on only in the fallback.
template <int N, int STRIDE>
__device__ __forceinline__ void gru_poll(f4r (&a)[N], const float* ap, int* err, long limit, bool& dead, bool watch_one) {
    long polls = 0;
    for (;;) {
        if (N > 4 && watch_one) {                      // a wide block behind agent-scope stores: watch ONE piece until it arrives, then fetch the block
            for (;;) {
                f4r one[1];
                ld_coherent4(one[0], ap + STRIDE * (N - 1));
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
        for (int q = 0; q < N; ++q) ld_coherent4(a[q], ap + STRIDE * q);
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
// x + 8 m (m = 0..3) of one block of 32 ids: group = x + 8 * (id / 32).  The grid is padded to whole blocks of 32; workgroups of
// groups that do not exist leave at once.
__host__ __device__ __forceinline__ int gru_group_of(int id) { return (id & 7) + 8 * (id >> 5); }
__host__ __device__ __forceinline__ int gru_member_of(int id) { return (id >> 3) & 3; }
__host__ __device__ __forceinline__ int gru_grid(int ngroups) { return 32 * ((ngroups + 7) / 8); }
constexpr int GRU_XCC_TABLE = 256;                     // flags[GRU_XCC_TABLE + workgroup id]
__device__ __forceinline__ bool gru_group_on_one_xcd(int* flags, int group, long limit, int* lds_word) {
    constexpr int members = 4;
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
            if (lane < members) v = __hip_atomic_load(flags + GRU_XCC_TABLE + (group & 7) + 8 * lane + 32 * (group >> 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// ---- the recurrences --------------------------------------------------------------------------------------------------------
// Workgroup = 16 batch rows x 64 hidden units of one direction, 4 waves; a wave owns 16 units for ALL gates and the WHOLE K
// reduction, so the gate math follows the MFMAs in registers (rounds 1-3 split K over 8 waves and summed 48 accumulator
// registers per wave through LDS behind three barriers).  Transposed product on v_mfma_f32_16x16x32_f16: A = the wave's
// weight rows (resident: 192 VGPRs of split-f16 halves, 1 wave per SIMD), B = the exchanged operand block of the 16 batch
// rows, so an accumulator lane holds 4 CONSECUTIVE units of one batch row: every global access of the gate math is 16 bytes.
// The operand block is fetched ONCE per workgroup (each wave polls a quarter of it), split into f16 halves and shared
// through LDS; positions along K are permuted (the same permutation on both operands) so that the polling loads are 64
// contiguous bytes per four lanes and the LDS stores 16 bytes.
constexpr int GRU_RB = 16;                             // batch rows per workgroup
constexpr int GRU_UB = 64;                             // hidden units per workgroup
constexpr int GRU_MEMBERS = GH / GRU_UB;               // workgroups per group
constexpr int FROW = GH + 8;                           // LDS pitch of a staged row in halves: rows 4 banks apart
constexpr int BROW = 3 * GH + 8;
static_assert(GRU_MEMBERS == 4, "gru_group_of / gru_member_of are written for four members");

// LDS position -> index along K.  Forward (K = 256): the staging lane (row, c) of wave w loads k = 64 w + 16 q + 4 c + i
// (q = 0..3) and owns positions 64 w + 16 c + 4 q + i.  An aligned run of 8 positions is two runs of 4 consecutive k, 16 apart.
__device__ __forceinline__ int gru_fwd_k_of(int p0) { return 64 * (p0 >> 6) + 16 * ((p0 >> 2) & 3) + 4 * ((p0 >> 4) & 3); }
// Backward (K = 768): wave w loads c = 192 w + 16 q + 4 c4 + i (q = 0..11) and owns positions 192 w + 48 c4 + 4 q + i.
__device__ __forceinline__ int gru_bwd_k_of(int p0) {
    const int w = p0 / 192, rem = p0 % 192;
    return 192 * w + 16 * ((rem % 48) >> 2) + 4 * (rem / 48);
}

// The saved gates (r, z, n, gh_n) travel from the forward to the backward recurrence only, and both kernels cut the work the
// same way, so they are stored TILE-major: [direction][t][row block][16-unit block][gate][lane][4] -- a wave's store of one
// gate is 1 KB contiguous (16 rows 4 KB apart cost 0.025 ms per store instruction and sequence at B = 256, contiguous 0.009:
// every poll waits for the wave's earlier stores, vmcnt counts in order).  Rows past B in the last block are never written.
__host__ __device__ __forceinline__ long gru_saves_tile(int d, int t, int T, int nrb, int rb, int ub) {
    return ((((long)d * T + t) * nrb + rb) * (GH / 16) + ub) * (4 * 16 * 16);
}

__device__ __forceinline__ void st_exchange4(float* ptr, const f4r& v, bool one_xcd) {
    if (one_xcd) {
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(ptr), "v"(v) : "memory");
    } else {
        st_coherent(ptr, make_float2(v[0], v[1]));
        st_coherent(ptr + 2, make_float2(v[2], v[3]));
    }
}
__device__ __forceinline__ floatx4 gru_mfma16(const half8& a, const half8& b, const floatx4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

struct GruSeqFwdP {
    const float* gi;           // [B][T][6H]: input projections incl. b_ih, forward gates then reverse gates
    const float* w[2];         // W_hh [3H][H]
    const float* bhh[2];       // [3H]
    float* hs;                 // [2][T][B][H]
    float* saves;              // r, z, n, gh_n in the tile layout of gru_saves_tile
    float* out;                // [B][T][2H]
    int* flags;                // workspace (zeroed by the launcher): the error word at [GRU_MAX_GROUPS]
    int B, T, ngroups;
    long spin_limit;
    int agent_scope;           // test hook: exchange with agent-scope stores although the group shares an XCD
};

__global__ __launch_bounds__(256) void gru_seq_fwd_kernel(GruSeqFwdP p) {
    __shared__ __attribute__((aligned(16))) _Float16 hb[2][2][GRU_RB * FROW];      // [step parity][hi, lo][row][position]: 33 KB
    __shared__ float wsc[4][3][16];
    __shared__ int xcd_word;
    const int group = gru_group_of(blockIdx.x), mb = gru_member_of(blockIdx.x);
    if (group >= p.ngroups) return;                    // filler workgroups of the XCD-aligned grid (whole workgroup, before any barrier)
    const int d = group & 1, r0 = (group >> 1) * GRU_RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q4 = lane >> 4;
    const int j0 = mb * GRU_UB + wave * 16;            // the wave's 16 hidden units
    const int B = p.B, T = p.T;
    const long bh = (long)B * GH;
    int* const err = p.flags + GRU_MAX_GROUPS;
    const long poll_limit = p.spin_limit > 8 ? (p.spin_limit >> 3) : 1;
    const bool one_xcd = gru_group_on_one_xcd(p.flags, group, poll_limit, &xcd_word) && !p.agent_scope;

    // A operand, resident for the whole sequence: row = unit j0 + c16 of gate g, positions 32 ks + 8 q4 .. + 8, as split-f16
    // halves scaled by a power of two of the ROW's own amax (the four lanes that share the row agree on it)
    half8 whi[3][8], wlo[3][8];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float* wr = p.w[d] + (long)(g * GH + j0 + c16) * GH;
        f4r b[8][2];
        float am = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int k0 = gru_fwd_k_of(32 * ks + 8 * q4);
            b[ks][0] = *reinterpret_cast<const f4r*>(wr + k0);
            b[ks][1] = *reinterpret_cast<const f4r*>(wr + k0 + 16);
            am = fmaxf(am, fmaxf(gru_amax4(b[ks][0]), gru_amax4(b[ks][1])));
        }
        am = fmaxf(am, __shfl_xor(am, 16, 64));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        const float sw = sed_sf_scale_of(am);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) gru_split8(b[ks][0], b[ks][1], sw, whi[g][ks], wlo[g][ks]);
        if (q4 == 0) wsc[wave][g][c16] = (1.0f / sw) * (1.0f / 8192.0f);      // h_prev is scaled by 2^13 (|h| <= 1)
    }
    __syncthreads();
    // output role: batch row r0 + c16, units j .. j + 3 (accumulator register i = unit 4 q4 + i of the wave's 16)
    float winv[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) winv[g][i] = wsc[wave][g][4 * q4 + i];
    const int row = r0 + c16;
    const bool live = row < B;
    const int rowc = live ? row : B - 1;
    const int j = j0 + 4 * q4;
    f4r bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = *reinterpret_cast<const f4r*>(p.bhh[d] + g * GH + j);
    // staging role: row r0 + (lane >> 2) of the operand block, k = 64 wave + 16 q + 4 (lane & 3) + i
    const int srow = lane >> 2, sc = lane & 3;
    const long soff = (long)min(r0 + srow, B - 1) * GH + 64 * wave + 4 * sc;
    const int spos = srow * FROW + 64 * wave + 16 * sc;
    const int fpos = c16 * FROW + 8 * q4;

    f4r hlast = {0.f, 0.f, 0.f, 0.f};
    bool dead = false;
    // Software pipeline around the wait: a poll ends with s_waitcnt vmcnt(0), which also waits for every OTHER memory operation
    // of the wave -- so the loads a step needs that do not depend on its predecessor (gi) are issued one step ahead, and the
    // stores nobody waits for (out, saved gates) one step late, both right behind the barrier where the MFMAs hide them
    // (stores in front of the poll cost 0.84 us per step, loads in front of it their full HBM latency).
    f4r pre[3], keep[5];
    int keep_t = -1;
    {
        const float* gir = p.gi + ((long)rowc * T + (d ? T - 1 : 0)) * 6 * GH + d * 3 * GH + j;
#pragma unroll
        for (int g = 0; g < 3; ++g) pre[g] = *reinterpret_cast<const f4r*>(gir + g * GH);
    }
    auto flush = [&]() {                               // out / saves of step keep_t
        if (live && keep_t >= 0) {
            *reinterpret_cast<f4r*>(p.out + ((long)row * T + keep_t) * 2 * GH + d * GH + j) = keep[4];
            float* s = p.saves + gru_saves_tile(d, keep_t, T, p.ngroups >> 1, group >> 1, mb * 4 + wave) + 4 * lane;
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<f4r*>(s + 256 * g) = keep[g];
        }
    };
    for (int k = 0; k < T; ++k) {
        const int t = d ? T - 1 - k : k;
        const f4r gr = pre[0], gz = pre[1], gn = pre[2];
        f4r gh[3] = {bias[0], bias[1], bias[2]};
        if (k > 0) {
            const float* h_prev = p.hs + ((long)d * T + (d ? t + 1 : t - 1)) * bh;
            f4r a[4];
            gru_poll<4, 16>(a, h_prev + soff, err, poll_limit, dead, false);
            half8 h0, l0, h1, l1;
            gru_split8(a[0], a[1], 8192.0f, h0, l0);
            gru_split8(a[2], a[3], 8192.0f, h1, l1);
            _Float16* const sh = &hb[k & 1][0][spos];
            _Float16* const sl = &hb[k & 1][1][spos];
            *reinterpret_cast<half8*>(sh) = h0;
            *reinterpret_cast<half8*>(sh + 8) = h1;
            *reinterpret_cast<half8*>(sl) = l0;
            *reinterpret_cast<half8*>(sl + 8) = l1;
            __syncthreads();                           // the only barrier of a step: the block of step k + 2 goes to this buffer
                                                       // after every wave has passed the barrier of step k + 1, i.e. has read it
        }
        if (k + 1 < T) {                               // next step's input projections
            const float* gir = p.gi + ((long)rowc * T + (d ? t - 1 : t + 1)) * 6 * GH + d * 3 * GH + j;
#pragma unroll
            for (int g = 0; g < 3; ++g) pre[g] = *reinterpret_cast<const f4r*>(gir + g * GH);
        }
        flush();                                       // previous step's out / saves
        if (k > 0) {
            // nine independent accumulator chains (three exact partial products x three gates): no MFMA waits for its predecessor
            floatx4 am[3], ac[3], ad[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) am[g] = ac[g] = ad[g] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const half8 ahi = *reinterpret_cast<const half8*>(&hb[k & 1][0][fpos + 32 * ks]);
                const half8 alo = *reinterpret_cast<const half8*>(&hb[k & 1][1][fpos + 32 * ks]);
#pragma unroll
                for (int g = 0; g < 3; ++g) ac[g] = gru_mfma16(wlo[g][ks], ahi, ac[g]);
#pragma unroll
                for (int g = 0; g < 3; ++g) ad[g] = gru_mfma16(whi[g][ks], alo, ad[g]);
#pragma unroll
                for (int g = 0; g < 3; ++g) am[g] = gru_mfma16(whi[g][ks], ahi, am[g]);
            }
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) gh[g][i] += (am[g][i] + (ac[g][i] + ad[g][i])) * winv[g][i];
        }
        f4r rr, zz, nn, hh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rr[i] = gru_sigmoid_hw(gr[i] + gh[0][i]);
            zz[i] = gru_sigmoid_hw(gz[i] + gh[1][i]);
            nn[i] = gru_tanh_hw(gn[i] + rr[i] * gh[2][i]);
            hh[i] = (1.0f - zz[i]) * nn[i] + zz[i] * hlast[i];     // this lane wrote h_{t-1}[row][j .. j+3] itself
        }
        if (live) st_exchange4(p.hs + ((long)d * T + t) * bh + (long)row * GH + j, hh, one_xcd);      // the hand-over: replaces the sentinel
        hlast = hh;
        keep[0] = rr; keep[1] = zz; keep[2] = nn; keep[3] = gh[2]; keep[4] = hh;
        keep_t = t;
    }
    flush();
}

struct GruSeqBwdP {
    const float* g_out;        // [B][T][2H]
    const float* wt[2];        // W_hh^T [H][3H]
    const float* hs;           // [2][T][B][H]
    const float* saves;        // gru_saves_tile layout
    float* dgi;                // [B][T][6H]
    float* dgh;                // [2][T][B][3H]
    float* dbias;              // nullable: [2 directions][row blocks][4: dr, dz, dn, dn*r][H] sums over (t, the block's rows)
    float* dgi_amax;           // nullable: amax slots of |dgi| (the operand scale of the split-f16 input-gradient GEMM)
    int* flags;
    int B, T, ngroups;
    long spin_limit;
    int agent_scope;           // test hook: exchange with agent-scope stores although the group shares an XCD
};

__global__ __launch_bounds__(256) void gru_seq_bwd_kernel(GruSeqBwdP p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 gb_dyn[];             // [step parity][hi, lo][row][position]: 2 x 49 KB
    constexpr int GPLANE = GRU_RB * BROW;
    __shared__ float qinv[2][4];                       // [step parity][quarter]: 1 / operand scale of the quarter a wave staged
    __shared__ float wsc[4][16];
    __shared__ int xcd_word;
    const int group = gru_group_of(blockIdx.x), mb = gru_member_of(blockIdx.x);
    if (group >= p.ngroups) return;
    const int d = group & 1, r0 = (group >> 1) * GRU_RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q4 = lane >> 4;
    const int j0 = mb * GRU_UB + wave * 16;
    const int B = p.B, T = p.T;
    const long bh = (long)B * GH;
    int* const err = p.flags + GRU_MAX_GROUPS;
    const long poll_limit = p.spin_limit > 8 ? (p.spin_limit >> 3) : 1;
    const bool one_xcd = gru_group_on_one_xcd(p.flags, group, poll_limit, &xcd_word) && !p.agent_scope;

    // dh_gemm[b][j] = sum_c dgh_later[b][c] * W_hh[c][j], c over 3H = 768.  A operand: row j0 + c16 of W_hh^T, resident
    half8 whi[24], wlo[24];
    {
        const float* wr = p.wt[d] + (long)(j0 + c16) * 3 * GH;
        float am = 0.f;
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) {
            const int k0 = gru_bwd_k_of(32 * ks + 8 * q4);
            am = fmaxf(am, fmaxf(gru_amax4(*reinterpret_cast<const f4r*>(wr + k0)), gru_amax4(*reinterpret_cast<const f4r*>(wr + k0 + 16))));
        }
        am = fmaxf(am, __shfl_xor(am, 16, 64));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        const float sw = sed_sf_scale_of(am);
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) {
            const int k0 = gru_bwd_k_of(32 * ks + 8 * q4);
            gru_split8(*reinterpret_cast<const f4r*>(wr + k0), *reinterpret_cast<const f4r*>(wr + k0 + 16), sw, whi[ks], wlo[ks]);
        }
        if (q4 == 0) wsc[wave][c16] = 1.0f / sw;
    }
    __syncthreads();
    float winv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) winv[i] = wsc[wave][4 * q4 + i];
    const int row = r0 + c16;
    const bool live = row < B;
    const int rowc = live ? row : B - 1;
    const int j = j0 + 4 * q4;
    const int srow = lane >> 2, sc = lane & 3;
    const long soff = (long)min(r0 + srow, B - 1) * 3 * GH + 192 * wave + 4 * sc;
    const int spos = srow * BROW + 192 * wave + 48 * sc;
    const int fpos = c16 * BROW + 8 * q4;

    f4r dhz = {0.f, 0.f, 0.f, 0.f};                    // dh * z of the step processed before (the later time step)
    // bias gradients (db_ih = column sums of dgi, db_hh = of dgh) accumulate here over the time steps
    f4r sb_r = dhz, sb_z = dhz, sb_n = dhz, sb_nr = dhz;
    float gmax = 0.f;                                  // max |dgi| this thread wrote
    bool dead = false;

    // software pipeline around the wait, as in the forward kernel: the loads of a step that do not depend on the later step
    // (output gradient, saved gates, previous hidden state) are issued one step ahead, the dgi stores one step late
    f4r pre[6], keep[3];
    int keep_t = -1;
    auto prefetch = [&](int kk) {                      // step kk of the processing order
        const int tt = d ? T - 1 - kk : kk;
        pre[0] = *reinterpret_cast<const f4r*>(p.g_out + ((long)rowc * T + tt) * 2 * GH + d * GH + j);
        const float* s = p.saves + gru_saves_tile(d, tt, T, p.ngroups >> 1, group >> 1, mb * 4 + wave) + 4 * lane;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[1 + g] = *reinterpret_cast<const f4r*>(s + 256 * g);
        pre[5] = f4r{0.f, 0.f, 0.f, 0.f};
        if (kk > 0) pre[5] = *reinterpret_cast<const f4r*>(p.hs + ((long)d * T + (d ? tt + 1 : tt - 1)) * bh + (long)rowc * GH + j);
    };
    auto flush = [&]() {
        if (live && keep_t >= 0) {
            float* gi_o = p.dgi + ((long)row * T + keep_t) * 6 * GH + d * 3 * GH + j;
#pragma unroll
            for (int g = 0; g < 3; ++g) *reinterpret_cast<f4r*>(gi_o + g * GH) = keep[g];
        }
    };
    prefetch(T - 1);
    for (int k = T - 1, done = 0; k >= 0; --k, ++done) {
        const int t = d ? T - 1 - k : k;
        f4r dh = pre[0];
        const f4r rr = pre[1], zz = pre[2], nn = pre[3], ghn = pre[4], hp = pre[5];
        dh += dhz;
        if (done > 0) {
            const float* dgh_later = p.dgh + ((long)d * T + (d ? t - 1 : t + 1)) * 3 * bh;
            f4r a[12];
            gru_poll<12, 16>(a, dgh_later + soff, err, poll_limit, dead, !one_xcd);
            // gradients have no fixed range: one power-of-two scale per staged quarter (16 rows x 192) and step from its amax
            float am = 0.f;
#pragma unroll
            for (int q = 0; q < 12; ++q) am = fmaxf(am, gru_amax4(a[q]));
            const float sa = sed_sf_scale_of(wave_max(am));
            _Float16* const gb = gb_dyn + (done & 1) * 2 * GPLANE;      // the block of step n + 2 goes here after every wave has
                                                                       // passed the barrier of step n + 1, i.e. has read this one
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                half8 hi, lo;
                gru_split8(a[2 * m], a[2 * m + 1], sa, hi, lo);
                *reinterpret_cast<half8*>(&gb[spos + 8 * m]) = hi;
                *reinterpret_cast<half8*>(&gb[GPLANE + spos + 8 * m]) = lo;
            }
            if (lane == 0) qinv[done & 1][wave] = 1.0f / sa;
            __syncthreads();
        }
        if (k > 0) prefetch(k - 1);
        flush();
        if (done > 0) {
            const _Float16* const gb = gb_dyn + (done & 1) * 2 * GPLANE;
            f4r g4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                floatx4 am4 = {0.f, 0.f, 0.f, 0.f}, ac4 = am4, ad4 = am4;
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const int ks = 6 * qt + m;
                    const half8 bhi = *reinterpret_cast<const half8*>(&gb[fpos + 32 * ks]);
                    const half8 blo = *reinterpret_cast<const half8*>(&gb[GPLANE + fpos + 32 * ks]);
                    ac4 = gru_mfma16(wlo[ks], bhi, ac4);
                    ad4 = gru_mfma16(whi[ks], blo, ad4);
                    am4 = gru_mfma16(whi[ks], bhi, am4);
                }
                const float u = qinv[done & 1][qt];
#pragma unroll
                for (int i = 0; i < 4; ++i) g4[i] += (am4[i] + (ac4[i] + ad4[i])) * u;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) dh[i] += g4[i] * winv[i];
        }
        f4r dr_pre, dz_pre, dn_pre, dn_r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dn = dh[i] * (1.0f - zz[i]);
            const float dz = dh[i] * (hp[i] - nn[i]);
            dn_pre[i] = dn * (1.0f - nn[i] * nn[i]);
            const float dr = dn_pre[i] * ghn[i];
            dr_pre[i] = dr * rr[i] * (1.0f - rr[i]);
            dz_pre[i] = dz * zz[i] * (1.0f - zz[i]);
            dn_r[i] = dn_pre[i] * rr[i];
            dhz[i] = dh[i] * zz[i];
        }
        if (live) {
            float* gh_o = p.dgh + ((long)d * T + t) * 3 * bh + (long)row * 3 * GH + j;
            st_exchange4(gh_o, dr_pre, one_xcd);       // the hand-over
            st_exchange4(gh_o + GH, dz_pre, one_xcd);
            st_exchange4(gh_o + 2 * GH, dn_r, one_xcd);
            sb_r += dr_pre; sb_z += dz_pre; sb_n += dn_pre; sb_nr += dn_r;
#pragma unroll
            for (int i = 0; i < 4; ++i) gmax = fmaxf(gmax, fmaxf(fabsf(dr_pre[i]), fmaxf(fabsf(dz_pre[i]), fabsf(dn_pre[i]))));
        }
        keep[0] = dr_pre; keep[1] = dz_pre; keep[2] = dn_pre;
        keep_t = t;
    }
    flush();
    if (p.dgi_amax) amax_publish_block(p.dgi_amax, gmax);
    if (p.dbias) {                                     // sum over the 16 batch rows of the block: the lanes with equal q4
        f4r* const sums[4] = {&sb_r, &sb_z, &sb_n, &sb_nr};
#pragma unroll
        for (int kind = 0; kind < 4; ++kind) {
            f4r v = *sums[kind];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += __shfl_xor(v[i], o, 64);
            if (c16 == 0) *reinterpret_cast<f4r*>(p.dbias + (((long)d * (p.ngroups >> 1) + (group >> 1)) * 4 + kind) * GH + j) = v;
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

// the backward kernel double-buffers its 49 KB operand block: dynamic LDS above the 64 KB static limit, raised per device
constexpr int GRU_BWD_LDS = 2 * 2 * GRU_RB * BROW * (int)sizeof(_Float16);
bool gru_bwd_lds_raised() {
    static int raised_dev = -1;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev != raised_dev) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gru_seq_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                GRU_BWD_LDS) != hipSuccess) return false;
        raised_dev = dev;
    }
    return true;
}

long g_spin_limit = 1L << 23;          // ~1 s of polling
int g_agent_scope = 0;

// all workgroups of the persistent kernels fit on the current device at once?  (cached per device)
bool gru_device_fits(int grid) {
    static int cached_dev = -1, cached_cap = 0;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return false;
    if (dev != cached_dev) {
        int cus = 0, per_f = 0, per_b = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_f, gru_seq_fwd_kernel, 256, 0) != hipSuccess) return false;
        if (!gru_bwd_lds_raised() || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_b, gru_seq_bwd_kernel, 256, GRU_BWD_LDS) != hipSuccess) return false;
        // one workgroup per CU is what the kernels are laid out for; the occupancy API may over-report by one block per CU
        // (MI355X_MICROARCH.md), so only its ">= 1" answer is used
        cached_cap = (per_f >= 1 && per_b >= 1) ? cus : 0;
        cached_dev = dev;
    }
    return grid <= cached_cap;
}

}  // namespace

SED_API int sed_gru_seq_supported(int B, int Hd) {
    if (!(Hd == GH && B > 0 && gru_grid(2 * sed_cdiv(B, GRU_RB)) <= 256)) return 0;
    return gru_device_fits(gru_grid(2 * sed_cdiv(B, GRU_RB))) ? 1 : 0;
}
SED_API long sed_gru_seq_ws_floats(void) { return GRU_FLAG_INTS; }
SED_API int sed_gru_seq_row_block(void) { return GRU_RB; }
SED_API long sed_gru_seq_saves_floats(int B, int T) { return B > 0 && T > 0 ? 2L * T * sed_cdiv(B, GRU_RB) * GRU_RB * 4 * GH : 0; }
SED_API int sed_gru_set_spin_limit(long spins) {
    g_spin_limit = spins > 0 ? spins : (1L << 23);
    return 0;
}
SED_API int sed_gru_force_agent_scope(int on) {
    g_agent_scope = on ? 1 : 0;
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
    const int ngroups = 2 * sed_cdiv(B, GRU_RB);
    if (B <= 0 || T <= 0 || Hd != GH || gru_grid(ngroups) > 256) return SED_EINVAL;
    hipError_t e = hipMemsetAsync(ws, 0, GRU_FLAG_INTS * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    // the hand-over buffer starts out as sentinels; every word of it is replaced by the kernel
    e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(hs), (int)GRU_SENTINEL, (size_t)2 * T * B * GH, stream);
    if (e != hipSuccess) return (int)e;
    GruSeqFwdP p{gi, {w_hh_f, w_hh_b}, {b_hh_f, b_hh_b}, hs, saves, out, reinterpret_cast<int*>(ws), B, T, ngroups, g_spin_limit, g_agent_scope};
    hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3(gru_grid(ngroups)), dim3(256), 0, stream, p);
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
    const int ngroups = 2 * sed_cdiv(B, GRU_RB);
    if (B <= 0 || T <= 0 || Hd != GH || gru_grid(ngroups) > 256) return SED_EINVAL;
    hipError_t e = hipMemsetAsync(ws, 0, GRU_FLAG_INTS * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dgh), (int)GRU_SENTINEL, (size_t)2 * T * B * 3 * GH, stream);
    if (e != hipSuccess) return (int)e;
    if (dgi_amax) {
        e = sed_amax_clear(dgi_amax, stream);
        if (e != hipSuccess) return (int)e;
    }
    GruSeqBwdP p{g_out, {wt_f, wt_b}, hs, saves, dgi, dgh, dbias_parts, dgi_amax, reinterpret_cast<int*>(ws), B, T, ngroups, g_spin_limit, g_agent_scope};
    if (!gru_bwd_lds_raised()) return SED_EINVAL;
    hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3(gru_grid(ngroups)), dim3(256), GRU_BWD_LDS, stream, p);
    SED_LAUNCH_CHECK();
    hipLaunchKernelGGL(gru_seq_check_kernel, dim3(256), dim3(256), 0, stream, reinterpret_cast<const int*>(ws), err_host, 2, dgi,
                       (long)B * T * 6 * GH);
    SED_LAUNCH_CHECK();
    return 0;
}

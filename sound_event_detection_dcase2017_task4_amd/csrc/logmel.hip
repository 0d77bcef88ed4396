// K1  log-mel front-end: reflect-pad STFT (Hann, n_fft 1024, hop 320) -> power -> 64-band mel -> 10*log10.
//
// Replaces the reference's `Spectrogram` + `LogmelFilterBank` calls (reference pytorch/models.py:284-285;
// torchlibrosa 0.0.4 semantics) WITHOUT materialising the (B2,1,T,513) power spectrogram.
// HBM traffic = waveform once + (T,64) out: 1,536,256 B per 10 s clip (fp32 in) -- the roofline denominator of
// SURVEY.md §8d.  The kernel is bound by the vector ALU and the LDS, not by HBM (22 flop/B), so it is written for
// instruction count and LDS bytes:
//
//  * two real frames = one 1024-point COMPLEX FFT (z = a + i b) on (re, im) register pairs: every butterfly add and every half
//    of a twiddle product is ONE packed-fp32 instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32; the +-i rotations and the
//    complex products use the op_sel / neg operand modifiers instead of extra instructions);
//  * factorisation 1024 = 32 x 32 (n = 32 n1 + n2, k = k1 + 32 k2): a half-wave (32 lanes) owns one FFT, 32 complex values per
//    lane; radix-32 in registers over n1 (lane = n2), twiddle W1024^(n2 k1) from 31 per-lane register pairs, ONE transposition
//    through padded LDS, radix-32 over n2 (lane = row k1, register = k2);
//  * rows k1 and 32 - k1 sit on ADJACENT lanes, so the mirror bin Z[1024 - k] of the real-spectrum unpack is the neighbour's
//    register 31 - k2 (one DPP move, no LDS); the two self-mirrored rows 0 and 16 are patched by selects; |.|^2 of both frames
//    is stored as ONE (Pa, Pb) pair per bin (the 1/4 of the unpack is folded into the mel weights);
//  * mel: the 866 non-zero taps of the 513x64 filter bank are cut into <= 12-bin tasks (105 of them), one ds_read_b64 + one
//    v_pk_fma_f32 per tap for both frames; a lane owns the same two tasks for every frame, its 24 weights live in registers, and
//    the host places the task windows so that the reads are bank-conflict free (ops.mel_task_tables);
//  * a wave walks 32 consecutive frames (8 iterations of 2 FFTs), so the ~150 table loads that fill the constant registers are
//    paid once per 16 FFTs;
//  * 10 log10(x) = 3.0103 log2(x) on the hardware v_log_f32 (1 ulp: < 1e-5 dB), the clamp mapped to exactly -100 dB.
//
// Cost model (tools/valu_ubench.hip on this part, every SIMD of the CU busy: v_add/v_fma_f32 3.3, v_pk_add/v_pk_fma_f32 5.2,
// ds_read_b64 10, ds_write_b64 26 cycles per wave-instruction per SIMD = 80 B/clk/CU of LDS stores).  Per frame pair: ~350
// packed (1820 cycles) + ~150 plain VALU (500) + 13 KB of LDS stores (670) + LDS loads (430) + VMEM / SALU (400) = 3800; the
// kernel measures 4450 (0.48 ms per 512 x 10 s waveforms).  The 16 x 16 x 4 factorisation of the first packed version (two
// transpositions + a natural-order spectrum round trip = 29 KB of LDS stores per pair, conflicted mel reads) measured the
// 5500 cycles its instruction count predicts (0.59 ms); neither two independent frame pairs in flight per wave nor prefetching
// the next pair's samples changed that, i.e. the kernel is throughput-bound, not latency-bound.  2 waves/SIMD (248 VGPRs):
// 3 waves/SIMD spills (0.83 ms), streaming twiddles / mel weights from L1-resident tables to reach 3-4 waves runs 1.3-1.7 ms.
#include "common.h"
#include "sed_hip.h"
#include <math.h>
SED_OBJECT_FLAGS(logmel)

namespace {

constexpr int NFFT = 1024;
constexpr int HOP = 320;
constexpr int MELW_MAX = 2048;
// frames per wave: 32 (8 iterations x 2 FFT pairs, 128 frames per workgroup: the ~150 table loads of a wave are amortised over 16 FFTs),
// or 8 for launches that would otherwise leave most CUs empty (4 clips per GPU: 8 waveforms x 8 workgroups of 128 frames = 64
// workgroups on 256 CUs, 41 us -- latency, not work)
constexpr int FPW_FULL = 32, FPW_SMALL = 8;
constexpr int TASK_TAPS = 12;
constexpr int MAX_TASKS = 128;

typedef float f2 __attribute__((ext_vector_type(2)));

// ---- complex arithmetic on (re, im) register pairs: one packed instruction each ----------------------------------
// An operand whose HIGH half feeds the low lane is always SRC0 here.  On this part a packed-fp32 instruction with
// op_sel[src1] = 1 and op_sel[src0] = 0 (src1 halves swapped, or src1.hi broadcast) returns wrong values now and then while a
// wave of ANOTHER kernel on the same CU executes v_mfma_f32_32x32x16_f16 -- never alone, never beside fp32 MFMAs, never with
// the high-half read on src0 (tools/pk_f32_beside_mfma_probe.hip, profiles/r03/pk_f32_beside_f16_mfma.txt; rocFFT's kernels
// -- torch.stft -- are hit the same way).  The training step never runs this kernel beside an MFMA kernel, but two ranks
// sharing one GPU (tests) or a prefetch stream would.
// a + (-i) b = (a.x + b.y, a.y - b.x) = (b.y + a.x, -b.x + a.y)
__device__ __forceinline__ f2 c_add_mi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(b), "v"(a));
    return r;
}
// a + (+i) b = (a.x - b.y, a.y + b.x) = (-b.y + a.x, b.x + a.y)
__device__ __forceinline__ f2 c_add_pi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(r) : "v"(b), "v"(a));
    return r;
}
// a * w, w = (wr, wi) in a VGPR pair: t = a * wr;  r = (t.x - a.y wi, t.y + a.x wi)
__device__ __forceinline__ f2 c_mul(f2 a, f2 w) {
    f2 t, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=&v"(t), "=v"(r) : "v"(a), "v"(w));
    return r;
}
// the same with a wave-uniform twiddle in an SGPR pair
__device__ __forceinline__ f2 c_mul_s(f2 a, f2 w) {
    f2 t, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=&v"(t), "=v"(r) : "v"(a), "s"(w));
    return r;
}
// a + conj(b) and a - conj(b)
__device__ __forceinline__ f2 c_add_conj(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 c_sub_conj(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// acc + p * w.x (HI = false) or acc + p * w.y (HI = true): the weight pair feeds two consecutive filter taps
template <bool HI>
__device__ __forceinline__ f2 pk_fma_tap(f2 p, f2 w, f2 acc) {
    f2 r;
    if (HI) asm("v_pk_fma_f32 %0, %2, %1, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(p), "v"(w), "v"(acc));   // w.y broadcast as SRC0 (see above)
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(p), "v"(w), "v"(acc));
    return r;
}

// 4-point DFT, in place; a2 may carry a pending factor -i (MI2) that is absorbed into the first butterfly.
template <bool MI2>
__device__ __forceinline__ void fft4(f2& a0, f2& a1, f2& a2, f2& a3) {
    const f2 t0 = MI2 ? c_add_mi(a0, a2) : a0 + a2;
    const f2 t1 = MI2 ? c_add_pi(a0, a2) : a0 - a2;
    const f2 t2 = a1 + a3;
    const f2 d = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = c_add_mi(t1, d);                        // t1 + (a1 - a3)(-i)
    a3 = c_add_pi(t1, d);
}

// 8-point DFT over e[0..7] (stride-agnostic references), natural order in and out: two 4-point DFTs + radix-2 (28 packed ops)
__device__ __forceinline__ void fft8(f2& e0, f2& e1, f2& e2, f2& e3, f2& e4, f2& e5, f2& e6, f2& e7) {
    constexpr float R = 0.70710678118654752f;
    fft4<false>(e0, e2, e4, e6);                 // E[0..3] in e0, e2, e4, e6
    fft4<false>(e1, e3, e5, e7);                 // O[0..3] in e1, e3, e5, e7
    const f2 rp = {R, R}, rn = {-R, -R};
    f2 t1 = c_add_mi(e3, e3);                    // O1 (1 - i)
    asm("v_pk_mul_f32 %0, %0, %1" : "+v"(t1) : "s"(rp));
    f2 t3 = c_add_pi(e7, e7);                    // O3 (1 + i)
    asm("v_pk_mul_f32 %0, %0, %1" : "+v"(t3) : "s"(rn));     // O3 W8^3 = -R (1 + i) O3
    const f2 E0 = e0, E1 = e2, E2 = e4, E3 = e6, O0 = e1, O2 = e5;
    e0 = E0 + O0;             e4 = E0 - O0;
    e1 = E1 + t1;             e5 = E1 - t1;
    e2 = c_add_mi(E2, O2);    e6 = c_add_pi(E2, O2);         // W8^2 = -i
    e3 = E3 + t3;             e7 = E3 - t3;
}

// In-register 32-point forward DFT, natural order in and out: n = 4 na + nb, k = ka + 8 kb: four 8-point DFTs over na,
// twiddle W32^(nb ka), eight 4-point DFTs over nb (218 packed instructions).
__device__ __forceinline__ void fft32(f2 (&x)[32]) {
    static constexpr float W32[22][2] = {
        {1.f, -0.f},
        {0.98078525066375732f, -0.19509032368659973f}, {0.92387950420379639f, -0.38268342614173889f},
        {0.83146959543228149f, -0.55557024478912354f}, {0.70710676908493042f, -0.70710676908493042f},
        {0.55557024478912354f, -0.83146959543228149f}, {0.38268342614173889f, -0.92387950420379639f},
        {0.19509032368659973f, -0.98078525066375732f}, {0.f, -1.f},
        {-0.19509032368659973f, -0.98078525066375732f}, {-0.38268342614173889f, -0.92387950420379639f},
        {-0.55557024478912354f, -0.83146959543228149f}, {-0.70710676908493042f, -0.70710676908493042f},
        {-0.83146959543228149f, -0.55557024478912354f}, {-0.92387950420379639f, -0.38268342614173889f},
        {-0.98078525066375732f, -0.19509032368659973f}, {-1.f, -0.f},
        {-0.98078525066375732f, 0.19509032368659973f}, {-0.92387950420379639f, 0.38268342614173889f},
        {-0.83146959543228149f, 0.55557024478912354f}, {-0.70710676908493042f, 0.70710676908493042f},
        {-0.55557024478912354f, 0.83146959543228149f}};
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
        fft8(x[nb], x[4 + nb], x[8 + nb], x[12 + nb], x[16 + nb], x[20 + nb], x[24 + nb], x[28 + nb]);
    // index 4 ka + nb now holds the ka-th output of column nb
#pragma unroll
    for (int ka = 1; ka < 8; ++ka)
#pragma unroll
        for (int nb = 1; nb < 4; ++nb) {
            const f2 w = {W32[nb * ka][0], W32[nb * ka][1]};
            x[4 * ka + nb] = c_mul_s(x[4 * ka + nb], w);
        }
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) fft4<false>(x[4 * ka], x[4 * ka + 1], x[4 * ka + 2], x[4 * ka + 3]);
    // index 4 ka + kb holds X[ka + 8 kb]: to natural order (register renaming)
    f2 y[32];
#pragma unroll
    for (int ka = 0; ka < 8; ++ka)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) y[ka + 8 * kb] = x[4 * ka + kb];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = y[i];
}

template <typename T> __device__ __forceinline__ float load_sample(const T* p, long i);
template <> __device__ __forceinline__ float load_sample<float>(const float* p, long i) { return p[i]; }
// utils/utilities.py:66-67  int16_to_float32: x / 32767.  (the 1/32767 is folded into the window registers)
template <> __device__ __forceinline__ float load_sample<short>(const short* p, long i) { return (float)p[i]; }

// sum of the (<= 4) task partials of this lane's band
__device__ __forceinline__ void band_sum(const f2* mp, int4 bd, int max_band_tasks, f2& m) {
    const int slot[4] = {bd.x, bd.y, bd.z, bd.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < max_band_tasks) {
            const f2 v = mp[slot[j] < 0 ? 0 : slot[j]];
            if (slot[j] >= 0) m += v;
        }
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr int T32 = 33;                          // transposition row stride in float2 (32 + 1 pad)
constexpr int WBUF32 = 32 * T32;                 // 1056 float2 per half-wave: transposition / powers / mel partials, aliased

__device__ __forceinline__ float dpp_swap1(float v) {      // value of lane ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

template <typename T, int FPW = FPW_FULL>
__global__ __launch_bounds__(256, 2) void logmel32_kernel(const T* __restrict__ wave, int L, int T_frames,
                                                          const float* __restrict__ window,      // [32 n1][32 n2] = natural order
                                                          const float2* __restrict__ tw1024t,    // [32 k1][32 n2] W1024^(n2*k1)
                                                          const int4* __restrict__ tasks, int ntasks,
                                                          const int4* __restrict__ bands, int max_band_tasks,
                                                          const float* __restrict__ mel_w, float amin, float floor_db,
                                                          float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) f2 lds[4 * 2 * WBUF32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int frame0 = blockIdx.x * (4 * FPW) + wv * FPW;
    if (frame0 >= T_frames) return;
    const T* x = wave + (long)b * L;
    f2* const tbw = lds + wv * 2 * WBUF32;             // two private buffers of this wave (one per half-wave / frame pair)
    f2* const tbh = tbw + h * WBUF32;
    // row of the transposed matrix this lane owns: lanes (2p, 2p+1) hold rows (p, 32 - p); lanes 0, 1 rows 0 and 16
    const int row = j == 0 ? 0 : (j == 1 ? 16 : ((j & 1) ? 32 - (j >> 1) : (j >> 1)));

    constexpr float in_scale = sizeof(T) == 2 ? (float)(1.0 / 32767.0) : 1.0f;
    float win[32];
#pragma unroll
    for (int n1 = 0; n1 < 32; ++n1) win[n1] = window[32 * n1 + j] * in_scale;
    f2 twa[32];                                        // W1024^(n2 * k1), n2 = j
#pragma unroll
    for (int k = 1; k < 32; ++k) { const float2 w = tw1024t[k * 32 + j]; twa[k] = f2{w.x, w.y}; }
    int tlo[2];
    f2 tw_[2][TASK_TAPS / 2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int t = lane + 64 * r;
        int4 tk = make_int4(0, 0, 0, 0);
        if (t < ntasks) tk = tasks[t];
        tlo[r] = tk.x;
#pragma unroll
        for (int i = 0; i < TASK_TAPS / 2; ++i)
            tw_[r][i] = f2{2 * i < tk.y ? 0.25f * mel_w[tk.z + 2 * i] : 0.f, 2 * i + 1 < tk.y ? 0.25f * mel_w[tk.z + 2 * i + 1] : 0.f};
    }
    const int4 bd = bands[lane];
    const bool is0 = j == 0, is1 = j == 1;

    for (int it = 0; it < FPW / 4; ++it) {
        const int ta0 = frame0 + 4 * it;               // this iteration: pairs (ta0, ta0+1) on lanes 0-31, (ta0+2, ta0+3) on 32-63
        if (ta0 >= T_frames) break;                    // wave-uniform
        const int ta = ta0 + 2 * h;
        const int base = ta * HOP - NFFT / 2;
        f2 z[32];
        const int base0 = ta0 * HOP - NFFT / 2;
        if (base0 >= 0 && base0 + 3 * HOP + NFFT <= L && ta0 + 3 < T_frames) {        // all four frames interior
            const T* xa = x + base + j;
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) z[n1] = f2{load_sample<T>(xa, 32 * n1), load_sample<T>(xa, HOP + 32 * n1)};
        } else {                                       // clip edges: F.pad(mode='reflect') indexing, frames past the end = 0
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) {
                int ia = base + 32 * n1 + j, ib = ia + HOP;
                ia = ia < 0 ? -ia : ia; ia = ia >= L ? 2 * (L - 1) - ia : ia;
                ib = ib < 0 ? -ib : ib; ib = ib >= L ? 2 * (L - 1) - ib : ib;
                const float va = (ta < T_frames && ia >= 0 && ia < L) ? load_sample<T>(x, ia) : 0.f;
                const float vb = (ta + 1 < T_frames && ib >= 0 && ib < L) ? load_sample<T>(x, ib) : 0.f;
                z[n1] = f2{va, vb};
            }
        }
#pragma unroll
        for (int n1 = 0; n1 < 32; ++n1) z[n1] = z[n1] * f2{win[n1], win[n1]};
        // ---- pass A: 32-point DFT over n1, twiddle W1024^(n2 k1)
        fft32(z);
#pragma unroll
        for (int k = 1; k < 32; ++k) z[k] = c_mul(z[k], twa[k]);
        wave_lds_fence();                              // previous iteration's readers of this buffer are done
#pragma unroll
        for (int k = 0; k < 32; ++k) tbh[k * T32 + j] = z[k];
        wave_lds_fence();
#pragma unroll
        for (int n2 = 0; n2 < 32; ++n2) z[n2] = tbh[row * T32 + n2];
        // ---- pass B: 32-point DFT over n2: z[k2] = Z[row + 32 k2]
        fft32(z);
        // ---- real-spectrum unpack in registers.  Mirror of bin row + 32 k2 is (32 - row) + 32 (31 - k2): the neighbour lane's
        // register 31 - k2.  Row 16 mirrors onto its own register 31 - k2, row 0 onto its own register (32 - k2) & 31.
        f2 pw[16];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const int r = 31 - k2;
            f2 q = f2{dpp_swap1(z[r].x), dpp_swap1(z[r].y)};
            const f2 own0 = z[(r + 1) & 31];           // row 0
            q.x = is1 ? z[r].x : (is0 ? own0.x : q.x);
            q.y = is1 ? z[r].y : (is0 ? own0.y : q.y);
            const f2 s = c_add_conj(z[k2], q), d = c_sub_conj(z[k2], q);
            const f2 s2 = s * s, d2 = d * d;
            pw[k2] = f2{s2.x + s2.y, d2.x + d2.y};
        }
        const f2 p512 = f2{4.f * z[16].x * z[16].x, 4.f * z[16].y * z[16].y};    // bin 512 = row 0, k2 = 16 (real and imaginary parts)
        wave_lds_fence();                              // the transposition reads are done: the buffer becomes the power pairs
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) tbh[row + 32 * k2] = pw[k2];
        if (is0) tbh[512] = p512;
        if (j < 16) tbh[513 + j] = f2{0.f, 0.f};       // read-ahead of the fixed 12-tap tasks stays finite
        wave_lds_fence();
        // ---- mel + log + store, one frame pair after the other on all 64 lanes
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f2* pb = tbw + u * WBUF32;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                f2 acc = {0.f, 0.f};
                const f2* pp = pb + tlo[r];
#pragma unroll
                for (int i = 0; i < TASK_TAPS / 2; ++i) {
                    acc = pk_fma_tap<false>(pp[2 * i], tw_[r][i], acc);
                    acc = pk_fma_tap<true>(pp[2 * i + 1], tw_[r][i], acc);
                }
                tbw[u * WBUF32 + 544 + lane + 64 * r] = acc;
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tu = ta0 + 2 * u;
            const f2* mp = tbw + u * WBUF32 + 544;
            f2 m = {0.f, 0.f};
            band_sum(mp, bd, max_band_tasks, m);
            float* o = out + ((long)b * T_frames + tu) * 64 + lane;
            if (tu < T_frames) o[0] = !(m.x <= amin) ? 3.0102999566398120f * __log2f(m.x) : floor_db;     // NaN takes the log branch: propagates like torch.clamp
            if (tu + 1 < T_frames) o[64] = !(m.y <= amin) ? 3.0102999566398120f * __log2f(m.y) : floor_db;
        }
    }
}

template <typename T>
int launch_logmel(const T* wave, int B2, int L, const float* window, const float* tw1024t,
                  const int* tasks, int ntasks, const int* bands, int max_band_tasks, const float* mel_w, int mel_nnz, float amin,
                  float* out, hipStream_t stream) {
    if (B2 <= 0 || L <= NFFT / 2 || (long)L + 2 * NFFT >= (1L << 31) || mel_nnz <= 0 || mel_nnz > MELW_MAX || ntasks <= 0 || ntasks > MAX_TASKS ||
        max_band_tasks <= 0 || max_band_tasks > 4)
        return SED_EINVAL;
    int T_frames = L / HOP + 1;
    const float floor_db = (float)(10.0 * log10((double)amin));
    if ((long)sed_cdiv(T_frames, 4 * FPW_FULL) * B2 < 256) {        // fewer workgroups than CUs: quarter the frames per workgroup
        dim3 grid(sed_cdiv(T_frames, 4 * FPW_SMALL), B2);
        hipLaunchKernelGGL((logmel32_kernel<T, FPW_SMALL>), grid, dim3(256), 0, stream, wave, L, T_frames, window,
                           reinterpret_cast<const float2*>(tw1024t), reinterpret_cast<const int4*>(tasks), ntasks,
                           reinterpret_cast<const int4*>(bands), max_band_tasks, mel_w, amin, floor_db, out);
    } else {
        dim3 grid(sed_cdiv(T_frames, 4 * FPW_FULL), B2);
        hipLaunchKernelGGL((logmel32_kernel<T, FPW_FULL>), grid, dim3(256), 0, stream, wave, L, T_frames, window,
                           reinterpret_cast<const float2*>(tw1024t), reinterpret_cast<const int4*>(tasks), ntasks,
                           reinterpret_cast<const int4*>(bands), max_band_tasks, mel_w, amin, floor_db, out);
    }
    SED_LAUNCH_CHECK();
    return 0;
}

}  // namespace

SED_API int sed_logmel_f32(const float* wave, int B2, int L, const float* window, const float* tw1024t,
                           const int* mel_tasks, int n_tasks, const int* mel_bands, int max_band_tasks,
                           const float* mel_w, int mel_nnz, float amin, float* out, hipStream_t stream) {
    return launch_logmel<float>(wave, B2, L, window, tw1024t, mel_tasks, n_tasks, mel_bands, max_band_tasks, mel_w, mel_nnz,
                                amin, out, stream);
}

SED_API int sed_logmel_i16(const short* wave, int B2, int L, const float* window, const float* tw1024t,
                           const int* mel_tasks, int n_tasks, const int* mel_bands, int max_band_tasks,
                           const float* mel_w, int mel_nnz, float amin, float* out, hipStream_t stream) {
    return launch_logmel<short>(wave, B2, L, window, tw1024t, mel_tasks, n_tasks, mel_bands, max_band_tasks, mel_w, mel_nnz,
                                amin, out, stream);
}

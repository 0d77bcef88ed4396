// K1  log-mel front-end: reflect-pad STFT (Hann, n_fft 1024, hop 320) -> power -> 64-band mel -> 10*log10.
//
// Replaces the reference's `Spectrogram` + `LogmelFilterBank` calls (reference pytorch/models.py:284-285;
// torchlibrosa 0.0.4 semantics) WITHOUT materialising the (B2,1,T,513) power
// spectrogram: each wave packs two real frames into one 1024-point complex FFT (16 x 4 x 16 factorisation: two in-register
// radix-16 passes, one LDS transpose, one 4-lane shuffle transpose), unpacks the two spectra, and lane m
// accumulates mel band m from the compact (non-zero only) filter table.  HBM traffic = waveform once +
// (T,64) out: 1,536,256 B per 10 s clip (fp32 in) -- the roofline denominator of SURVEY.md §8d.
#include "common.h"
#include "sed_hip.h"
#include <math.h>

namespace {

constexpr int NFFT = 1024;
constexpr int HOP = 320;
constexpr int TROW = 68;                         // transpose row stride in float2 (64 + 4 pad: conflict-free)
constexpr int NBINS = 513;
constexpr int PSTR = 516;
constexpr int MELW_MAX = 1024;

struct cpx { float re, im; };

__device__ __forceinline__ void fft4(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2,
                                     float& r3, float& i3) {
    float t0r = r0 + r2, t0i = i0 + i2, t1r = r0 - r2, t1i = i0 - i2;
    float t2r = r1 + r3, t2i = i1 + i3;
    float t3r = i1 - i3, t3i = -(r1 - r3);       // (a1 - a3) * (-i)
    r0 = t0r + t2r; i0 = t0i + t2i;
    r1 = t1r + t3r; i1 = t1i + t3i;
    r2 = t0r - t2r; i2 = t0i - t2i;
    r3 = t1r - t3r; i3 = t1i - t3i;
}

__device__ __forceinline__ void cmul(float& r, float& i, float wr, float wi) {
    float nr = r * wr - i * wi;
    float ni = r * wi + i * wr;
    r = nr; i = ni;
}

// In-register 16-point forward DFT, natural order in and out (4x4 Cooley-Tukey, n = 4a+b, k = c+4d).
__device__ __forceinline__ void fft16(float (&re)[16], float (&im)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
#pragma unroll
    for (int b = 0; b < 4; ++b) fft4(re[b], im[b], re[4 + b], im[4 + b], re[8 + b], im[8 + b], re[12 + b], im[12 + b]);
    // now element (c,b) sits at index 4c+b.  twiddle W16^(b*c)
    cmul(re[5], im[5], C1, -S1);  cmul(re[6], im[6], R, -R);    cmul(re[7], im[7], S1, -C1);
    cmul(re[9], im[9], R, -R);    { float t = re[10]; re[10] = im[10]; im[10] = -t; }   cmul(re[11], im[11], -R, -R);
    cmul(re[13], im[13], S1, -C1); cmul(re[14], im[14], -R, -R); cmul(re[15], im[15], -C1, S1);
#pragma unroll
    for (int c = 0; c < 4; ++c) fft4(re[4 * c], im[4 * c], re[4 * c + 1], im[4 * c + 1], re[4 * c + 2], im[4 * c + 2],
                                     re[4 * c + 3], im[4 * c + 3]);
    // index 4c+d now holds X[c+4d]: transpose to natural order
#define SWP(a, b) { float t = re[a]; re[a] = re[b]; re[b] = t; t = im[a]; im[a] = im[b]; im[b] = t; }
    SWP(1, 4) SWP(2, 8) SWP(3, 12) SWP(6, 9) SWP(7, 13) SWP(11, 14)
#undef SWP
}

template <typename T> __device__ __forceinline__ float load_sample(const T* p, long i);
template <> __device__ __forceinline__ float load_sample<float>(const float* p, long i) { return p[i]; }
// utils/utilities.py:66-67  int16_to_float32: x / 32767.
template <> __device__ __forceinline__ float load_sample<short>(const short* p, long i) { return (float)p[i] / 32767.0f; }

// v2 structure (occupancy first): no LDS sample staging (frames are read straight from global memory: 256-B
// coalesced segments, the 3.2x overlap is served by L2), twiddles streamed from L1-resident tables instead of living in
// 64 VGPRs, spectra / powers / mel partials aliased onto ONE 8.7 KB per-wave LDS buffer  =>  <= 128 VGPRs and 38.9 KB
// per workgroup  =>  4 workgroups (16 waves) per CU instead of 2 (8 waves).  The mel stage is balanced: the 866
// non-zero filter taps are cut into <= 12-tap tasks (105 of them) spread over the lanes in two rounds (24 iterations
// instead of 47 x 2), partial sums combined per band through LDS.
constexpr int FPW = 8;                            // frames per wave (4 FFT pairs), 32 frames per workgroup
constexpr int TASK_TAPS = 12;
constexpr int MAX_TASKS = 128;

template <typename T>
__global__ __launch_bounds__(256, 3) void logmel_kernel(const T* __restrict__ wave, int L, int T_frames,
                                                        const float* __restrict__ window,      // [16][64] = natural order
                                                        const float2* __restrict__ tw1024t,    // [16 k1][64 lane] W1024^(lane*k1)
                                                        const float2* __restrict__ tw64t,      // [16 e = i'*4+s][4 g] W64^((4i'+g)*s)
                                                        const int4* __restrict__ tasks,        // [ntasks] {lo, cnt, off, band}
                                                        int ntasks, const int2* __restrict__ bands,   // [64] {first task, #tasks}
                                                        const float* __restrict__ mel_w, int mel_nnz, float amin,
                                                        float floor_db, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* melw_s = reinterpret_cast<float*>(smem_raw);                       // [MELW_MAX]
    float2* tbuf_all = reinterpret_cast<float2*>(melw_s + MELW_MAX);          // 4 x [16*TROW]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y;
    const int frame0 = blockIdx.x * (4 * FPW) + wv * FPW;
    const T* x = wave + (long)b * L;
    for (int j = tid; j < mel_nnz; j += 256) melw_s[j] = mel_w[j];
    float win[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) win[n1] = window[64 * n1 + lane];
    const int k1 = lane >> 2, g = lane & 3;
    __syncthreads();

    float2* tb = tbuf_all + wv * 16 * TROW;
    float* pa = reinterpret_cast<float*>(tb);          // powers alias the (dead) spectrum buffer
    float* pb = pa + PSTR;
    float2* mp = reinterpret_cast<float2*>(pb + PSTR); // mel partials [MAX_TASKS]

    for (int pr = 0; pr < FPW / 2; ++pr) {
        const int ta = frame0 + 2 * pr;
        if (ta >= T_frames) break;                     // wave-uniform
        const long base_a = (long)ta * HOP - NFFT / 2; // signal index of n = 0 of frame a (frame b: + HOP)
        const bool fast = base_a >= 0 && base_a + HOP + NFFT <= (long)L && ta + 1 < T_frames;
        float re[16], im[16];
        if (fast) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                re[n1] = load_sample<T>(x, base_a + 64 * n1 + lane) * win[n1];
                im[n1] = load_sample<T>(x, base_a + HOP + 64 * n1 + lane) * win[n1];
            }
        } else {                                       // clip edges: F.pad(mode='reflect') indexing, frames past the end = 0
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                long ia = base_a + 64 * n1 + lane, ib = ia + HOP;
                if (ia < 0) ia = -ia;
                if (ia >= L) ia = 2L * (L - 1) - ia;
                if (ib < 0) ib = -ib;
                if (ib >= L) ib = 2L * (L - 1) - ib;
                float va = (ia >= 0 && ia < L) ? load_sample<T>(x, ia) : 0.f;
                float vb = (ib >= 0 && ib < L && ta + 1 < T_frames) ? load_sample<T>(x, ib) : 0.f;
                re[n1] = va * win[n1];
                im[n1] = vb * win[n1];
            }
        }
        // pass A: 16-point DFT over n1, twiddle W1024^(lane*k1)
        fft16(re, im);
#pragma unroll
        for (int k = 1; k < 16; ++k) { float2 w = tw1024t[k * 64 + lane]; cmul(re[k], im[k], w.x, w.y); }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 16; ++k) tb[k * TROW + lane] = make_float2(re[k], im[k]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i) { float2 v = tb[k1 * TROW + 4 * i + g]; re[i] = v.x; im[i] = v.y; }
        // pass B1: 4-point DFT over p for each i', twiddle W64^(q*s), q = 4i'+g
#pragma unroll
        for (int ip = 0; ip < 4; ++ip) {
            fft4(re[ip], im[ip], re[4 + ip], im[4 + ip], re[8 + ip], im[8 + ip], re[12 + ip], im[12 + ip]);
#pragma unroll
            for (int s = 1; s < 4; ++s) { float2 w = tw64t[(ip * 4 + s) * 4 + g]; cmul(re[4 * s + ip], im[4 * s + ip], w.x, w.y); }
        }
        // 4x4 transpose across the quad's lanes: lane s must own all q = 4i'+g
        float vr[16], vi[16];
#pragma unroll
        for (int ip = 0; ip < 4; ++ip) {
            float ar[4], ai[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { ar[s] = re[4 * s + ip]; ai[s] = im[4 * s + ip]; }
            {
                bool odd = g & 1;
                float s0r = odd ? ar[0] : ar[1], s0i = odd ? ai[0] : ai[1];
                float s1r = odd ? ar[2] : ar[3], s1i = odd ? ai[2] : ai[3];
                float r0r = __shfl_xor(s0r, 1, 64), r0i = __shfl_xor(s0i, 1, 64);
                float r1r = __shfl_xor(s1r, 1, 64), r1i = __shfl_xor(s1i, 1, 64);
                if (odd) { ar[0] = r0r; ai[0] = r0i; ar[2] = r1r; ai[2] = r1i; }
                else     { ar[1] = r0r; ai[1] = r0i; ar[3] = r1r; ai[3] = r1i; }
            }
            {
                bool hi = g & 2;
                float s0r = hi ? ar[0] : ar[2], s0i = hi ? ai[0] : ai[2];
                float s1r = hi ? ar[1] : ar[3], s1i = hi ? ai[1] : ai[3];
                float r0r = __shfl_xor(s0r, 2, 64), r0i = __shfl_xor(s0i, 2, 64);
                float r1r = __shfl_xor(s1r, 2, 64), r1i = __shfl_xor(s1i, 2, 64);
                if (hi) { ar[0] = r0r; ai[0] = r0i; ar[1] = r1r; ai[1] = r1i; }
                else    { ar[2] = r0r; ai[2] = r0i; ar[3] = r1r; ai[3] = r1i; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { vr[4 * ip + j] = ar[j]; vi[4 * ip + j] = ai[j]; }
        }
        // pass B2: 16-point DFT over q; lane (k1, s=g) output u -> bin k1 + 16*(s + 4u)
        fft16(vr, vi);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 16; ++u) tb[k1 + 16 * g + 64 * u] = make_float2(vr[u], vi[u]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // unpack the two real spectra and take |.|^2 into registers, then overwrite the spectrum buffer with them
        float ppa[9], ppb[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            int k = lane + 64 * i;
            ppa[i] = 0.f; ppb[i] = 0.f;
            if (k < NBINS) {
                float2 zk = tb[k], zn = tb[(NFFT - k) & (NFFT - 1)];
                float ar_ = zk.x + zn.x, ai_ = zk.y - zn.y;
                float br_ = zk.x - zn.x, bi_ = zk.y + zn.y;
                ppa[i] = 0.25f * (ar_ * ar_ + ai_ * ai_);
                ppb[i] = 0.25f * (br_ * br_ + bi_ * bi_);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            int k = lane + 64 * i;
            if (k < NBINS) { pa[k] = ppa[i]; pb[k] = ppb[i]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // mel: <= 12-tap tasks over the lanes (two rounds), then one lane per band combines its tasks
#pragma unroll
        for (int r = 0; r < MAX_TASKS / 64; ++r) {
            int t = lane + 64 * r;
            if (t < ntasks) {
                int4 tk = tasks[t];
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int i = 0; i < TASK_TAPS; ++i)
                    if (i < tk.y) {
                        float w = melw_s[tk.z + i];
                        sa = fmaf(w, pa[tk.x + i], sa);
                        sb = fmaf(w, pb[tk.x + i], sb);
                    }
                mp[t] = make_float2(sa, sb);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            int2 bd = bands[lane];
            float ma = 0.f, mb = 0.f;
            for (int j = 0; j < bd.y; ++j) { float2 v = mp[bd.x + j]; ma += v.x; mb += v.y; }
            float* o = out + ((long)b * T_frames + ta) * 64 + lane;
            // fp32 log10 (2 ulp) everywhere except AT the clamp, where the reference yields exactly 10*log10(amin)
            o[0] = ma > amin ? 10.0f * log10f(ma) : floor_db;
            if (ta + 1 < T_frames) o[64] = mb > amin ? 10.0f * log10f(mb) : floor_db;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

constexpr size_t LOGMEL_SMEM = (size_t)MELW_MAX * 4 + 4 * 16 * TROW * 8;

template <typename T>
int launch_logmel(const T* wave, int B2, int L, const float* window, const float* tw1024t, const float* tw64t,
                  const int* tasks, int ntasks, const int* bands, const float* mel_w, int mel_nnz, float amin, float* out,
                  hipStream_t stream) {
    if (B2 <= 0 || L <= NFFT / 2 || mel_nnz <= 0 || mel_nnz > MELW_MAX || ntasks <= 0 || ntasks > MAX_TASKS) return SED_EINVAL;
    int T_frames = L / HOP + 1;
    dim3 grid(sed_cdiv(T_frames, 4 * FPW), B2);
    hipLaunchKernelGGL(logmel_kernel<T>, grid, dim3(256), LOGMEL_SMEM, stream, wave, L, T_frames, window,
                       reinterpret_cast<const float2*>(tw1024t), reinterpret_cast<const float2*>(tw64t),
                       reinterpret_cast<const int4*>(tasks), ntasks, reinterpret_cast<const int2*>(bands), mel_w, mel_nnz,
                       amin, (float)(10.0 * log10((double)amin)), out);
    SED_LAUNCH_CHECK();
    return 0;
}

}  // namespace

SED_API int sed_logmel_f32(const float* wave, int B2, int L, const float* window, const float* tw1024t,
                           const float* tw64t, const int* mel_tasks, int n_tasks, const int* mel_bands,
                           const float* mel_w, int mel_nnz, float amin, float* out, hipStream_t stream) {
    return launch_logmel<float>(wave, B2, L, window, tw1024t, tw64t, mel_tasks, n_tasks, mel_bands, mel_w, mel_nnz, amin, out, stream);
}

SED_API int sed_logmel_i16(const short* wave, int B2, int L, const float* window, const float* tw1024t,
                           const float* tw64t, const int* mel_tasks, int n_tasks, const int* mel_bands,
                           const float* mel_w, int mel_nnz, float amin, float* out, hipStream_t stream) {
    return launch_logmel<short>(wave, B2, L, window, tw1024t, tw64t, mel_tasks, n_tasks, mel_bands, mel_w, mel_nnz, amin, out, stream);
}

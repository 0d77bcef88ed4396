// f16 MFMA issue-rate micro-benchmark for gfx950 (what the split-f16 convolution can reach at best on a given box):
// v_mfma_f32_32x32x16_f16 streams over 4 / 8 independent accumulators at 1 and 2 waves per SIMD, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_ubench.hip -o /tmp/mfma_f16_ubench && /tmp/mfma_f16_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// RANDOM = true: operands with random mantissas, four different register pairs in rotation -- what the split-f16 kernels
// feed the pipe (hi and lo halves of real activations).  The part clocks to its power budget, and switching activity
// is data dependent: smooth / constant operands overstate what a real kernel can sustain (MI355X_MICROARCH.md "DVFS").
template <int NACC, bool RANDOM = false>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    half8 x, y, xs[4], ys[4];
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(1.0f + j * 0.01f); }
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u; xs[q][j] = (_Float16)(((int)(h >> 8) % 20001 - 10000) * 1.0e-4f * 1.7f);
            h = h * 1664525u + 1013904223u; ys[q][j] = (_Float16)(((int)(h >> 8) % 20001 - 10000) * 1.0e-4f * 0.9f);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 36 / NACC * NACC / NACC; ++rep)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = RANDOM ? __builtin_amdgcn_mfma_f32_32x32x16_f16(xs[(a + rep) & 3], ys[(a + 2 * rep + 1) & 3], acc[a], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool RANDOM = false>
void run(const char* name, float* out, int blocks_per_cu) {
    const int iters = 2000, nb = 256 * blocks_per_cu;
    const int per_iter = (36 / NACC * NACC / NACC) * NACC;
    hipLaunchKernelGGL((k<NACC, RANDOM>), dim3(nb), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<NACC, RANDOM>), dim3(nb), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double mfmas = (double)nb * 4 * iters * per_iter;
    const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-40s %d WG/CU: %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 2516; %.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, blocks_per_cu, ms, tf,
           100 * tf / 2516.6, 2.4e9 * ms * 1e-3 / (mfmas / 1024.0));
}

// sustained mode: `mfma_f16_ubench smooth|random <seconds>` keeps the pipe busy so that rocm-smi can be sampled beside it
template <bool RANDOM>
static void sustain(float* out, double seconds) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    double done_ms = 0.0, tf = 0.0;
    long launches = 0;
    while (done_ms < seconds * 1e3) {
        (void)hipEventRecord(a);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<4, RANDOM>), dim3(256 * 2), dim3(256), 0, 0, out, 20000);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        done_ms += ms; launches += 20;
        tf = 20.0 * (256.0 * 2 * 4 * 20000 * 36) * 32768.0 / (ms * 1e-3) / 1e12;
    }
    printf("%s operands sustained %.1f s: last window %.1f TFLOP/s (%ld launches)\n", RANDOM ? "RANDOM" : "smooth", done_ms * 1e-3, tf, launches);
}

int main(int argc, char** argv) {
    float* out; (void)hipMalloc(&out, 256 * 3 * 256 * sizeof(float));
    if (argc >= 3) {
        if (argv[1][0] == 'r') sustain<true>(out, atof(argv[2])); else sustain<false>(out, atof(argv[2]));
        return 0;
    }
    run<4>("32x32x16 f16, 4 accumulators", out, 1); run<4>("32x32x16 f16, 4 accumulators", out, 2);
    run<9>("32x32x16 f16, 9 accumulators", out, 1); run<9>("32x32x16 f16, 9 accumulators", out, 2);
    run<4>("32x32x16 f16, 4 accumulators (again, warm)", out, 2);
    run<4, true>("RANDOM operands, 4 accumulators", out, 1); run<4, true>("RANDOM operands, 4 accumulators", out, 2);
    run<4, true>("RANDOM operands, 4 accumulators", out, 3); run<9, true>("RANDOM operands, 9 accumulators", out, 2);
    run<4, true>("RANDOM operands, 4 acc (again, warm)", out, 3);
    return 0;
}

// VALU issue-rate micro-benchmark for gfx950: cycles per wave-instruction per SIMD of scalar / packed fp32 ops and LDS ops,
// at 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o /tmp/valu_ubench && /tmp/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ f2 lds[256 * 4];
    f2 a[8];
    float s[8];
    for (int i = 0; i < 8; ++i) { a[i] = f2{threadIdx.x * 0.001f + i, 1.0f}; s[i] = threadIdx.x * 0.002f + i; }
    f2 w = {1.0001f, 0.9999f};
    lds[threadIdx.x] = a[0];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 32 independent-ish scalar FMAs (8 chains x 4)
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9"
                              : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "v"(w.x), "v"(w.y));)
        } else if (MODE == 1) {   // 32 packed FMAs
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(w));)
        } else if (MODE == 2) {   // 32 packed adds with op_sel / neg modifiers
            REP8(asm volatile("v_pk_add_f32 %0, %0, %4 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %4 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                              "v_pk_add_f32 %2, %2, %4 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %4 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(w));)
        } else if (MODE == 3) {   // 32 scalar adds
            REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                              : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]) : "v"(w.x));)
        } else if (MODE == 4) {   // 32 ds_read_b64 (conflict-free)
            REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:2048\n ds_read_b64 %2, %4 offset:4096\n ds_read_b64 %3, %4 offset:6144\n s_waitcnt lgkmcnt(0)"
                              : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]) : "v"((unsigned)(threadIdx.x * 8)));)
        } else if (MODE == 5) {   // 32 ds_write_b64
            REP8(asm volatile("ds_write_b64 %4, %0\n ds_write_b64 %4, %1 offset:2048\n ds_write_b64 %4, %2 offset:4096\n ds_write_b64 %4, %3 offset:6144\n s_waitcnt lgkmcnt(0)"
                              :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"((unsigned)(threadIdx.x * 8)) : "memory");)
        } else if (MODE == 6) {   // 16 packed FMAs interleaved with 16 scalar FMAs
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_fma_f32 %2, %2, %5, %6\n v_pk_fma_f32 %1, %1, %4, %4\n v_fma_f32 %3, %3, %5, %6"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(s[0]), "+v"(s[1]) : "v"(w), "v"(w.x), "v"(w.y));)
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, float* out) {
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {      // waves per SIMD = blocks per CU (256 threads = 1 wave per SIMD)
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, 10);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double cyc = ms * 1e-3 * 2.4e9 / (iters * 32.0 * wps);
        printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wps, cyc);
    }
}
int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    run<0>("v_fma_f32", out); run<3>("v_add_f32", out); run<1>("v_pk_fma_f32", out); run<2>("v_pk_add_f32 (op_sel, neg)", out);
    run<6>("v_pk_fma_f32 + v_fma_f32", out); run<4>("ds_read_b64", out); run<5>("ds_write_b64", out);
    return 0;
}

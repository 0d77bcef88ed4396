// STANDALONE reproduction of the packed-fp32-beside-f16-MFMA observation: no kernel of this package is
// involved.  Stream A runs a probe kernel whose lanes evaluate packed-fp32 instructions in the operand forms in question and
// the same arithmetic with scalar fp32 instructions (IEEE: bit-identical); stream B runs a BARE MFMA stream (no memory, no
// LDS) -- v_mfma_f32_32x32x16_f16 with smooth or with random operands, v_mfma_f32_32x32x16_bf16, v_mfma_f32_32x32x2_f32, or
// nothing.  Smooth vs random operands change the power draw and the clock the part settles at (tools/mfma_power_probe.sh:
// 2396 MHz / 1213 W vs 1745 MHz / 1285 W): if the mismatches were a voltage-droop effect they would follow the POWER of the
// aggressor; if they follow the aggressor's INSTRUCTION (f16 vs f32 MFMA) at either power level, it is a structural hazard.
// The runner script lowers the power cap with rocm-smi for a third data point where the driver permits it.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_f32_standalone_repro.hip -o /tmp/pkrepro && /tmp/pkrepro [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define FORMS 8
static const char* const FORM_NAMES[FORMS] = {
    "plain fma", "add src1 swapped [0,1]", "add src0 swapped [1,0]", "add src1.hi->lo + neg [0,1]", "add src0.hi->lo + neg [1,0]",
    "fma src1.hi broadcast [0,1,0]", "fma src0.hi broadcast [1,0,0]", "mul src1 swapped [0,1]"};

__global__ __launch_bounds__(256) void pk_probe(unsigned* bad, int iters, unsigned seed) {
    const float s0 = (float)(((threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u + seed) >> 8) * (1.0f / 16777216.0f) + 0.25f;
    unsigned nb[FORMS] = {0, 0, 0, 0, 0, 0, 0, 0};
    f2 b = {0.9990234375f, 1.0009765625f}, c = {0.001953125f, -0.0029296875f};
    float ax = s0, ay = s0 * 1.37f;
    for (int it = 0; it < iters; ++it) {
        const f2 a = {ax, ay};
        f2 r;
#define CHECK(k, ex, ey) if (r.x != (ex) || r.y != (ey)) ++nb[k];
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        CHECK(0, __builtin_fmaf(ax, b.x, c.x), __builtin_fmaf(ay, b.y, c.y))
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(c));      // FRAGILE form
        CHECK(1, ax + c.y, ay + c.x)
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(c), "v"(a));      // same sums, high-half read on src0
        CHECK(2, c.y + ax, c.x + ay)
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(c));   // FRAGILE (FFT a + (-i) b)
        CHECK(3, ax + c.y, ay - c.x)
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(c), "v"(a));   // mirrored: safe
        CHECK(4, c.y + ax, -c.x + ay)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // FRAGILE
        CHECK(5, __builtin_fmaf(ax, b.y, c.x), __builtin_fmaf(ay, b.y, c.y))
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(b), "v"(a), "v"(c));   // mirrored: safe
        CHECK(6, __builtin_fmaf(b.y, ax, c.x), __builtin_fmaf(b.y, ay, c.y))
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));      // FRAGILE
        CHECK(7, ax * b.y, ay * b.x)
        ax = __builtin_fmaf(ax, 0.99951171875f, 0.00048828125f * (float)(it & 7)); ay = __builtin_fmaf(ay, 1.00048828125f, -0.000732421875f);
    }
#pragma unroll
    for (int k = 0; k < FORMS; ++k) if (nb[k]) atomicAdd(&bad[k], nb[k]);
}

// bare MFMA streams: MODE 0 = f16 smooth operands, 1 = f16 random operands, 2 = bf16 random, 3 = fp32 (32x32x2) random;
// NACC independent accumulators (1 = a dependent chain: every MFMA waits for its predecessor, the pipe is ~half idle)
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void aggressor(float* out, int iters) {
    floatx16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = (float)(a + 1) * 0.125f;      // distinct: the compiler must keep all of them
    half8 xs[4], ys[4];
    bf16x8 xb[4], yb[4];
    float xf[4], yf[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 4; ++q) {
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u; const float u = ((int)(h >> 8) % 20001 - 10000) * 1.0e-4f * 1.7f;
            h = h * 1664525u + 1013904223u; const float v = ((int)(h >> 8) % 20001 - 10000) * 1.0e-4f * 0.9f;
            xs[q][j] = MODE == 0 ? (_Float16)(threadIdx.x * 0.001f + j) : (_Float16)u;
            ys[q][j] = MODE == 0 ? (_Float16)(1.0f + j * 0.01f) : (_Float16)v;
            xb[q][j] = (__bf16)u; yb[q][j] = (__bf16)v;
            xf[q] = u; yf[q] = v;
        }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 9; ++rep)
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) {
                const int a = NACC == 1 ? 0 : a4;
                const int i = MODE == 0 ? 0 : (a4 + rep) & 3, j = MODE == 0 ? 0 : (a4 + 2 * rep + 1) & 3;
                if (MODE <= 1) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xs[i], ys[j], acc[a], 0, 0, 0);
                else if (MODE == 2) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[i], yb[j], acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[i], yf[j], acc[a], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// f16 MFMA stream whose operands are re-read from LDS before every group of four MFMAs (ds_read_b128, random data): the shape
// of the package's convolution kernels (fragment reads + MFMAs), still without any global memory traffic
// MF = 0: the ds_read_b128 stream alone (values summed on the VALU), 1: + f16 MFMAs, 2: + fp32 MFMAs
template <int MF>
__global__ __launch_bounds__(256) void aggressor_lds(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[16 * 1024];          // 32 KB
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 777u;
    for (int i = threadIdx.x; i < 16 * 1024; i += 256) {
        h = h * 1664525u + 1013904223u;
        lds[i] = (_Float16)(((int)(h >> 8) % 20001 - 10000) * 1.0e-4f * 1.3f);
    }
    __syncthreads();
    floatx16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = (float)(a + 1) * 0.125f;
    const half8* base = reinterpret_cast<const half8*>(lds);
    int off = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 9; ++rep) {
            const half8 a0 = base[(off + rep * 64) & 2047], a1 = base[(off + rep * 64 + 517) & 2047];
            const half8 b0 = base[(off + rep * 64 + 1031) & 2047], b1 = base[(off + rep * 64 + 1543) & 2047];
            if (MF == 1) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
            } else if (MF == 2) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a0[0], (float)b0[1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a0[2], (float)b1[3], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a1[4], (float)b0[5], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a1[6], (float)b1[7], acc[3], 0, 0, 0);
            } else {
                acc[0][rep] += (float)a0[0] + (float)b0[1]; acc[1][rep] += (float)a0[2] + (float)b1[3];
                acc[2][rep] += (float)a1[4] + (float)b0[5]; acc[3][rep] += (float)a1[6] + (float)b1[7];
            }
        }
        off = (off + 37) & 2047;
    }
    float sum = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) sum += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 400;
    unsigned* bad; float* out;
    (void)hipMalloc(&bad, 64); (void)hipMalloc(&out, 512 * 256 * sizeof(float));
    hipStream_t sa, sb; (void)hipStreamCreate(&sa); (void)hipStreamCreate(&sb);
    const char* names[10] = {"nothing", "f16 MFMA 32x32x16, 4 accumulators, SMOOTH operands (high clock, lower power)",
                            "f16 MFMA 32x32x16, 4 accumulators, RANDOM operands (power-limited clock)",
                            "f16 MFMA 32x32x16, dependent chain, SMOOTH operands", "f16 MFMA 32x32x16, dependent chain, RANDOM operands",
                            "bf16 MFMA 32x32x16, 4 accumulators, RANDOM operands", "fp32 MFMA 32x32x2, 4 accumulators, RANDOM operands",
                            "f16 MFMA 32x32x16 fed by ds_read_b128 from LDS (random data): the convolution kernels' shape",
                             "the same ds_read_b128 stream WITHOUT any MFMA (VALU adds)", "the same ds_read_b128 stream feeding fp32 MFMAs 32x32x2"};
    for (int ag = 0; ag < 10; ++ag) {
        (void)hipMemset(bad, 0, 64);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, sa);
        for (int it = 0; it < launches; ++it) {
            // the aggressor fills HALF of each CU's wave slots (1 workgroup of 4 waves per CU x 256 CUs); the probe takes the rest
            if (ag == 1) hipLaunchKernelGGL((aggressor<0, 4>), dim3(256), dim3(256), 0, sb, out, 1500);
            if (ag == 2) hipLaunchKernelGGL((aggressor<1, 4>), dim3(256), dim3(256), 0, sb, out, 1500);
            if (ag == 3) hipLaunchKernelGGL((aggressor<0, 1>), dim3(256), dim3(256), 0, sb, out, 750);
            if (ag == 4) hipLaunchKernelGGL((aggressor<1, 1>), dim3(256), dim3(256), 0, sb, out, 750);
            if (ag == 5) hipLaunchKernelGGL((aggressor<2, 4>), dim3(256), dim3(256), 0, sb, out, 1500);
            if (ag == 6) hipLaunchKernelGGL((aggressor<3, 4>), dim3(256), dim3(256), 0, sb, out, 200);
            if (ag == 7) hipLaunchKernelGGL((aggressor_lds<1>), dim3(512), dim3(256), 0, sb, out, 750);
            if (ag == 8) hipLaunchKernelGGL((aggressor_lds<0>), dim3(512), dim3(256), 0, sb, out, 750);
            if (ag == 9) hipLaunchKernelGGL((aggressor_lds<2>), dim3(512), dim3(256), 0, sb, out, 750);
            hipLaunchKernelGGL(pk_probe, dim3(512), dim3(256), 0, sa, bad, 100, (unsigned)(it * 7919));
        }
        (void)hipEventRecord(e1, sa);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned h[FORMS];
        (void)hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
        printf("beside %-78s %.0f ms, %.3g evaluations per form:", names[ag], ms, (double)launches * 512 * 256 * 100);
        for (int k = 0; k < FORMS; ++k) printf("  %s: %u", FORM_NAMES[k], h[k]);
        printf("\n");
    }
    return 0;
}

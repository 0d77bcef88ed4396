// Packed-fp32 VALU forms beside f16 MFMAs (MI355X / gfx950 observation, round 3).
// Every lane evaluates packed-fp32 instructions in several operand-modifier forms and the same arithmetic with scalar fp32
// instructions; IEEE says they are bit-identical.  Mismatches are counted per form.  Run it alone and with a kernel that
// issues v_mfma_f32_32x32x16_f16 on a second stream (tools/pk_f32_beside_mfma_probe.py).
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define FORMS 14
__global__ __launch_bounds__(256) void pk_probe(unsigned* bad, int iters, unsigned seed) {
    const float s0 = (float)(((threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u + seed) >> 8) * (1.0f / 16777216.0f) + 0.25f;
    unsigned nb[FORMS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f2 b = {0.9990234375f, 1.0009765625f}, c = {0.001953125f, -0.0029296875f};
    float ax = s0, ay = s0 * 1.37f;
    for (int it = 0; it < iters; ++it) {
        const f2 a = {ax, ay};
        f2 r;
#define CHECK(k, ex, ey) if (r.x != (ex) || r.y != (ey)) ++nb[k];
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));                                   // 0 plain fma
        CHECK(0, __builtin_fmaf(ax, b.x, c.x), __builtin_fmaf(ay, b.y, c.y))
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(c));                                                // 1 plain add
        CHECK(1, ax + c.x, ay + c.y)
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(c));                   // 2 add, src1 halves swapped
        CHECK(2, ax + c.y, ay + c.x)
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(c));                                   // 3 add, neg only
        CHECK(3, ax + c.x, ay - c.y)
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(c));      // 4 add, swapped + neg (FFT: a + (-i) b)
        CHECK(4, ax + c.y, ay - c.x)
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));                                // 5 mul, src0 low half broadcast (compiler form)
        CHECK(5, ax * b.x, ax * b.y)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));                  // 6 fma, src1 / src2 low halves broadcast (compiler form)
        CHECK(6, __builtin_fmaf(ax, b.x, c.x), __builtin_fmaf(ay, b.x, c.x))
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));                   // 7 mul, src1 halves swapped
        CHECK(7, ax * b.y, ay * b.x)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // 8 fma, src1 halves swapped
        CHECK(8, __builtin_fmaf(ax, b.y, c.x), __builtin_fmaf(ay, b.x, c.y))
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(c));                   // 9 add, src0 halves swapped
        CHECK(9, ay + c.x, ax + c.y)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // 10 fma, src1 high half broadcast
        CHECK(10, __builtin_fmaf(ax, b.y, c.x), __builtin_fmaf(ay, b.y, c.y))
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // 11 complex multiply tail: src0 swapped, src1 high broadcast, neg
        CHECK(11, __builtin_fmaf(-ay, b.y, c.x), __builtin_fmaf(ax, b.y, c.y))
        {
            const f2 sw = {0.70710678118654752f, -0.38268342614173889f};
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(sw));                                           // 12 mul, src1 = SGPR pair
            CHECK(12, ax * sw.x, ay * sw.y)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(sw), "v"(c));   // 13 the same tail with an SGPR twiddle
            CHECK(13, __builtin_fmaf(-ay, sw.y, c.x), __builtin_fmaf(ax, sw.y, c.y))
        }
        ax = __builtin_fmaf(ax, 0.99951171875f, 0.00048828125f * (float)(it & 7)); ay = __builtin_fmaf(ay, 1.00048828125f, -0.000732421875f);
    }
#pragma unroll
    for (int k = 0; k < FORMS; ++k) if (nb[k]) atomicAdd(&bad[k], nb[k]);
}
extern "C" int pk_probe_launch(unsigned* bad, int blocks, int iters, unsigned seed, hipStream_t s) {
    hipLaunchKernelGGL(pk_probe, dim3(blocks), dim3(256), 0, s, bad, iters, seed);
    return (int)hipGetLastError();
}

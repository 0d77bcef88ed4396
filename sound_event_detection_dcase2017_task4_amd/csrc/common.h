// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SED_API extern "C" __attribute__((visibility("default")))

// Every C-ABI entry point returns 0 on success or a hipError_t / negative argument-error code.
#define SED_EINVAL (-22)
#define SED_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

static inline int sed_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// The ONE expression used everywhere for "BatchNorm (folded to scale/shift) then ReLU", so the
// forward value and every recomputed backward mask agree bit for bit.
__device__ __forceinline__ float bn_relu(float y, float scale, float shift) {
    return fmaxf(fmaf(y, scale, shift), 0.0f);
}
__device__ __forceinline__ bool bn_relu_active(float y, float scale, float shift) {
    return fmaf(y, scale, shift) > 0.0f;
}

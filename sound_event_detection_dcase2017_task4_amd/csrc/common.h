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

// Every object of the library records the hash of the hipcc flags it was compiled with (build.py defines it); sed_version()
// reports it, "mixed" when the objects disagree, and the Python loader refuses anything but the default flags unless
// SED_ALLOW_EXPERIMENT=1 -- a library built with a timing-experiment -D can no longer be picked up silently.
#ifndef SED_BUILD_FLAGS_HASH
#define SED_BUILD_FLAGS_HASH "unknown"
#endif
#define SED_OBJECT_FLAGS(name) extern "C" __attribute__((visibility("hidden"))) const char sed_objflags_##name[] = SED_BUILD_FLAGS_HASH;
// every translation unit of libsed_hip.so besides heads.hip, which holds sed_version() and compares their flags hashes with
// its own (= build.SOURCES without the suffix, minus heads: tests/test_capi_and_host.py keeps the two lists equal)
#define SED_OBJECTS(X) X(logmel) X(bn) X(conv) X(conv_wino2) X(conv_sf16) X(gemm_sf16) X(attention) X(gru)

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

// Device-resident amax values (the split-f16 operand scales) are SED_AMAX_SLOTS floats, not one: a producer block
// publishes its maximum (>= 0, so the uint order of the bits is the float order) with ONE atomic on slot (block id mod
// slots), a consumer takes the maximum over the slots.  Thousands of atomics on a single word serialise in the L2 at
// ~12 ns each: bn_bwd_apply with one atomic per wave spent 190 us per launch in them at batch 32 (profiles/r03).
#define SED_AMAX_SLOTS 64
__device__ __forceinline__ void amax_publish_block(float* __restrict__ out, float v) {   // every thread of the block calls
    __shared__ float amax_wm__[16];
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) amax_wm__[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = amax_wm__[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, amax_wm__[w]);
        atomicMax(reinterpret_cast<unsigned*>(out) + (blockIdx.x & (SED_AMAX_SLOTS - 1)), __float_as_uint(m));
    }
}
__device__ __forceinline__ float amax_read(const float* __restrict__ p) { return wave_max(p[threadIdx.x & (SED_AMAX_SLOTS - 1)]); }
// Callers that hand in amax buffers carved out of a pool they zero themselves register the pool's address range
// (sed_amax_prezeroed_range); only pointers inside a registered range skip the memset -- any other buffer is zeroed here.
bool sed_amax_is_prezeroed__(const float* p);
static inline hipError_t sed_amax_clear(float* amax_out, hipStream_t stream) {
    return (amax_out && !sed_amax_is_prezeroed__(amax_out)) ? hipMemsetAsync(amax_out, 0, SED_AMAX_SLOTS * sizeof(float), stream)
                                                            : hipSuccess;
}

// Split-f16 operand format (csrc/conv_sf16.hip): x = (hi + lo) / s with hi = f16(s*x), lo = f16(s*x - hi), s = the power of two
// that brings the tensor's amax (or an upper BOUND of it) to [2^13, 2^14).  A tensor stored "as pairs" holds, per channel
// pair, two dwords {hi0 | hi1 << 16, lo0 | lo1 << 16}: the bytes of the fp32 tensor, and the consumers' staging is a copy.
__device__ __forceinline__ float sed_sf_scale_of(float amax) {
    if (!(amax > 0.f) || !(amax < __builtin_inff())) return 1.f;
    int e;
    frexpf(amax, &e);
    e = 14 - e;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return ldexpf(1.f, e);
}
__device__ __forceinline__ void sed_sf_split2(float a, float b, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(a), "v"(b));
}
// four scaled values -> the 16 bytes {h01, l01, h23, l23} of their two channel pairs, streamed out
__device__ __forceinline__ void sed_store_pairs4(void* base, long idx4, float4 v) {
    unsigned h01, l01, h23, l23;
    sed_sf_split2(v.x, v.y, h01, l01);
    sed_sf_split2(v.z, v.w, h23, l23);
    const floatx4 ov = {__uint_as_float(h01), __uint_as_float(l01), __uint_as_float(h23), __uint_as_float(l23)};
    __builtin_nontemporal_store(ov, reinterpret_cast<floatx4*>(base) + idx4);
}
// ... and back: the four fp32 values (times s) of such 16 bytes
__device__ __forceinline__ float4 sed_load_pairs4(const void* base, long idx4) {
    const floatx4 r = reinterpret_cast<const floatx4*>(base)[idx4];
    const unsigned h01 = __float_as_uint(r[0]), l01 = __float_as_uint(r[1]), h23 = __float_as_uint(r[2]), l23 = __float_as_uint(r[3]);
    auto f = [](unsigned w, int hi) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(hi ? (w >> 16) : (w & 0xffffu))); };
    return make_float4(f(h01, 0) + f(l01, 0), f(h01, 1) + f(l01, 1), f(h23, 0) + f(l23, 0), f(h23, 1) + f(l23, 1));
}

// Streaming 16-byte store of a tensor that is written once and read by a LATER kernel after everything else has passed through the
// caches: non-temporal (conv1_fwd writes its 4.2 GB at 4.5 instead of 3.4 TB/s with it: profiles/r03).
__device__ __forceinline__ void store_nt4(float* base, long idx4, float4 v) {
    const floatx4 ov = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(ov, reinterpret_cast<floatx4*>(base) + idx4);
}

__device__ __forceinline__ float4 load_nt4(const float* base, long idx4) {
    const floatx4 v = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(base) + idx4);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// The ONE expression used everywhere for "BatchNorm (folded to scale/shift) then ReLU", so the
// forward value and every recomputed backward mask agree bit for bit.
__device__ __forceinline__ float bn_relu(float y, float scale, float shift) {
    return fmaxf(fmaf(y, scale, shift), 0.0f);
}
__device__ __forceinline__ bool bn_relu_active(float y, float scale, float shift) {
    return fmaf(y, scale, shift) > 0.0f;
}

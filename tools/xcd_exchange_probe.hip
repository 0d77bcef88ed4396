// What does one step of the GRU's hand-over THROUGH THE DATA cost, and how much of it is the trip to the device-wide coherence
// point?  8 workgroups of a group (ids g + 16 j: one XCD when the dispatcher deals workgroups round-robin over the 8 XCDs)
// each publish 4 KB per step and then poll the group's 32 KB until no sentinel is left -- the all-gather of csrc/gru.hip
// without its arithmetic.  Variants: loads `sc1` (agent scope: served by the coherence point) / `sc0` (workgroup scope: vector
// L1 bypassed, served by the XCD's own L2) / `sc0 sc1`; stores agent-scope atomics / plain / `sc0`.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_probe tools/xcd_exchange_probe.hip && /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4r __attribute__((ext_vector_type(4)));
constexpr unsigned SENT = 0x7fc0deadu;
constexpr int MEMBERS = 8, GROUPS = 16, WG_FLOATS = 1024;      // 4 KB per workgroup and step

template <int LD> __device__ __forceinline__ void ld4(f4r& d, const float* p) {
    if (LD == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(d) : "v"(p) : "memory");
    if (LD == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(d) : "v"(p) : "memory");
    if (LD == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(d) : "v"(p) : "memory");
    if (LD == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(d) : "v"(p) : "memory");
    if (LD == 4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(d) : "v"(p) : "memory");
    if (LD == 5) asm volatile("buffer_inv sc0\n\tglobal_load_dwordx4 %0, %1, off sc0" : "=v"(d) : "v"(p) : "memory");
}
template <int ST> __device__ __forceinline__ void st2(float* p, float2 v) {
    if (ST == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (ST == 4) asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
}

template <int LD, int ST, bool SPREAD>
__global__ __launch_bounds__(512) void probe(float* buf, int T, int* xcc, float* sink, int* err) {
    // SPREAD: the members of a group are 8 CONSECUTIVE workgroups = one on each XCD
    const int g = SPREAD ? blockIdx.x / MEMBERS : blockIdx.x % GROUPS, j = SPREAD ? blockIdx.x % MEMBERS : blockIdx.x / GROUPS, tid = threadIdx.x;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = (int)(id & 0xf);
    }
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        float* step = buf + ((long)t * GROUPS + g) * MEMBERS * WG_FLOATS;
        // publish: the value depends on what was gathered in the step before (a true dependency chain)
        const float v = (float)(t + 1) + acc * 1e-30f;
        st2<ST>(step + j * WG_FLOATS + 2 * tid, float2{v, v + 0.5f});
        f4r a[4];
        long polls = 0;
        for (;;) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ld4<LD>(a[q], step + q * 2048 + 4 * tid);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(a[q]));
            bool bad = false;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                for (int i = 0; i < 4; ++i) bad |= __float_as_uint(a[q][i]) == SENT;
            if (__builtin_amdgcn_ballot_w64(bad) == 0) break;
            __builtin_amdgcn_s_sleep(1);
            if (++polls > 200000 || ((polls & 63) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { if ((tid & 63) == 0) atomicAdd(err, 1); break; }
        }
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += a[q][0] + a[q][1] + a[q][2] + a[q][3];
        acc += s;
        __syncthreads();                                   // the GRU's per-step workgroup barrier (K reduction through LDS)
    }
    sink[blockIdx.x * 512 + tid] = acc;
}

template <int LD, int ST, bool SPREAD = false>
void run(const char* name, int T) {
    float *buf, *sink; int *xcc, *err;
    const size_t n = (size_t)T * GROUPS * MEMBERS * WG_FLOATS;
    hipMalloc(&buf, n * 4); hipMalloc(&sink, 128 * 512 * 4); hipMalloc(&xcc, 128 * 4); hipMalloc(&err, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemsetD32((hipDeviceptr_t)buf, SENT, n); hipMemset(err, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<LD, ST, SPREAD>), dim3(128), dim3(512), 0, 0, buf, T, xcc, sink, err);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<float> h(128 * 512); std::vector<int> hx(128); int herr;
    hipMemcpy(h.data(), sink, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xcc, 128 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    // expected: every thread gathered sum_t 16 values: 8 x (t+1) + 8 x (t+1.5) = 16 (t+1) + 4
    double expect = 0; for (int t = 0; t < T; ++t) expect += 16.0 * (t + 1) + 4.0;
    int wrong = 0; for (float v : h) wrong += !(fabs(v - expect) <= 1e-3 * expect);
    int split = 0;
    for (int g = 0; g < GROUPS; ++g) for (int j = 1; j < MEMBERS; ++j) split += SPREAD ? hx[g * MEMBERS + j] != hx[g * MEMBERS] : hx[g + GROUPS * j] != hx[g];
    printf("%-46s %7.3f us/step   wrong threads %d, timeouts %d, group members off their group's XCD %d (xcc of wg 0..15:", name, best * 1e3 / T, wrong, herr, split);
    for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
    printf(")\n");
    hipFree(buf); hipFree(sink); hipFree(xcc); hipFree(err);
}

int main() {
    const int T = 2000;
    run<0, 0>("loads sc1, stores agent-scope atomic (shipped)", T);
    run<0, 3>("loads sc1, stores sc1", T);
    run<0, 1>("loads sc1, stores plain", T);
    run<0, 2>("loads sc1, stores sc0", T);
    run<0, 4>("loads sc1, stores nt", T);
    run<3, 1>("loads sc1 nt, stores plain", T);
    run<2, 1>("loads sc0 sc1, stores plain", T);
    run<4, 1>("loads nt, stores plain", T);
    run<5, 1>("buffer_inv sc0 + loads sc0, stores plain", T);
    run<0, 0, true>("SPREAD over 8 XCDs: loads sc1, stores agent atomic", T);
    run<0, 3, true>("SPREAD over 8 XCDs: loads sc1, stores sc1", T);
    run<0, 1, true>("SPREAD over 8 XCDs: loads sc1, stores plain", T);
    run<2, 1, true>("SPREAD over 8 XCDs: loads sc0 sc1, stores plain", T);
    return 0;
}

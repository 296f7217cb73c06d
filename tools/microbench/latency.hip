// Dependent-load latency on one wave: global (by footprint) and LDS.  Diagnostic tool, not product.
// hipcc --offload-arch=gfx950 -O3 -o latency latency.hip && ./latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ void chase_global(const unsigned* __restrict__ buf, unsigned start, int iters, unsigned long long* out, unsigned* sink) {
    unsigned idx = start + threadIdx.x * 0;  // all lanes chase the same chain (single address per load)
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) { asm volatile("" : "+v"(idx)); idx = buf[idx]; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; sink[0] = idx; }
}
// each lane chases its own chain (64 different lines per load, like a gather)
__global__ void chase_global_div(const unsigned* __restrict__ buf, int n, int iters, unsigned long long* out, unsigned* sink) {
    unsigned idx = (threadIdx.x * 977u) % n;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) idx = buf[idx];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    sink[threadIdx.x] = idx;
}
// dependent 8-byte loads (global_load_dwordx2) at 8-byte aligned or 4-byte aligned addresses
struct __attribute__((packed, aligned(4))) f2u { unsigned a, b; };
__global__ void chase_x2(const unsigned* __restrict__ buf, int n, int iters, int odd, unsigned long long* out, unsigned* sink) {
    unsigned idx = ((threadIdx.x * 977u) % (n / 2)) * 2 + odd;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    for (int i = 0; i < iters; i++) {
        const f2u v = *reinterpret_cast<const f2u*>(buf + idx);
        acc += v.b;
        idx = v.a;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    sink[threadIdx.x] = idx + acc;
}
__global__ void chase_lds(int iters, unsigned long long* out, unsigned* sink) {
    __shared__ unsigned l[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) l[i] = (i * 61 + 17) & 4095;
    __syncthreads();
    unsigned idx = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) idx = l[idx];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    sink[threadIdx.x] = idx;
}
__global__ void valu_chain(int iters, unsigned long long* out, float* sink) {
    float x = threadIdx.x * 0.001f + 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) { x = __builtin_fmaf(x, 1.0001f, 0.5f); x = __builtin_fmaf(x, 0.9999f, -0.5f); x = __builtin_fmaf(x, 1.0001f, 0.25f); x = __builtin_fmaf(x, 0.9999f, -0.25f); }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    sink[threadIdx.x] = x;
}

int main() {
    unsigned long long* d_out; unsigned* d_sink; float* d_fs;
    hipMalloc(&d_out, 16); hipMalloc(&d_sink, 4096); hipMalloc(&d_fs, 4096);
    const int iters = 20000;
    std::mt19937 rng(1);
    for (size_t bytes : {4096ul, 16384ul, 262144ul, 2097152ul, 33554432ul, 268435456ul, 1073741824ul}) {
        size_t n = bytes / 4;
        // one random cycle over line-granular slots (128 B apart) so every hop is a new line
        size_t lines = n / 32;
        std::vector<unsigned> perm(lines); std::iota(perm.begin(), perm.end(), 0u); std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<unsigned> h(n, 0);
        for (size_t i = 0; i < lines; i++) h[(size_t)perm[i] * 32] = perm[(i + 1) % lines] * 32;
        unsigned* d; hipMalloc(&d, bytes); hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        unsigned long long c[2] = {0, 0};
        for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(chase_global, 1, 64, 0, 0, d, perm[0] * 32, iters, d_out, d_sink); hipMemcpy(c, d_out, 16, hipMemcpyDeviceToHost); }
        printf("global chase, footprint %10zu B (%zu lines): %7.1f ticks/load = %6.1f ns (one address per wave, vector load)\n", bytes, lines, (double)c[0] / iters, (double)c[1] * 10.0 / iters);
        // divergent: random permutation of all words
        if (bytes <= 33554432ul) {
            std::vector<unsigned> p2(n); std::iota(p2.begin(), p2.end(), 0u); std::shuffle(p2.begin(), p2.end(), rng);
            std::vector<unsigned> h2(n);
            for (size_t i = 0; i < n; i++) h2[p2[i]] = p2[(i + 1) % n];
            hipMemcpy(d, h2.data(), bytes, hipMemcpyHostToDevice);
            for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(chase_global_div, 1, 64, 0, 0, d, (int)n, iters, d_out, d_sink); hipMemcpy(c, d_out, 16, hipMemcpyDeviceToHost); }
            printf("global chase, footprint %10zu B: %7.1f ticks/load = %6.1f ns (64 divergent lanes)\n", bytes, (double)c[0] / iters, (double)c[1] * 10.0 / iters);
        }
        hipFree(d);
    }
    for (int odd = 0; odd < 2; odd++) {
        // 4 KB buffer: a random cycle over the even (or odd) word slots; the second word is junk
        const size_t n = 1024;
        std::vector<unsigned> slots(n / 2); std::iota(slots.begin(), slots.end(), 0u); std::shuffle(slots.begin(), slots.end(), rng);
        std::vector<unsigned> h(n + 2, 7u);
        for (size_t i = 0; i < slots.size(); i++) h[slots[i] * 2 + odd] = slots[(i + 1) % slots.size()] * 2 + odd;
        unsigned* d; hipMalloc(&d, (n + 2) * 4); hipMemcpy(d, h.data(), (n + 2) * 4, hipMemcpyHostToDevice);
        unsigned long long c2[2] = {0, 0};
        for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(chase_x2, 1, 64, 0, 0, d, (int)n, iters, odd, d_out, d_sink); hipMemcpy(c2, d_out, 16, hipMemcpyDeviceToHost); }
        printf("global dwordx2 chase, 4 KB, %s: %7.1f ticks/load = %6.1f ns (64 divergent lanes)\n", odd ? "4-byte aligned (odd word)" : "8-byte aligned", (double)c2[0] / iters, (double)c2[1] * 10.0 / iters);
        hipFree(d);
    }
    unsigned long long c[2] = {0, 0};
    for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(chase_lds, 1, 64, 0, 0, iters, d_out, d_sink); hipMemcpy(c, d_out, 16, hipMemcpyDeviceToHost); }
    printf("LDS chase (ds_read_b32): %7.1f ticks/load = %6.1f ns\n", (double)c[0] / iters, (double)c[1] * 10.0 / iters);
    for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(valu_chain, 1, 64, 0, 0, iters, d_out, d_fs); hipMemcpy(c, d_out, 16, hipMemcpyDeviceToHost); }
    printf("dependent v_fma chain: %7.2f ticks/fma = %6.2f ns (one wave)\n", (double)c[0] / (iters * 4), (double)c[1] * 10.0 / (iters * 4));
    return 0;
}

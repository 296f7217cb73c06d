// How many waves per microsecond can MI355X's dispatchers START?  One-wave (or four-wave) workgroups that end at once, with the
// register footprint of the march kernel (64 VGPRs -> 8 waves per SIMD) or a small one; grid of 4 M waves.
// hipcc --offload-arch=gfx950 -O3 -o wave_rate wave_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VGPRS, int WORK>
__global__ __launch_bounds__(256) void k_exit(float* out, int n) {
    float acc[VGPRS];
#pragma unroll
    for (int i = 0; i < VGPRS; i++) acc[i] = (float)(threadIdx.x + i);
    if (n < 0) { // never true: keeps the registers allocated
#pragma unroll
        for (int i = 0; i < VGPRS; i++) out[threadIdx.x * VGPRS + i] = acc[i];
    }
    if (WORK > 0) { // a little dependent work per wave
        float x = (float)blockIdx.x;
        for (int i = 0; i < WORK; i++) x = __builtin_fmaf(x, 1.0001f, 0.5f);
        if (x == 12345.678f) out[0] = x;
    }
}

template <int VGPRS, int WORK>
void run(const char* name, int threads, float* d) {
    const long waves = 4L << 20;
    const long blocks = waves * 64 / threads;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_exit<VGPRS, WORK>), dim3((unsigned)blocks), dim3(threads), 0, 0, d, 1);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-44s %4d threads/WG: %8.3f ms for %ld waves = %7.1f waves/us (%5.1f per XCD)\n", name, threads, ms, waves, waves / (ms * 1e3), waves / (ms * 1e3) / 8);
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 24);
    run<4, 0>("4 VGPRs, ends at once", 64, d);
    run<4, 0>("4 VGPRs, ends at once", 256, d);
    run<60, 0>("60 live VGPRs (64 allocated), ends at once", 64, d);
    run<60, 0>("60 live VGPRs (64 allocated), ends at once", 256, d);
    run<4, 200>("4 VGPRs, 200 dependent FMAs", 64, d);
    run<4, 2000>("4 VGPRs, 2000 dependent FMAs", 64, d);
    run<60, 2000>("60 VGPRs, 2000 dependent FMAs", 64, d);
    return 0;
}

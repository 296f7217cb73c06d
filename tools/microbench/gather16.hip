// 16-bit brick records (round 2): throughput and correctness of the tap fetch the march would use on 256-B bricks
// (5^3 int16 samples, in-brick index lx*25 + lz*5 + ly): the 8 taps of a cell sit at halfwords l + {0,1,5,6,25,26,30,31},
// i.e. at 2-byte-aligned byte offsets.  Variants:
//   A  4 x global_load_dword at 2-byte alignment (one per y-pair)
//   B  2 x global_load_dwordx4 at 2-byte alignment (one per x-plane: halfwords 0,1,5,6 of the 8 loaded)
// and, for reference, the round-1 fetch on 512-B fp32 bricks (4 x dwordx2).  First checks that unaligned multi-dword
// global loads return the right bytes on this chip.  Diagnostic tool, not product.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o gather16 gather16.hip && ./gather16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef const char __attribute__((address_space(1))) * gchar_p;
struct __attribute__((packed, aligned(2))) U1 { uint32_t v; };
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lerp1(float a, float b, float w) { return __builtin_fmaf(w, b - a, a); }
__device__ __forceinline__ float lo16(unsigned w) { return (float)(int16_t)(w & 0xffffu); }
__device__ __forceinline__ float hi16(unsigned w) { return (float)(int16_t)(w >> 16); }

__device__ __forceinline__ uint4v load_x4_unaligned(const char* p) {
    uint4v r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}

__global__ void check_kernel(const char* __restrict__ base, unsigned* __restrict__ out) {
    const unsigned o = threadIdx.x * 2u + 2u * (blockIdx.x * 37u);  // every 2-byte alignment, crossing 128-B lines
    const U1 a = *(const U1*)(base + o);
    uint4v b = load_x4_unaligned(base + o);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b));
    out[(blockIdx.x * 64 + threadIdx.x) * 5 + 0] = a.v;
    out[(blockIdx.x * 64 + threadIdx.x) * 5 + 1] = b.x;
    out[(blockIdx.x * 64 + threadIdx.x) * 5 + 2] = b.y;
    out[(blockIdx.x * 64 + threadIdx.x) * 5 + 3] = b.z;
    out[(blockIdx.x * 64 + threadIdx.x) * 5 + 4] = b.w;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void gather16_kernel(const char* __restrict__ bricks, unsigned nbricks_mask, int iters, float* __restrict__ out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned state = gid * 2654435761u + 12345u;
    unsigned brick = (gid >> 6) * 97u;
    float acc = 0.0f;
    const float fx = 0.3f, fy = 0.6f, fz = 0.2f;
    for (int i = 0; i < iters; i++) {
        state = state * 1664525u + 1013904223u;
        const unsigned lx = (state >> 8) & 3u, ly = (state >> 12) & 3u, lz = (state >> 16) & 3u;
        if (((state >> 20) & 7u) == 0u) brick += 1u + ((state >> 24) & 3u);
        const unsigned b = brick & nbricks_mask;
        const char* p = bricks + ((size_t)b << 8) + ((lx * 25u + lz * 5u + ly) << 1);
        float y00a, y00b, y01a, y01b, y10a, y10b, y11a, y11b;
        if (VARIANT == 0) {
            const unsigned w0 = ((const U1*)p)->v, w1 = ((const U1*)(p + 10))->v, w2 = ((const U1*)(p + 50))->v, w3 = ((const U1*)(p + 60))->v;
            y00a = lo16(w0); y00b = hi16(w0); y01a = lo16(w1); y01b = hi16(w1);
            y10a = lo16(w2); y10b = hi16(w2); y11a = lo16(w3); y11b = hi16(w3);
        } else {
            uint4v u = load_x4_unaligned(p), w = load_x4_unaligned(p + 50);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(u), "+v"(w));
            y00a = lo16(u.x); y00b = hi16(u.x); y01a = hi16(u.z); y01b = lo16(u.w);
            y10a = lo16(w.x); y10b = hi16(w.x); y11a = hi16(w.z); y11b = lo16(w.w);
        }
        const float a00 = lerp1(y00a, y00b, fy), a01 = lerp1(y01a, y01b, fy), a10 = lerp1(y10a, y10b, fy), a11 = lerp1(y11a, y11b, fy);
        acc += lerp1(lerp1(a00, a01, fz), lerp1(a10, a11, fz), fx);
    }
    out[gid] = acc;
}

// Variant C: "cell records" — every cell keeps its own 8 corner values as 8 x int16 = 16 B (64 cells = 1 KB per brick, 4x
// the bytes of the int16 brick): ONE aligned global_load_dwordx4 per sample.
__global__ __launch_bounds__(256) void gather_cells16_kernel(const uint4v* __restrict__ cells, unsigned nbricks_mask, int iters, float* __restrict__ out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned state = gid * 2654435761u + 12345u;
    unsigned brick = (gid >> 6) * 97u;
    float acc = 0.0f;
    const float fx = 0.3f, fy = 0.6f, fz = 0.2f;
    for (int i = 0; i < iters; i++) {
        state = state * 1664525u + 1013904223u;
        const unsigned lx = (state >> 8) & 3u, ly = (state >> 12) & 3u, lz = (state >> 16) & 3u;
        if (((state >> 20) & 7u) == 0u) brick += 1u + ((state >> 24) & 3u);
        const unsigned b = brick & nbricks_mask;
        const uint4v u = cells[((size_t)b << 6) + (lx * 16u + lz * 4u + ly)];
        const float a00 = lerp1(lo16(u.x), hi16(u.x), fy), a01 = lerp1(lo16(u.y), hi16(u.y), fy), a10 = lerp1(lo16(u.z), hi16(u.z), fy),
                    a11 = lerp1(lo16(u.w), hi16(u.w), fy);
        acc += lerp1(lerp1(a00, a01, fz), lerp1(a10, a11, fz), fx);
    }
    out[gid] = acc;
}

__global__ __launch_bounds__(256) void gather32_kernel(const float* __restrict__ bricks, unsigned nbricks_mask, int iters, float* __restrict__ out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned state = gid * 2654435761u + 12345u;
    unsigned brick = (gid >> 6) * 97u;
    float acc = 0.0f;
    const float fx = 0.3f, fy = 0.6f, fz = 0.2f;
    for (int i = 0; i < iters; i++) {
        state = state * 1664525u + 1013904223u;
        const unsigned lx = (state >> 8) & 3u, ly = (state >> 12) & 3u, lz = (state >> 16) & 3u;
        if (((state >> 20) & 7u) == 0u) brick += 1u + ((state >> 24) & 3u);
        const unsigned b = brick & nbricks_mask;
        const float* p = bricks + ((size_t)b << 7) + (lx * 25u + lz * 5u + ly);
        const float a00 = lerp1(p[0], p[1], fy), a01 = lerp1(p[5], p[6], fy), a10 = lerp1(p[25], p[26], fy), a11 = lerp1(p[30], p[31], fy);
        acc += lerp1(lerp1(a00, a01, fz), lerp1(a10, a11, fz), fx);
    }
    out[gid] = acc;
}

// Calibration of the FETCH_SIZE counter for the int16 tap pattern: every cell of every brick is sampled exactly once
// (lane = cell, 4 x dword at +0/+10/+50/+60 B), so the unique bytes are known: nbricks * 256 (250 of them touched).
// `./gather16 calibrate` runs only this kernel (under `rocprofv3 --pmc FETCH_SIZE`) over a 512-MiB pool.
__global__ __launch_bounds__(256) void sweep16_kernel(const char* __restrict__ bricks, size_t ncells, float* __restrict__ out) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= ncells) return;
    const size_t b = gid >> 6;
    const unsigned l = (unsigned)gid & 63u, lx = l >> 4, lz = (l >> 2) & 3u, ly = l & 3u;
    const char* p = bricks + (b << 8) + ((lx * 25u + lz * 5u + ly) << 1);
    const unsigned w0 = ((const U1*)p)->v, w1 = ((const U1*)(p + 10))->v, w2 = ((const U1*)(p + 50))->v, w3 = ((const U1*)(p + 60))->v;
    out[gid & 0xfffff] = ((lo16(w0) + hi16(w0)) + (lo16(w1) + hi16(w1))) + ((lo16(w2) + hi16(w2)) + (lo16(w3) + hi16(w3)));
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "calibrate")) {
        const size_t nbricks = (size_t)1 << 21;  // 512 MiB pool of 256-B bricks
        char* bricks;
        float* out;
        if (hipMalloc(&bricks, nbricks * 256 + 64) != hipSuccess) return 1;
        (void)hipMemset(bricks, 0, nbricks * 256 + 64);
        (void)hipMalloc(&out, sizeof(float) << 20);
        const size_t ncells = nbricks * 64;
        for (int rep = 0; rep < 3; rep++)
            hipLaunchKernelGGL(sweep16_kernel, dim3((unsigned)(ncells / 256)), dim3(256), 0, 0, bricks, ncells, out);
        (void)hipDeviceSynchronize();
        printf("sweep16_kernel: %zu bricks, %zu unique bytes read per launch\n", nbricks, nbricks * 256);
        return 0;
    }
    // ---- correctness of unaligned loads
    {
        const size_t n = 1 << 16;
        std::vector<uint8_t> h(n);
        for (size_t i = 0; i < n; i++) h[i] = (uint8_t)(i * 131u + (i >> 8) * 7u + 3u);
        char* d;
        unsigned* o;
        hipMalloc(&d, n);
        hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
        const int blocks = 64;
        hipMalloc(&o, sizeof(unsigned) * blocks * 64 * 5);
        hipLaunchKernelGGL(check_kernel, dim3(blocks), dim3(64), 0, 0, d, o);
        std::vector<unsigned> r(blocks * 64 * 5);
        if (hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("unaligned check: HIP error\n"); return 1; }
        size_t bad = 0;
        for (int b = 0; b < blocks; b++)
            for (int t = 0; t < 64; t++) {
                const size_t off = (size_t)t * 2 + 2 * ((size_t)b * 37);
                unsigned want[5];
                memcpy(&want[0], &h[off], 4);
                memcpy(&want[1], &h[off], 16);
                for (int k = 0; k < 5; k++) bad += r[((size_t)b * 64 + t) * 5 + k] != want[k];
            }
        printf("unaligned global_load_dword / dwordx4 at every 2-byte alignment: %zu mismatches of %zu words\n", bad, r.size());
        if (bad) return 1;
        hipFree(d);
        hipFree(o);
    }
    const int blocks = 256 * 8 * 4, threads = 256, iters = 256;
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const struct { const char* name; unsigned nbricks; } cases[] = {
        {"32 bricks   (L1-resident)", 32u}, {"4096 bricks (L2-resident)", 4096u}, {"262144 bricks = the 256^3 pool", 262144u},
    };
    for (const auto& c : cases) {
        char* b16;
        float* b32;
        hipMalloc(&b16, (size_t)c.nbricks * 256 + 64);
        hipMemset(b16, 0, (size_t)c.nbricks * 256 + 64);
        hipMalloc(&b32, (size_t)c.nbricks * 512);
        hipMemset(b32, 0, (size_t)c.nbricks * 512);
        uint4v* c16;
        hipMalloc(&c16, (size_t)c.nbricks * 1024);
        hipMemset(c16, 0, (size_t)c.nbricks * 1024);
        for (int variant = 0; variant < 4; variant++) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL(gather16_kernel<0>, dim3(blocks), dim3(threads), 0, 0, b16, c.nbricks - 1, iters, out);
                else if (variant == 1) hipLaunchKernelGGL(gather16_kernel<1>, dim3(blocks), dim3(threads), 0, 0, b16, c.nbricks - 1, iters, out);
                else if (variant == 2) hipLaunchKernelGGL(gather32_kernel, dim3(blocks), dim3(threads), 0, 0, b32, c.nbricks - 1, iters, out);
                else hipLaunchKernelGGL(gather_cells16_kernel, dim3(blocks), dim3(threads), 0, 0, c16, c.nbricks - 1, iters, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double samples = (double)blocks * threads * iters;
            const char* vn[] = {"int16 bricks, 4 x dword (2-byte aligned)", "int16 bricks, 2 x dwordx4 (2-byte aligned)", "fp32 bricks, 4 x dwordx2 (round 1)",
                                "int16 cell records, 1 x dwordx4 (4x bytes)"};
            printf("%-32s %-44s %8.3f ms  %7.1f Gsamples/s\n", c.name, vn[variant], best, samples / best / 1e6);
        }
        hipFree(b16);
        hipFree(b32);
        hipFree(c16);
    }
    return 0;
}

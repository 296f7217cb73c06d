// Throughput of the march's inner operation in isolation: an 8-tap trilinear sample from 512-B bricks (4 x dwordx2 per
// lane, taps at +0,+20,+100,+120 bytes) plus the 14-op lerp tree, at full occupancy, for three footprints.  Every lane
// walks its own pseudo-random but INDEPENDENT sequence of cells (no dependence on the loaded values), so this is the
// rate the memory pipeline (TA / L1 / L2 / HBM) and the VALU sustain for this access pattern — the ceiling the march
// kernel's busy phase can be compared with.  Diagnostic tool, not product.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o gather gather.hip && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__device__ __forceinline__ float lerp1(float a, float b, float w) { return __builtin_fmaf(w, b - a, a); }

template <bool WITH_ALU>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ bricks, unsigned nbricks_mask, int iters, float* __restrict__ out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned state = gid * 2654435761u + 12345u;
    // neighbouring lanes start in neighbouring cells of the same brick, like the rays of an 8x8 pixel tile
    unsigned brick = (gid >> 6) * 97u;
    float acc = 0.0f;
    const float fx = 0.3f, fy = 0.6f, fz = 0.2f;
    for (int i = 0; i < iters; i++) {
        state = state * 1664525u + 1013904223u;
        const unsigned lx = (state >> 8) & 3u, ly = (state >> 12) & 3u, lz = (state >> 16) & 3u;
        if (((state >> 20) & 7u) == 0u) brick += 1u + ((state >> 24) & 3u);  // change brick every ~8 samples
        const unsigned b = brick & nbricks_mask;
        const float* p = bricks + ((size_t)b << 7) + (lx * 25u + lz * 5u + ly);
        const float y00a = p[0], y00b = p[1], y01a = p[5], y01b = p[6], y10a = p[25], y10b = p[26], y11a = p[30], y11b = p[31];
        if (WITH_ALU) {
            const float a00 = lerp1(y00a, y00b, fy), a01 = lerp1(y01a, y01b, fy), a10 = lerp1(y10a, y10b, fy), a11 = lerp1(y11a, y11b, fy);
            acc += lerp1(lerp1(a00, a01, fz), lerp1(a10, a11, fz), fx);
        } else {
            acc += ((y00a + y00b) + (y01a + y01b)) + ((y10a + y10b) + (y11a + y11b));
        }
    }
    out[gid] = acc;
}

// The same sample with the lanes of a wave COHERENT, like the 64 rays of an 8x8-pixel tile near a surface: one wave-uniform pseudo-random
// base cell per iteration, lane (i, j) of the 8x8 tile samples the cell 3i/8, 3j/8 cells further along x and z (the tile spans 3x3 cells;
// cells beyond the brick's edge fall into the next brick).  Lanes that share a cell share its lines: the L1 looks up fewer lines per
// instruction than for independent cells — the ceiling a march with coherent lanes runs against.
__global__ __launch_bounds__(256) void gather_tile_kernel(const float* __restrict__ bricks, unsigned nbricks_mask, int iters, float* __restrict__ out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    unsigned state = (gid >> 6) * 2654435761u + 12345u;  // wave-uniform sequence
    unsigned brick = (gid >> 6) * 97u;
    const unsigned ox = ((lane & 7u) * 3u) >> 3, oz = ((lane >> 3) * 3u) >> 3;
    float acc = 0.0f;
    const float fx = 0.3f, fy = 0.6f, fz = 0.2f;
    for (int i = 0; i < iters; i++) {
        state = state * 1664525u + 1013904223u;
        const unsigned bx = (state >> 8) & 3u, ly = (state >> 12) & 3u, bz = (state >> 16) & 3u;
        if (((state >> 20) & 7u) == 0u) brick += 1u + ((state >> 24) & 3u);
        const unsigned cx = bx + ox, cz = bz + oz;
        const unsigned b = (brick + (cx >> 2) * 17u + (cz >> 2) * 5u) & nbricks_mask;  // a neighbouring brick past the edge
        const float* p = bricks + ((size_t)b << 7) + ((cx & 3u) * 25u + (cz & 3u) * 5u + ly);
        const float y00a = p[0], y00b = p[1], y01a = p[5], y01b = p[6], y10a = p[25], y10b = p[26], y11a = p[30], y11b = p[31];
        const float a00 = lerp1(y00a, y00b, fy), a01 = lerp1(y01a, y01b, fy), a10 = lerp1(y10a, y10b, fy), a11 = lerp1(y11a, y11b, fy);
        acc += lerp1(lerp1(a00, a01, fz), lerp1(a10, a11, fz), fx);
    }
    out[gid] = acc;
}

// The same sample from a "cell record" layout: every cell stores its own 8 corner values contiguously (32 B, 8x the
// memory of the shared-corner bricks) and is read with 2 x dwordx4.  Half the vector-memory instructions per sample.
template <bool WITH_ALU>
__global__ __launch_bounds__(256) void gather_cells_kernel(const float4* __restrict__ cells, unsigned nbricks_mask, int iters, float* __restrict__ out) {
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
        const float4* p = cells + (((size_t)b << 6) + (lx * 16u + lz * 4u + ly)) * 2;  // 64 cells x 2 float4 per brick
        const float4 u = p[0], w = p[1];
        if (WITH_ALU) {
            const float a00 = lerp1(u.x, u.y, fy), a01 = lerp1(u.z, u.w, fy), a10 = lerp1(w.x, w.y, fy), a11 = lerp1(w.z, w.w, fy);
            acc += lerp1(lerp1(a00, a01, fz), lerp1(a10, a11, fz), fx);
        } else {
            acc += ((u.x + u.y) + (u.z + u.w)) + ((w.x + w.y) + (w.z + w.w));
        }
    }
    out[gid] = acc;
}

// Calibration of the FETCH_SIZE counter for THIS access pattern: every cell of every brick is sampled exactly once (lane
// = cell, 4 x dwordx2 at +0/+20/+100/+120 B), so the unique bytes are known: nbricks * 512.  `./gather calibrate` runs
// only this kernel (under `rocprofv3 --pmc FETCH_SIZE`), over a pool larger than the 256-MiB Infinity Cache.
__global__ __launch_bounds__(256) void sweep_kernel(const float* __restrict__ bricks, size_t ncells, float* __restrict__ out) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= ncells) return;
    const size_t b = gid >> 6;
    const unsigned l = (unsigned)gid & 63u, lx = l >> 4, lz = (l >> 2) & 3u, ly = l & 3u;
    const float* p = bricks + (b << 7) + (lx * 25u + lz * 5u + ly);
    out[gid & 0xfffff] = ((p[0] + p[1]) + (p[5] + p[6])) + ((p[25] + p[26]) + (p[30] + p[31]));
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "calibrate")) {
        const size_t nbricks = (size_t)1 << 20;  // 512 MiB pool
        float *bricks, *out;
        hipMalloc(&bricks, nbricks * 512);
        hipMemset(bricks, 0, nbricks * 512);
        hipMalloc(&out, sizeof(float) << 20);
        const size_t ncells = nbricks * 64;
        for (int rep = 0; rep < 3; rep++)
            hipLaunchKernelGGL(sweep_kernel, dim3((unsigned)(ncells / 256)), dim3(256), 0, 0, bricks, ncells, out);
        hipDeviceSynchronize();
        printf("sweep_kernel: %zu bricks, %zu unique bytes read per launch\n", nbricks, nbricks * 512);
        return 0;
    }
    const int blocks = 256 * 8 * 4, threads = 256, iters = 256;  // 8192 workgroups: every CU at its occupancy limit
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const struct { const char* name; unsigned nbricks; } cases[] = {
        {"L1-resident  (16 KB per CU-ish: 32 bricks)", 32u},
        {"L2-resident  (2 MB: 4096 bricks)", 4096u},
        {"MALL/HBM     (128 MB: 262144 bricks = the 256^3 pool)", 262144u},
    };
    for (const auto& c : cases) {
        float* bricks;
        hipMalloc(&bricks, (size_t)c.nbricks * 512);
        hipMemset(bricks, 0, (size_t)c.nbricks * 512);
        for (int alu = 0; alu < 2; alu++) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                if (alu) hipLaunchKernelGGL(gather_kernel<true>, dim3(blocks), dim3(threads), 0, 0, bricks, c.nbricks - 1, iters, out);
                else hipLaunchKernelGGL(gather_kernel<false>, dim3(blocks), dim3(threads), 0, 0, bricks, c.nbricks - 1, iters, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double samples = (double)blocks * threads * iters;
            printf("%-58s %-9s %8.3f ms  %7.1f Gsamples/s  %6.2f TB/s (32 B/sample)\n", c.name, alu ? "taps+lerp" : "taps only", best,
                   samples / best / 1e6, samples * 32 / best / 1e9);
        }
        {   // coherent lanes (an 8x8 tile over 3x3 cells)
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(gather_tile_kernel, dim3(blocks), dim3(threads), 0, 0, bricks, c.nbricks - 1, iters, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double samples = (double)blocks * threads * iters;
            printf("  the same with COHERENT lanes (8x8 tile over 3x3 cells)    %-9s %8.3f ms  %7.1f Gsamples/s  %6.2f TB/s (32 B/sample)\n", "taps+lerp",
                   best, samples / best / 1e6, samples * 32 / best / 1e9);
        }
        hipFree(bricks);
        // cell records: same number of bricks, 2 KB each
        float4* cells;
        hipMalloc(&cells, (size_t)c.nbricks * 2048);
        hipMemset(cells, 0, (size_t)c.nbricks * 2048);
        {
            float best = 1e30f;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(gather_cells_kernel<true>, dim3(blocks), dim3(threads), 0, 0, cells, c.nbricks - 1, iters, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double samples = (double)blocks * threads * iters;
            printf("  same bricks as 32-B cell records (2 x dwordx4, 4x bytes)   %-9s %8.3f ms  %7.1f Gsamples/s  %6.2f TB/s (32 B/sample)\n", "taps+lerp",
                   best, samples / best / 1e6, samples * 32 / best / 1e9);
        }
        hipFree(cells);
    }
    return 0;
}

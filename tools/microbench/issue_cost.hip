// What does ONE vector (or scalar) instruction of a given kind cost a wave on MI355X — as a chain of dependent instructions and as
// independent ones, with one wave per SIMD and with the march kernel's eight?  The march loop is latency-limited at 8 waves per SIMD
// (DESIGN.md section 4): what its time is made of is the instructions on a wave's in-order path, by KIND — round 5 found the packed
// fp32 forms (v_pk_fma_f32 / v_pk_add_f32) slower than the two scalar instructions they replace; this table is where such a thing
// shows without building the kernel twice.
//   hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost
// Every kernel runs REPS x 64 instructions of one kind between two s_memtime reads; printed: cycles per instruction as one wave sees
// it (dependent chain / four independent chains), alone on its SIMD and with seven neighbours running the same stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int REPS = 200;
typedef float float2v __attribute__((ext_vector_type(2)));

#define KERNEL(NAME, DEP, IND)                                                                                          \
    __global__ __launch_bounds__(64) void k_##NAME##_dep(unsigned long long* out, float seed) {                         \
        float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0f, d = 2.0f;                                              \
        int ia = (int)threadIdx.x, ib = 3;                                                                              \
        float2v p = {a, b}, q = {c, d};                                                                                 \
        unsigned long long t0, t1, m = 0;                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));                                                \
        for (int r = 0; r < REPS; r++)                                                                                  \
            asm volatile(".rept 64\n\t" DEP "\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+s"(m), "+v"(p), "+v"(q) : : "vcc", "scc");  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                                                \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                \
        if (a + b + c + d + ia + ib + (float)m + p.x + p.y + q.x + q.y == 12345.678f) out[0] = 0;                                               \
    }                                                                                                                   \
    __global__ __launch_bounds__(64) void k_##NAME##_ind(unsigned long long* out, float seed) {                         \
        float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.0f, d = 2.0f;                                              \
        int ia = (int)threadIdx.x, ib = 3;                                                                              \
        float2v p = {a, b}, q = {c, d};                                                                                 \
        unsigned long long t0, t1, m = 0;                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));                                                \
        for (int r = 0; r < REPS; r++)                                                                                  \
            asm volatile(".rept 16\n\t" IND "\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+s"(m), "+v"(p), "+v"(q) : : "vcc", "scc");  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                                                \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                \
        if (a + b + c + d + ia + ib + (float)m + p.x + p.y + q.x + q.y == 12345.678f) out[0] = 0;                                               \
    }

// operands: %0..%3 floats a b c d, %4 %5 ints, %6 a scalar pair, %7 %8 register pairs (two floats each)
KERNEL(fma, "v_fma_f32 %0, %0, %1, %0", "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3")
KERNEL(add, "v_add_f32 %0, %0, %1", "v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3")
KERNEL(floor, "v_floor_f32 %0, %0", "v_floor_f32 %0, %0\n\tv_floor_f32 %1, %1\n\tv_floor_f32 %2, %2\n\tv_floor_f32 %3, %3")
KERNEL(med3, "v_med3_f32 %0, %0, %1, 0", "v_med3_f32 %0, %0, %0, 0\n\tv_med3_f32 %1, %1, %1, 0\n\tv_med3_f32 %2, %2, %2, 0\n\tv_med3_f32 %3, %3, %3, 0")
KERNEL(cvt_i32, "v_cvt_i32_f32 %4, %0\n\tv_cvt_f32_i32 %0, %4", "v_cvt_i32_f32 %4, %0\n\tv_cvt_f32_i32 %1, %5\n\tv_cvt_i32_f32 %5, %2\n\tv_cvt_f32_i32 %3, %4")
KERNEL(cvt_flr, "v_cvt_flr_i32_f32 %4, %0\n\tv_cvt_f32_i32 %0, %4", "v_cvt_flr_i32_f32 %4, %0\n\tv_cvt_f32_i32 %1, %5\n\tv_cvt_flr_i32_f32 %5, %2\n\tv_cvt_f32_i32 %3, %4")
KERNEL(sdwa, "v_cvt_f32_i32_sdwa %0, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_cvt_i32_f32 %4, %0",
       "v_cvt_f32_i32_sdwa %0, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\tv_cvt_f32_i32_sdwa %1, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
       "v_cvt_f32_i32_sdwa %2, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\tv_cvt_f32_i32_sdwa %3, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
KERNEL(bfe_cvt, "v_bfe_i32 %5, %4, 0, 16\n\tv_cvt_f32_i32 %0, %5", "v_bfe_i32 %5, %4, 0, 16\n\tv_cvt_f32_i32 %0, %5\n\tv_ashrrev_i32 %5, 16, %4\n\tv_cvt_f32_i32 %1, %5")
KERNEL(mad24, "v_mad_u32_u24 %4, %4, %5, %4", "v_mad_u32_u24 %4, %4, %4, %4\n\tv_mad_u32_u24 %5, %5, %5, %5\n\tv_mad_u32_u24 %4, %4, %4, %4\n\tv_mad_u32_u24 %5, %5, %5, %5")
KERNEL(bfe, "v_bfe_u32 %4, %4, %5, 4", "v_bfe_u32 %4, %4, %5, 4\n\tv_bfe_u32 %5, %5, %4, 4\n\tv_and_b32 %4, 3, %4\n\tv_lshlrev_b32 %5, 2, %5")
KERNEL(add3, "v_add3_u32 %4, %4, %5, %4", "v_add3_u32 %4, %4, %4, %4\n\tv_or3_b32 %5, %5, %5, %5\n\tv_lshl_add_u32 %4, %4, 2, %4\n\tv_add3_u32 %5, %5, %5, %5")
KERNEL(cndmask, "v_cndmask_b32_e64 %0, %0, %1, %6", "v_cndmask_b32_e64 %0, %0, %0, %6\n\tv_cndmask_b32_e64 %1, %1, %1, %6\n\tv_cndmask_b32_e64 %2, %2, %2, %6\n\tv_cndmask_b32_e64 %3, %3, %3, %6")
KERNEL(cmp_cnd, "v_cmp_lt_f32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc", "v_cmp_lt_f32_e32 vcc, %0, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3\n\tv_cndmask_b32_e32 %0, %0, %1, vcc")
KERNEL(cmp_salu, "v_cmp_lt_f32_e64 %6, %0, %1\n\ts_and_b64 vcc, %6, exec\n\tv_cndmask_b32_e32 %0, %0, %1, vcc", "v_cmp_lt_f32_e64 %6, %0, %1\n\ts_and_b64 vcc, %6, exec\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_add_f32 %2, %2, %2")
KERNEL(pk_fma, "v_pk_fma_f32 %7, %7, %8, %7", "v_pk_fma_f32 %7, %7, %7, %7\n\tv_pk_fma_f32 %8, %8, %8, %8\n\tv_pk_fma_f32 %7, %7, %7, %7\n\tv_pk_fma_f32 %8, %8, %8, %8")
KERNEL(pk_add, "v_pk_add_f32 %7, %7, %8", "v_pk_add_f32 %7, %7, %7\n\tv_pk_add_f32 %8, %8, %8\n\tv_pk_add_f32 %7, %7, %7\n\tv_pk_add_f32 %8, %8, %8")
KERNEL(salu, "s_and_b64 %6, %6, exec", "s_and_b64 %6, %6, exec\n\ts_or_b64 vcc, vcc, exec\n\ts_and_b64 %6, %6, exec\n\ts_or_b64 vcc, vcc, exec")
#undef KERNEL

typedef void (*kernel_t)(unsigned long long*, float);

static void run(const char* name, kernel_t dep, kernel_t ind, int per_rept_dep, int per_rept_ind, unsigned long long* d) {
    double res[4];
    int k = 0;
    printf("%-34s", name);
    for (int waves_per_simd : {1, 8}) {
        const int blocks = 256 * 4 * waves_per_simd;
        for (kernel_t f : {dep, ind}) {
            std::vector<unsigned long long> h(blocks);
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(f, dim3(blocks), dim3(64), 0, 0, d, 1.5f);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), d, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double sum = 0;
            for (auto v : h) sum += (double)v;
            const int n = (f == dep) ? 64 * per_rept_dep : 16 * per_rept_ind;
            /* s_memtime counts at 100 MHz on this chip: convert with the shader clock the run is at (2.4 GHz nominal) */
            res[k++] = sum / blocks / ((double)REPS * n);
        }
    }
    printf("  alone: dependent %7.3f  independent %7.3f    8 waves/SIMD: dependent %7.3f  independent %7.3f   (s_memtime ticks per instruction)\n", res[0], res[1], res[2],
           res[3]);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long* d;
    hipMalloc(&d, 256 * 4 * 8 * sizeof(unsigned long long));
#define RUN(NAME, LABEL, ND, NI) run(LABEL, k_##NAME##_dep, k_##NAME##_ind, ND, NI, d)
    RUN(fma, "v_fma_f32", 1, 4);
    RUN(add, "v_add_f32", 1, 4);
    RUN(pk_fma, "v_pk_fma_f32 (on register pairs)", 1, 4);
    RUN(pk_add, "v_pk_add_f32", 1, 4);
    RUN(floor, "v_floor_f32", 1, 4);
    RUN(med3, "v_med3_f32", 1, 4);
    RUN(cvt_i32, "v_cvt_i32_f32 + v_cvt_f32_i32", 2, 4);
    RUN(cvt_flr, "v_cvt_flr_i32_f32 + v_cvt_f32_i32", 2, 4);
    RUN(sdwa, "v_cvt_f32_i32_sdwa (+cvt back)", 2, 4);
    RUN(bfe_cvt, "v_bfe_i32 + v_cvt_f32_i32", 2, 4);
    RUN(mad24, "v_mad_u32_u24", 1, 4);
    RUN(bfe, "v_bfe_u32 / and / lshl", 1, 4);
    RUN(add3, "v_add3 / or3 / lshl_add", 1, 4);
    RUN(cndmask, "v_cndmask_b32_e64 (SGPR mask)", 1, 4);
    RUN(cmp_cnd, "v_cmp -> vcc -> v_cndmask", 3, 4);
    RUN(cmp_salu, "v_cmp -> s_and -> v_cndmask", 3, 4);
    RUN(salu, "s_and_b64 / s_or_b64", 1, 4);
    return 0;
}

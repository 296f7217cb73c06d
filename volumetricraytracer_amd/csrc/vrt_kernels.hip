/*
 * vrt_kernels.hip — gfx950 (CDNA4) kernels of the volumetric SDF ray-marcher.
 *
 * One wavefront lane per primary ray, one 64-lane wave per 8x8 pixel tile, one wave per workgroup (four consecutive
 * workgroups of an XCD cover a 16x16-pixel tile); the grid's y axis is the FRAME: one launch marches a block of frames, each
 * with its own camera from the kernarg segment.  The march is a sphere-trace over the trilinear interpolant of
 * the density grid; taps come either from the dense grid or from 4^3-cell bricks (5^3 samples:
 * 512 B of fp32 = four 128-B lines, or 256 B of int16 — the reference's own volume texel — = two)
 * so that the 8 taps of a sample share one brick.  A two-level empty-space table (brick bytes +
 * sub-block nibbles) lets the ray cross cells without surface without touching the bricks.
 * Normal, shadow test, shading, cube-map lookup and tone-map stay in registers.
 * Latency-bound gather work: no MFMA.
 *
 * The arithmetic contract (operation order, where FMA is used) is DESIGN.md §3; it is
 * restated independently by oracle/vrt_oracle.cpp.  This file is compiled with
 * -ffp-contract=off; every fused multiply-add is an explicit __builtin_fmaf.
 *
 * Replaces (paths relative to /root/reference/VolumetricRaytracer/VolumetricRaytracer/
 * Renderer/DX/Resources/Shaders/): VRRaygen Raytracing.hlsl:26-39, VRIntersection :147-336,
 * VRIntersectionShadowRay :338-442, VRClosestHit (NoTex) Raytracing_NoTex.hlsl:41-139,
 * VRMiss :444-449, helpers Include/Ray.hlsli, Include/Voxel.hlsli, Include/Lighting.hlsli.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "vrt_device.h"
#include "vrt_launch.h"
#include "voxelize_core.h"

namespace vrt {

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ F3 normalize3(F3 a) {
    float inv = 1.0f / sqrtf(dot3(a, a));
    return a * inv;
}
__device__ __forceinline__ float maxf_(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float minf_(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ F3 mul33(const float* m, F3 v) {
    return f3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[3] * v.x + m[4] * v.y) + m[5] * v.z,
              (m[6] * v.x + m[7] * v.y) + m[8] * v.z);
}
__device__ __forceinline__ float lerp1(float a, float b, float w) { return __builtin_fmaf(w, b - a, a); }

/* 1-ulp hardware approximations (v_rcp_f32 / v_rsq_f32 / v_log_f32 / v_exp_f32).  Used ONLY downstream
 * of the hit decision — normal length, BRDF, tone-map — where a last-bit difference moves a colour
 * channel by ~1e-7.  Everything that feeds a discrete decision (ray setup, slab tests, the march,
 * texel selection) uses correctly rounded division / sqrt and matches the CPU oracle bit for bit. */
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_pow(float x, float e) { return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * e); }

/* Volume as the march sees it (all in registers; wave-uniform for single-instance scenes). */
/* Global-address-space view of a pointer that was itself loaded from memory (keeps the taps
 * on global_load instead of flat_load). */
typedef const float __attribute__((address_space(1))) * gfloat_p;
typedef const unsigned __attribute__((address_space(1))) * guint_p;

typedef const uint8_t __attribute__((address_space(1))) * gbyte_p;
typedef const char __attribute__((address_space(1))) * gchar_p;

struct VolRef {
    gchar_p p;      /* dense grid or brick pool, by PATH */
    gbyte_p skip;   /* empty-space table, level 1 (brick leap counts), or null */
    guint_p nib;    /* empty-space table, level 2 (sub-block nibbles); valid with skip */
    gbyte_p cube;   /* Cube modes: distance-to-solid table */
    int N;
    int nb;
    float extent, inv_cell, cell, dscale, step_max;
    float alo[3], ahi[3]; /* active box (object space) of volumes with an empty-space table: the bounding box of the near bricks */
};

/* Where the taps of an internal path come from. */
template <int PATH>
__device__ __host__ constexpr int data_path() {
    return PATH == VRT_PATH_DENSE ? VRT_PATH_DENSE : PATH == kPathCells16 ? kPathCells16 : (PATH == kPathBrick16 || PATH == kPathCube16) ? kPathBrick16 : VRT_PATH_BRICK;
}

template <int DP>
__device__ __forceinline__ VolRef load_vol(const DVolume* __restrict__ v) {
    VolRef r;
    r.p = (gchar_p)((DP == VRT_PATH_DENSE) ? (const void*)v->dense : (DP == kPathCells16) ? v->cells : v->bricks);
    r.skip = (gbyte_p)v->skip;
    r.nib = (guint_p)v->nib;
    r.cube = (gbyte_p)v->cube_skip;
    r.N = v->N;
    r.nb = v->nb;
    r.extent = v->extent;
    r.inv_cell = v->inv_cell;
    r.cell = v->cell;
    r.dscale = v->density_scale;
    r.step_max = v->step_max;
    for (int a = 0; a < 3; a++) {
        r.alo[a] = v->abox_lo[a];
        r.ahi[a] = v->abox_hi[a];
    }
    return r;
}

/* The 8 corner taps of one cell, in the order the lerp tree consumes them. */
struct Taps {
    float y00a, y00b, y01a, y01b, y10a, y10b, y11a, y11b; /* (x,z) = 00,01,10,11; a = y, b = y+1 */
};

/* a*b + c with 24-bit unsigned operands (full-rate v_mad_u32_u24; v_mul_lo_u32 is quarter rate). */
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

/* The table word of a brick: base in scalar registers + a 32-bit byte offset (the tables of a 1024^3 volume are 64 MiB), so that the load needs
 * one shift, not a 64-bit address. */
__device__ __forceinline__ unsigned table_word(const VolRef& V, unsigned brick) {
    return *(guint_p)((gchar_p)V.nib + (brick << 2));
}

/* Index of the brick that holds cell (cx,cy,cz) — also the index into the brick tables (skip, nib, cube_skip). */
__device__ __forceinline__ unsigned brick_index(const VolRef& V, int cx, int cy, int cz) {
    const unsigned nb = (unsigned)V.nb;
    return mad24(mad24((unsigned)cx >> 2, nb, (unsigned)cz >> 2), nb, (unsigned)cy >> 2);
}

/* Sample index of cell (cx,cy,cz)'s origin corner inside its brick record. */
__device__ __forceinline__ unsigned brick_local(int cx, int cy, int cz) {
    return ((unsigned)cx & 3u) * 25u + ((unsigned)cz & 3u) * 5u + ((unsigned)cy & 3u);
}

/* Taps of a cell from its fp32 brick (the brick index is passed in so that the march computes it once for the
 * taps and the empty-space tables): 4 x dwordx2, one per y-pair. */
__device__ __forceinline__ Taps fetch8_brick(const VolRef& V, unsigned brick, int cx, int cy, int cz) {
    const unsigned off = ((brick << 7) + brick_local(cx, cy, cz)) << 2; /* bytes */
    const gfloat_p b = (gfloat_p)(V.p + off);
    Taps t;
    t.y00a = b[0];
    t.y00b = b[1];
    t.y01a = b[5];
    t.y01b = b[6];
    t.y10a = b[25];
    t.y10b = b[26];
    t.y11a = b[30];
    t.y11b = b[31];
    return t;
}

/* Taps of a cell from its int16 brick (VRT_FORMAT_TEXEL16): 4 x dword at 2-byte alignment, one per y-pair (gfx950 runs
 * with unaligned global access enabled), each unpacked with two v_cvt_f32_i32_sdwa (sign-extended word select). */
struct __attribute__((packed, aligned(2))) Pair16 {
    unsigned v;
};
typedef const Pair16 __attribute__((address_space(1))) * gpair16_p;
__device__ __forceinline__ float lo16f(unsigned w) { return (float)(short)(w & 0xffffu); }
__device__ __forceinline__ float hi16f(unsigned w) { return (float)(short)(w >> 16); }
__device__ __forceinline__ Taps fetch8_brick16(const VolRef& V, unsigned brick, int cx, int cy, int cz) {
    const unsigned off = ((brick << 7) + brick_local(cx, cy, cz)) << 1; /* bytes */
    const gchar_p b = V.p + off;
    const unsigned w0 = ((gpair16_p)b)->v, w1 = ((gpair16_p)(b + 10))->v, w2 = ((gpair16_p)(b + 50))->v, w3 = ((gpair16_p)(b + 60))->v;
    Taps t;
    t.y00a = lo16f(w0);
    t.y00b = hi16f(w0);
    t.y01a = lo16f(w1);
    t.y01b = hi16f(w1);
    t.y10a = lo16f(w2);
    t.y10b = hi16f(w2);
    t.y11a = lo16f(w3);
    t.y11b = hi16f(w3);
    return t;
}

/* Taps of a cell from its 16-byte cell record (VRT_PATH_CELLS): ONE aligned dwordx4. */
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Taps fetch8_cells16(const VolRef& V, unsigned brick, int cx, int cy, int cz) {
    const unsigned rec = (brick << 6) + (((unsigned)cx & 3u) << 4) + (((unsigned)cz & 3u) << 2) + ((unsigned)cy & 3u);
    const uint4v w = *(const uint4v __attribute__((address_space(1)))*)(V.p + ((size_t)rec << 4));
    Taps t;
    t.y00a = lo16f(w.x);
    t.y00b = hi16f(w.x);
    t.y01a = lo16f(w.y);
    t.y01b = hi16f(w.y);
    t.y10a = lo16f(w.z);
    t.y10b = hi16f(w.z);
    t.y11a = lo16f(w.w);
    t.y11b = hi16f(w.w);
    return t;
}

/* Fetch the taps of cell (cx,cy,cz) on data path DP (VRT_PATH_DENSE, VRT_PATH_BRICK, kPathBrick16 or kPathCells16).  Offsets are
 * unsigned 32-bit byte offsets from the volume base (<= 4 GiB pools), built with 24-bit multiplies (cells < 2^10,
 * bricks < 2^24). */
template <int DP>
__device__ __forceinline__ Taps fetch8_at(const VolRef& V, unsigned brick, int cx, int cy, int cz) {
    if constexpr (DP == VRT_PATH_DENSE) {
        Taps t;
        const unsigned N = (unsigned)V.N;
        const unsigned off = mad24(mad24((unsigned)cx, N, (unsigned)cz), N, (unsigned)cy);
        const gfloat_p b = (gfloat_p)(V.p + ((size_t)off << 2));
        const unsigned NN = N * N;
        t.y00a = b[0];
        t.y00b = b[1];
        t.y01a = b[N];
        t.y01b = b[N + 1];
        t.y10a = b[NN];
        t.y10b = b[NN + 1];
        t.y11a = b[NN + N];
        t.y11b = b[NN + N + 1];
        return t;
    } else if constexpr (DP == kPathBrick16) {
        return fetch8_brick16(V, brick, cx, cy, cz);
    } else if constexpr (DP == kPathCells16) {
        return fetch8_cells16(V, brick, cx, cy, cz);
    } else {
        return fetch8_brick(V, brick, cx, cy, cz);
    }
}
template <int DP>
__device__ __forceinline__ Taps fetch8(const VolRef& V, int cx, int cy, int cz) {
    return fetch8_at<DP>(V, DP == VRT_PATH_DENSE ? 0u : brick_index(V, cx, cy, cz), cx, cy, cz);
}

/* Trilinear interpolant from the taps, lerp by lerp as the oracle does them (each one sub + one fma): over z, over x, then
 * over y.  Scalar instructions on purpose: rounds 2-4 ran the z- and x-lerps on both y values at once with v_pk_add_f32 / v_pk_fma_f32
 * (6 packed + 2 instead of 14), which is fewer instructions and, measured in round 5, SLOWER: with the packed forms gone from the
 * march (here and in cell_at) config 3 gains 1.1 %, a frame whose every wave marches 2.4 % (profiles/r05_ab_step_asm.txt (k), (l)). */
__device__ __forceinline__ float lerp8(const Taps& t, float fx, float fy, float fz) {
    const float a0a = lerp1(t.y00a, t.y01a, fz), a0b = lerp1(t.y00b, t.y01b, fz); /* x0: lerp over z, for y and y+1 */
    const float a1a = lerp1(t.y10a, t.y11a, fz), a1b = lerp1(t.y10b, t.y11b, fz); /* x1 */
    return lerp1(lerp1(a0a, a1a, fx), lerp1(a0b, a1b, fx), fy);                   /* over x, then over y */
}

template <int DP>
__device__ __forceinline__ float trilinear(const VolRef& V, int cx, int cy, int cz, float fx, float fy, float fz) {
    return lerp8(fetch8<DP>(V, cx, cy, cz), fx, fy, fz);
}

/* Ray / box [-e,e]^3 slab test with inf-safe reciprocals (Ray.hlsli:111-134). */
__device__ __forceinline__ bool slab(F3 o, F3 d, float e, float t_cur, float& t_enter, float& t_exit) {
    const float inf = __builtin_inff();
    bool px = d.x > 0.0f, py = d.y > 0.0f, pz = d.z > 0.0f;
    float ix = d.x != 0.0f ? 1.0f / d.x : (px ? inf : -inf);
    float iy = d.y != 0.0f ? 1.0f / d.y : (py ? inf : -inf);
    float iz = d.z != 0.0f ? 1.0f / d.z : (pz ? inf : -inf);
    float tminx = ((px ? -e : e) - o.x) * ix, tmaxx = ((px ? e : -e) - o.x) * ix;
    float tminy = ((py ? -e : e) - o.y) * iy, tmaxy = ((py ? e : -e) - o.y) * iy;
    float tminz = ((pz ? -e : e) - o.z) * iz, tmaxz = ((pz ? e : -e) - o.z) * iz;
    t_enter = maxf_(maxf_(tminx, tminy), tminz);
    t_exit = minf_(minf_(tmaxx, tmaxy), tmaxz);
    return t_exit > t_enter && t_exit >= 0.0f && t_enter <= t_cur;
}

/* Interval of the ray inside the box [lo,hi] with the reciprocals given: the volume's ACTIVE box — the bounding box of
 * the bricks that can hold surface.  Outside it the empty-space table would only leap; clipping the march to it saves
 * those iterations (most rays of a frame never enter it).  Oracle: same operations in the same order. */
__device__ __forceinline__ void slab_interval(F3 o, F3 d, F3 inv, const float* lo, const float* hi, float& t_enter, float& t_exit) {
    const bool px = d.x > 0.0f, py = d.y > 0.0f, pz = d.z > 0.0f;
    const float tminx = ((px ? lo[0] : hi[0]) - o.x) * inv.x, tmaxx = ((px ? hi[0] : lo[0]) - o.x) * inv.x;
    const float tminy = ((py ? lo[1] : hi[1]) - o.y) * inv.y, tmaxy = ((py ? hi[1] : lo[1]) - o.y) * inv.y;
    const float tminz = ((pz ? lo[2] : hi[2]) - o.z) * inv.z, tmaxz = ((pz ? hi[2] : lo[2]) - o.z) * inv.z;
    t_enter = maxf_(maxf_(tminx, tminy), tminz);
    t_exit = minf_(minf_(tmaxx, tmaxy), tmaxz);
}

/* Inf-safe reciprocals of a direction (Ray.hlsli:111-134). */
__device__ __forceinline__ F3 safe_rcp3(F3 d) {
    const float inf = __builtin_inff();
    return f3(d.x != 0.0f ? 1.0f / d.x : (d.x > 0.0f ? inf : -inf), d.y != 0.0f ? 1.0f / d.y : (d.y > 0.0f ? inf : -inf),
              d.z != 0.0f ? 1.0f / d.z : (d.z > 0.0f ? inf : -inf));
}

/* The same test with the reciprocals given (they only depend on the direction). */
__device__ __forceinline__ bool slab_inv(F3 o, F3 d, F3 inv, float e, float t_cur, float& t_enter, float& t_exit) {
    const bool px = d.x > 0.0f, py = d.y > 0.0f, pz = d.z > 0.0f;
    const float tminx = ((px ? -e : e) - o.x) * inv.x, tmaxx = ((px ? e : -e) - o.x) * inv.x;
    const float tminy = ((py ? -e : e) - o.y) * inv.y, tmaxy = ((py ? e : -e) - o.y) * inv.y;
    const float tminz = ((pz ? -e : e) - o.z) * inv.z, tmaxz = ((pz ? e : -e) - o.z) * inv.z;
    t_enter = maxf_(maxf_(tminx, tminy), tminz);
    t_exit = minf_(minf_(tmaxx, tmaxy), tmaxz);
    return t_exit > t_enter && t_exit >= 0.0f && t_enter <= t_cur;
}

/* General box slab for the BVH (world space). */
__device__ __forceinline__ bool slab_box(F3 o, F3 d, const DBvhNode& n, float t_cur) {
    const float inf = __builtin_inff();
    bool px = d.x > 0.0f, py = d.y > 0.0f, pz = d.z > 0.0f;
    float ix = d.x != 0.0f ? 1.0f / d.x : (px ? inf : -inf);
    float iy = d.y != 0.0f ? 1.0f / d.y : (py ? inf : -inf);
    float iz = d.z != 0.0f ? 1.0f / d.z : (pz ? inf : -inf);
    float tminx = ((px ? n.lo[0] : n.hi[0]) - o.x) * ix, tmaxx = ((px ? n.hi[0] : n.lo[0]) - o.x) * ix;
    float tminy = ((py ? n.lo[1] : n.hi[1]) - o.y) * iy, tmaxy = ((py ? n.hi[1] : n.lo[1]) - o.y) * iy;
    float tminz = ((pz ? n.lo[2] : n.hi[2]) - o.z) * iz, tmaxz = ((pz ? n.hi[2] : n.lo[2]) - o.z) * iz;
    float te = maxf_(maxf_(tminx, tminy), tminz);
    float tx = minf_(minf_(tmaxx, tmaxy), tmaxz);
    /* conservative: a NaN (origin on a face plane, zero direction) must not cull */
    return !(tx < te) && !(tx < 0.0f) && !(te > t_cur);
}

/* A lane's hit counter (at most 3: one per level of the closest-hit loop) also carries, above bit kExhaustedShift, how many of the
 * lane's marches ran out of budget (max_steps positions visited with the ray still inside the volume): at most 3 levels x 11 rays
 * x 64 instances of them, in the 24 bits above.  The sample counters count samples only: 65535 positions x 33 rays x 64 instances
 * stay below 2^32 (round 2 packed this count into the sample counters' upper 12 bits, which long budgets over many instances
 * could carry into). */
constexpr unsigned kExhaustedShift = 8;
constexpr unsigned kExhaustedOne = 1u << kExhaustedShift;

/* Diagnostic-build accumulators (wave-uniform, shader-clock cycles); unused otherwise. */
struct DiagAcc {
    unsigned long long mem = 0;   /* address ready → interpolated value available (loads + lerps) */
    unsigned long long loop = 0;  /* whole march-loop iterations */
    unsigned iters = 0;           /* march-loop iterations of this lane */
    unsigned fetches = 0;         /* iterations whose taps were back within kDiagFastFetch cycles (every lane hit a cache) */
};
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

/* One ray against one instance, in the instance's object / voxel space. */
struct RaySeg {
    F3 oo, od;            /* object-space origin and (un-normalised) direction */
    F3 uo, ud;            /* the same ray in voxel units: u(t) = uo + ud*t */
    float t_enter, t0, t_end;
    float ds;             /* density → ray-parameter distance: density_scale / |od| */
    float smax;           /* step_max / |od| */
    float smax_relax;     /* the longest stretched step: smax * max(k_relax, 1) */
    float t_skip_end;     /* up to this t the hit threshold is at most smax / 2: a position without an active cell in reach may be skipped */
    float cmax;           /* N - 2 */
    float base_min;       /* step_min + cone_eps * t_base */
    float leap_unit;      /* one brick edge (4 cells) in ray-parameter units */
    float cell_unit;      /* one cell edge */
    bool clipped;         /* the march starts at the active box, not at the volume's own face */
};

/* Clip the march interval [t0, t_end] of a ray set up against the volume box to the volume's active box.  False: the ray
 * misses the active box (nothing to march).  Only while the whole interval lies where an inactive cell cannot produce a hit
 * (t <= t_skip_end: the hit threshold stays below half the step clamp); a volume so far away that a pixel's footprint exceeds
 * that is marched over its whole box, like the oracle does. */
__device__ __forceinline__ bool clip_to_active_box(const VolRef& V, F3 inv, RaySeg& R) {
    R.clipped = false;
    if (V.skip == nullptr || !(R.t_end <= R.t_skip_end)) return true;
    float ta, tb;
    slab_interval(R.oo, R.od, inv, V.alo, V.ahi, ta, tb);
    if (!(tb > ta) || !(tb >= 0.0f)) return false;
    if (ta > R.t0) {
        R.t0 = ta;
        R.clipped = true;
    }
    R.t_end = minf_(R.t_end, tb);
    return true;
}

/* Transform the ray into the instance, slab-test its volume box and derive the march constants.
 * Returns false when the box is missed (nothing else is then valid). */
template <bool CLIP = true>
__device__ __forceinline__ bool setup_ray(const DFrame& F, const DInstance* __restrict__ I, const VolRef& V, F3 o, F3 d,
                                          float t_cur, float t_base, RaySeg& R) {
    F3 rel = f3(o.x - I->pos[0], o.y - I->pos[1], o.z - I->pos[2]);
    R.oo = mul33(I->w2o, rel);
    R.od = mul33(I->w2o, d);
    float t_exit;
    const F3 inv = safe_rcp3(R.od);
    if (!slab_inv(R.oo, R.od, inv, V.extent, t_cur, R.t_enter, t_exit)) return false;
    const float inv_len = 1.0f / sqrtf(dot3(R.od, R.od));
    R.ds = V.dscale * inv_len;
    R.smax = V.step_max * inv_len;
    R.smax_relax = F.k_relax > 1.0f ? R.smax * F.k_relax : R.smax;
    R.t_skip_end = F.cone_eps > 0.0f ? (0.5f * R.smax - F.eps_hit) / F.cone_eps
                                     : (F.eps_hit + F.eps_hit <= R.smax ? __builtin_inff() : -__builtin_inff());
    R.uo = f3((R.oo.x + V.extent) * V.inv_cell, (R.oo.y + V.extent) * V.inv_cell, (R.oo.z + V.extent) * V.inv_cell);
    R.ud = R.od * V.inv_cell;
    R.cmax = (float)(V.N - 2);
    R.t0 = (R.t_enter > 0.0f ? R.t_enter : 0.0f) + F.eps_in;
    R.t_end = minf_(t_exit, t_cur);
    /* smallest step: one pixel-footprint radius at the total path length t_base + t */
    R.base_min = __builtin_fmaf(t_base, F.cone_eps, F.step_min);
    R.leap_unit = (4.0f * V.cell) * inv_len;
    R.cell_unit = R.leap_unit * 0.25f;
    R.clipped = false;
    if constexpr (CLIP) return clip_to_active_box(V, inv, R); /* (the Cube modes march the whole volume box) */
    return true;
}

/* setup_ray for the directional light's shadow ray: direction-only terms come precomputed with the instance
 * (DInstance::sh_*, same arithmetic on the host), so only the origin is transformed here.  Bit-identical to
 * setup_ray(F, I, V, o, light_dir, ...). */
__device__ __forceinline__ bool setup_shadow_ray(const DFrame& F, const DInstance* __restrict__ I, const VolRef& V, F3 o, float t_cur,
                                                 float t_base, RaySeg& R) {
    F3 rel = f3(o.x - I->pos[0], o.y - I->pos[1], o.z - I->pos[2]);
    R.oo = mul33(I->w2o, rel);
    R.od = f3(I->sh_od[0], I->sh_od[1], I->sh_od[2]);
    float t_exit;
    if (!slab_inv(R.oo, R.od, f3(I->sh_inv[0], I->sh_inv[1], I->sh_inv[2]), V.extent, t_cur, R.t_enter, t_exit)) return false;
    const float inv_len = I->sh_inv_len;
    R.ds = V.dscale * inv_len;
    R.smax = V.step_max * inv_len;
    R.smax_relax = F.k_relax > 1.0f ? R.smax * F.k_relax : R.smax;
    R.t_skip_end = F.cone_eps > 0.0f ? (0.5f * R.smax - F.eps_hit) / F.cone_eps
                                     : (F.eps_hit + F.eps_hit <= R.smax ? __builtin_inff() : -__builtin_inff());
    R.uo = f3((R.oo.x + V.extent) * V.inv_cell, (R.oo.y + V.extent) * V.inv_cell, (R.oo.z + V.extent) * V.inv_cell);
    R.ud = R.od * V.inv_cell;
    R.cmax = (float)(V.N - 2);
    R.t0 = (R.t_enter > 0.0f ? R.t_enter : 0.0f) + F.eps_in;
    R.t_end = minf_(t_exit, t_cur);
    R.base_min = __builtin_fmaf(t_base, F.cone_eps, F.step_min);
    R.leap_unit = (4.0f * V.cell) * inv_len;
    R.cell_unit = R.leap_unit * 0.25f;
    return clip_to_active_box(V, f3(I->sh_inv[0], I->sh_inv[1], I->sh_inv[2]), R);
}

/* Cell + fraction of the sample at ray parameter t: cell = clamp(floor(u), 0, N-2) (v_med3_f32). */
struct Cell {
    int cx, cy, cz;
    float fx, fy, fz;
};
/* v_min_f32 / v_max_f32 as they are: fminf / fmaxf make the compiler put a quieting v_max x,x in front of every operand
 * it cannot prove free of signalling NaNs (IEEE mode), which in the march loop is most of them.  Same results as fminf /
 * fmaxf for every input (a NaN operand loses against a number). */
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ Cell cell_at(const RaySeg& R, float t) {
    const float ux = __builtin_fmaf(R.ud.x, t, R.uo.x), uy = __builtin_fmaf(R.ud.y, t, R.uo.y); /* (not packed: see lerp8) */
    const float uz = __builtin_fmaf(R.ud.z, t, R.uo.z);
    const float cxf = __builtin_amdgcn_fmed3f(floorf(ux), 0.0f, R.cmax);
    const float cyf = __builtin_amdgcn_fmed3f(floorf(uy), 0.0f, R.cmax);
    const float czf = __builtin_amdgcn_fmed3f(floorf(uz), 0.0f, R.cmax);
    Cell c;
    c.fx = ux - cxf;
    c.fy = uy - cyf;
    c.fz = uz - czf;
    c.cx = (int)cxf;
    c.cy = (int)cyf;
    c.cz = (int)czf;
    return c;
}

/* Empty-space leap (ray-parameter units) of cell c from the table word of its brick: the cell's sub-block nibble, that many
 * cell edges (at most 15).  The table is small (nb^3 words) and stays in L2.  Round 2 also read the brick-level byte (leaps of
 * whole bricks where the nearest near brick is >= 2 bricks away): two of every five vector loads of a frame were table loads, and
 * the texture path is what the march keeps busiest (TD 0.88): ONE table load per new brick, +3.7 % (the march is clipped to the
 * active box anyway: leaps beyond 15 cells were 2 % of the skipped positions). */
__device__ __forceinline__ float leap_of(const RaySeg& R, const Cell& c, unsigned nibw) {
    const unsigned k = (((unsigned)c.cx >> 1) & 1u) * 4u + (((unsigned)c.cz >> 1) & 1u) * 2u + (((unsigned)c.cy >> 1) & 1u);
    return (float)((nibw >> (4u * k)) & 15u) * R.cell_unit;
}

constexpr int kRefine = 3; /* secant samples spent on a hit that overshot into the surface */

/*
 * A step that lands inside the surface (s < 0) overshot: band-edge cells of shell volumes interpolate
 * towards the background value, and a trilinear SDF is not exactly 1-Lipschitz.  Walk back to the
 * crossing with kRefine regula-falsi samples between the last outside sample (ta, sa > 0) and the first
 * inside one (tb, sb < 0); returns the last secant point and its cell.  Taps come from global memory
 * (a few samples per overshooting hit).
 */
template <int PATH>
__device__ __forceinline__ float refine_hit(const VolRef& V, const RaySeg& R, float ta, float sa, float tb, float sb, Cell& c,
                                            unsigned& steps) {
    float tm = tb;
#pragma unroll 1
    for (int r = 0; r < kRefine; r++) {
        tm = __builtin_fmaf(tb - ta, sa / (sa - sb), ta);
        c = cell_at(R, tm);
        const float sm = trilinear<PATH>(V, c.cx, c.cy, c.cz, c.fx, c.fy, c.fz) * R.ds;
        steps++;
        if (sm < 0.0f) {
            tb = tm;
            sb = sm;
        } else {
            ta = tm;
            sa = sm;
        }
    }
    return tm;
}

/*
 * Hit polish.  The cone threshold stops a ray up to a few pixel footprints in FRONT of the surface; the
 * reference reports the zero crossing itself and takes its normal there (Voxel.hlsli:691-804).  The hit decision stands; a
 * closest hit's position moves on to the crossing by the secant rule: first estimate from the march's own last two samples
 * (when the previous one is a real, unclamped sample), else one sphere step; n samples, each followed by the secant through
 * the last two points (equal samples: converged, stay); never behind the stop point, never more than 8 thresholds ahead,
 * never beyond the interval.  Restated operation by operation in the oracle's march_instance.
 */
template <int PATH>
__device__ __forceinline__ float polish_hit(const DFrame& F, const VolRef& V, const RaySeg& R, float t, float s, float t_prev,
                                            float s_prev, int n, Cell& c, unsigned& steps) {
    const float t_far = t + 8.0f * __builtin_fmaf(t, F.cone_eps, F.eps_hit);
    float ta = t, sa = s;
    float tb = (s_prev > s && s_prev < R.smax) ? t + (s * (t - t_prev)) / (s_prev - s) : t + s;
    tb = vmin(vmax(tb, t), t_far);
#pragma unroll 1
    for (int r = 0; r < n; r++) {
        const Cell cb = cell_at(R, tb);
        const float sb = trilinear<PATH>(V, cb.cx, cb.cy, cb.cz, cb.fx, cb.fy, cb.fz) * R.ds;
        steps++;
        const float tm = sa != sb ? vmin(vmax(__builtin_fmaf(tb - ta, sb / (sa - sb), tb), t), t_far) : tb;
        ta = tb;
        sa = sb;
        tb = tm;
    }
    if (!(tb <= R.t_end)) return t;
    c = cell_at(R, tb);
    return tb;
}

/* World-space normal at a hit found in cell c after `iter` march iterations.  Taps always come from
 * global memory here (once per hit).  ZO: the kernel honours VRT_FLAG_REFERENCE_BOUNDARY_TEXELS (zero_outside at run time): a neighbour
 * cell beyond the grid is then the reference's — its samples outside the volume texture read 0 (GetDensity, Voxel.hlsli:607-617), so its
 * interpolant is the boundary plane's times the in-texture side's weight: cell N-1 = (1 - f) x cell N-2 at fraction 1, cell -1 = f x cell 0
 * at fraction 0 (oracle: the same products).  Default: the neighbour cell clamped to the grid. */
template <int PATH, bool EXACT = false, bool ZO = false>
__device__ __forceinline__ F3 hit_normal(const DInstance* __restrict__ I, const VolRef& V, const RaySeg& R, const Cell& c,
                                         int iter, bool zero_outside = false) {
    F3 n;
    if (iter == 0 && R.t_enter >= 0.0f && !R.clipped) {
        /* surface cut by the volume boundary: AABB-face normal (Raytracing.hlsl:198-226) */
        float tb = R.t_enter - 0.1f;
        float rx = __builtin_fmaf(R.od.x, tb, R.oo.x);
        float ry = __builtin_fmaf(R.od.y, tb, R.oo.y);
        float rz = __builtin_fmaf(R.od.z, tb, R.oo.z);
        float e = V.extent;
        n.x = rx > e ? 1.0f : (rx < -e ? -1.0f : 0.0f);
        n.y = ry > e ? 1.0f : (ry < -e ? -1.0f : 0.0f);
        n.z = rz > e ? 1.0f : (rz < -e ? -1.0f : 0.0f);
    } else {
        /* central differences of the interpolant one cell either side (Voxel.hlsli:783-804) */
        const int N2 = V.N - 2;
        int xp = c.cx + 1 > N2 ? N2 : c.cx + 1, xm = c.cx - 1 < 0 ? 0 : c.cx - 1;
        int yp = c.cy + 1 > N2 ? N2 : c.cy + 1, ym = c.cy - 1 < 0 ? 0 : c.cy - 1;
        int zp = c.cz + 1 > N2 ? N2 : c.cz + 1, zm = c.cz - 1 < 0 ? 0 : c.cz - 1;
        const bool zo = ZO && zero_outside;
        const bool hx = zo && c.cx + 1 > N2, hy = zo && c.cy + 1 > N2, hz = zo && c.cz + 1 > N2; /* the +1 neighbour lies beyond the grid */
        const bool lx = zo && c.cx - 1 < 0, ly = zo && c.cy - 1 < 0, lz = zo && c.cz - 1 < 0;    /* the -1 neighbour */
        const float spx = 1.0f - c.fx, spy = 1.0f - c.fy, spz = 1.0f - c.fz;
        /* one axis at a time (a real loop): the six interpolations' 24 tap loads all in flight at once were the register peak of the
           whole kernel (48 registers of taps); a hit happens once per ray, its latency is not what the kernel waits for */
        n = f3(0.0f, 0.0f, 0.0f);
#pragma unroll 1
        for (int a = 0; a < 3; a++) {
            const int px_ = a == 0 ? xp : c.cx, mx_ = a == 0 ? xm : c.cx;
            const int py_ = a == 1 ? yp : c.cy, my_ = a == 1 ? ym : c.cy;
            const int pz_ = a == 2 ? zp : c.cz, mz_ = a == 2 ? zm : c.cz;
            float v;
            if constexpr (ZO) {
                const bool hi_out = a == 0 ? hx : a == 1 ? hy : hz, lo_out = a == 0 ? lx : a == 1 ? ly : lz;
                float vp = trilinear<PATH>(V, px_, py_, pz_, (a == 0 && hi_out) ? 1.0f : c.fx, (a == 1 && hi_out) ? 1.0f : c.fy, (a == 2 && hi_out) ? 1.0f : c.fz);
                float vm = trilinear<PATH>(V, mx_, my_, mz_, (a == 0 && lo_out) ? 0.0f : c.fx, (a == 1 && lo_out) ? 0.0f : c.fy, (a == 2 && lo_out) ? 0.0f : c.fz);
                /* (products and differences of this iteration, selected by axis: a select chain over the cell's own fields would be
                   turned into an indexed load from a stack copy of the cell) */
                const float vps = a == 0 ? spx * vp : a == 1 ? spy * vp : spz * vp;
                const float vms = a == 0 ? c.fx * vm : a == 1 ? c.fy * vm : c.fz * vm;
                v = (hi_out ? vps : vp) - (lo_out ? vms : vm);
            } else {
                v = trilinear<PATH>(V, px_, py_, pz_, c.fx, c.fy, c.fz) - trilinear<PATH>(V, mx_, my_, mz_, c.fx, c.fy, c.fz);
            }
            n.x = a == 0 ? v : n.x;
            n.y = a == 1 ? v : n.y;
            n.z = a == 2 ? v : n.z;
        }
    }
    float l2 = dot3(n, n);
    if (!(l2 > 0.0f)) {
        n = f3(0.0f, 0.0f, 0.0f);
    } else if constexpr (EXACT) {
        n = n * (1.0f / sqrtf(l2)); /* the normal steers a mirror bounce: keep it bit-identical to the oracle */
    } else {
        n = n * fast_rsq(l2);
    }
    return mul33(I->o2w, n);
}

/*
 * Cube render modes (SH/Raytracing_Cube*.hlsl:142-295, GoToNextVoxel Voxel.hlsli:133-187): exact traversal of
 * the voxel grid.  Voxel (x,y,z) is the cube of cell (x,y,z), solid when its density is <= 0.  A node is one
 * cell, or — where cube_skip says the nearest solid voxel is D bricks away — the box of (2D-1)^3 bricks around
 * the current brick, crossed in one step like a merged node of the reference's octree.  Hit = entry point of
 * the first solid voxel, normal = the face the ray came through (AABB face for the first voxel of a ray that
 * entered from outside, zero when the ray started inside the volume).  One step = one node visit (1 table
 * byte, + 4 B density in bricks that hold solid voxels).  Restated line by line in oracle march_cube.
 */
template <int NORMAL, bool B16>
__device__ __forceinline__ bool march_cube(const DFrame& F, const DInstance* __restrict__ I, const DVolume* __restrict__ Vd, F3 o,
                                           F3 d, float t_cur, float& t_hit, F3& n_world, unsigned& steps, unsigned& ex) {
    const VolRef V = load_vol<VRT_PATH_BRICK>(Vd);
    RaySeg R;
    if (!setup_ray<false>(F, I, V, o, d, t_cur, 0.0f, R)) return false;
    const float inf = __builtin_inff();
    const float ix = R.ud.x != 0.0f ? 1.0f / R.ud.x : inf;
    const float iy = R.ud.y != 0.0f ? 1.0f / R.ud.y : inf;
    const float iz = R.ud.z != 0.0f ? 1.0f / R.ud.z : inf;
    const int cmax = V.N - 2;
    float t = R.t0;
    const Cell c0 = cell_at(R, t);
    int cx = c0.cx, cy = c0.cy, cz = c0.cz;
    int axis_in = -1;
    const unsigned nb = (unsigned)V.nb;
    const int max_steps = F.max_steps;
    for (int i = 0; i < max_steps; i++) {
        if (t > R.t_end) return false;
        steps++;
        const unsigned brick = mad24(mad24((unsigned)cx >> 2, nb, (unsigned)cz >> 2), nb, (unsigned)cy >> 2);
        const int dist = (int)V.cube[brick];
        int lox, hix, loy, hiy, loz, hiz;
        if (dist == 0) {
            const unsigned local = brick_local(cx, cy, cz);
            float den;
            if constexpr (B16) den = (float)*(const short __attribute__((address_space(1)))*)(V.p + (((brick << 7) + local) << 1));
            else den = *(gfloat_p)(V.p + (((brick << 7) + local) << 2));
            if (den <= 0.0f) {
                t_hit = t;
                if constexpr (NORMAL != 0) {
                    F3 n = f3(0.0f, 0.0f, 0.0f);
                    if (axis_in >= 0) {
                        const float udx = axis_in == 0 ? R.ud.x : axis_in == 1 ? R.ud.y : R.ud.z;
                        const float f = udx > 0.0f ? -1.0f : 1.0f;
                        n = f3(axis_in == 0 ? f : 0.0f, axis_in == 1 ? f : 0.0f, axis_in == 2 ? f : 0.0f);
                    } else if (R.t_enter >= 0.0f) {
                        const float tb = R.t_enter - 0.1f;
                        const float rx = __builtin_fmaf(R.od.x, tb, R.oo.x);
                        const float ry = __builtin_fmaf(R.od.y, tb, R.oo.y);
                        const float rz = __builtin_fmaf(R.od.z, tb, R.oo.z);
                        const float e = V.extent;
                        n.x = rx > e ? 1.0f : (rx < -e ? -1.0f : 0.0f);
                        n.y = ry > e ? 1.0f : (ry < -e ? -1.0f : 0.0f);
                        n.z = rz > e ? 1.0f : (rz < -e ? -1.0f : 0.0f);
                    }
                    n_world = mul33(I->o2w, n);
                }
                return true;
            }
            lox = hix = cx;
            loy = hiy = cy;
            loz = hiz = cz;
        } else {
            const int r = dist - 1;
            lox = ((cx >> 2) - r) * 4;
            hix = ((cx >> 2) + r) * 4 + 3;
            loy = ((cy >> 2) - r) * 4;
            hiy = ((cy >> 2) + r) * 4 + 3;
            loz = ((cz >> 2) - r) * 4;
            hiz = ((cz >> 2) + r) * 4 + 3;
            lox = lox < 0 ? 0 : lox;
            loy = loy < 0 ? 0 : loy;
            loz = loz < 0 ? 0 : loz;
            hix = hix > cmax ? cmax : hix;
            hiy = hiy > cmax ? cmax : hiy;
            hiz = hiz > cmax ? cmax : hiz;
        }
        const float bx = (float)(R.ud.x > 0.0f ? hix + 1 : lox);
        const float by = (float)(R.ud.y > 0.0f ? hiy + 1 : loy);
        const float bz = (float)(R.ud.z > 0.0f ? hiz + 1 : loz);
        const float tx = R.ud.x != 0.0f ? (bx - R.uo.x) * ix : inf;
        const float ty = R.ud.y != 0.0f ? (by - R.uo.y) * iy : inf;
        const float tz = R.ud.z != 0.0f ? (bz - R.uo.z) * iz : inf;
        const int axis = tx < ty ? (tx < tz ? 0 : 2) : (ty < tz ? 1 : 2);
        const float t_new = axis == 0 ? tx : axis == 1 ? ty : tz;
        if (!(t_new <= R.t_end)) return false; /* leaves the march interval (or NaN) before the next node */
        const float ux = floorf(__builtin_fmaf(R.ud.x, t_new, R.uo.x));
        const float uy = floorf(__builtin_fmaf(R.ud.y, t_new, R.uo.y));
        const float uz = floorf(__builtin_fmaf(R.ud.z, t_new, R.uo.z));
        cx = axis == 0 ? (R.ud.x > 0.0f ? hix + 1 : lox - 1) : (int)__builtin_amdgcn_fmed3f(ux, (float)lox, (float)hix);
        cy = axis == 1 ? (R.ud.y > 0.0f ? hiy + 1 : loy - 1) : (int)__builtin_amdgcn_fmed3f(uy, (float)loy, (float)hiy);
        cz = axis == 2 ? (R.ud.z > 0.0f ? hiz + 1 : loz - 1) : (int)__builtin_amdgcn_fmed3f(uz, (float)loz, (float)hiz);
        const int ca = axis == 0 ? cx : axis == 1 ? cy : cz;
        if (ca < 0 || ca > cmax) return false;
        t = maxf_(t_new, t);
        axis_in = axis;
    }
    if (max_steps > 0 && !(t > R.t_end)) ex += kExhaustedOne;
    return false;
}

/*
 * What a lane does with the sample s it just took at t (the oracle's march loop, same operations in the same order):
 *
 *     s_clamped = min(s, smax)
 *     back      = max(s_clamped, 0) + chk < t - t_prev          over-relaxation (k_relax > 1): the step that led here was a
 *                                                               stretched one (chk = the previous sample's empty radius, else
 *                                                               +inf) and the two empty spheres do not overlap: something may
 *                                                               have been jumped over, the ray goes BACK to the previous
 *                                                               sample's plain step (that sample stays the "previous" one)
 *     if (!back && s < fma(t, cone_eps, eps_hit)) { s_hit = s; t_end = -inf; }      hit: t, t_prev, s_prev, i stay
 *     else {
 *         i++
 *         s_gate    = 0.8 * s_prev
 *         from      = back ? t_prev : t                         where the next step starts
 *         radius    = back ? s_prev : s_clamped                 the empty radius there
 *         adv_min   = max(fma(from, cone_eps, base_min), back ? 0 : leap)
 *         plain     = max(radius, adv_min)
 *         stretched = max(min(s * k_relax, smax_relax), adv_min)
 *         relax     = !back && stretched > plain && s >= s_gate && t + stretched <= R.t_end
 *                                                               stretched only while the distance is not falling fast (a ray
 *                                                               running at a surface would overshoot and come back) and the
 *                                                               next sample stays inside the interval
 *         t_prev = from;  s_prev = radius;  chk = relax ? radius : +inf
 *         t = from + ((relax || k_relax < 1) ? stretched : plain)
 *     }
 *
 * Written out as ISA: the compiler's form of these lines carried 12 more vector and 7 more scalar instructions per position
 * (register copies at the joins of its branches, the second branch's mask bookkeeping) out of ~93 + 42 in the loop, and a
 * position's time is made of the instructions on its wave's in-order path.  Here the hit lanes and the others are
 * separated by the exec mask once, every state variable is updated in place, nothing is copied, and the relax condition narrows the
 * exec mask with three v_cmpx instead of combining compare results in scalar registers.  gfx950 hazards kept by hand inside the
 * block: a mask written by a VALU compare is read by a VALU select no earlier than the third instruction after it (2 wait states),
 * masks that come out of an SALU instruction need none.
 */
constexpr float kRelaxGate = 0.8f; /* over-relaxation: a step is stretched only when the sample is at least this fraction of the one before */
/* a frame constant where the assembler wants a scalar register (it is wave-uniform; where the compiler already knows, this folds away) */
__device__ __forceinline__ float uniform(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }
static_assert(__builtin_bit_cast(unsigned, kRelaxGate) == 0x3f4ccccdu, "the literal in step_from_sample");
__device__ __forceinline__ void step_from_sample(const DFrame& F, const RaySeg& R, float s, float leap, int relax_always,
                                                 float& t, float& t_prev, float& s_prev, float& chk, int& i, float& t_end, float& s_hit, unsigned& steps) {
    float a, b, c, d;
    unsigned long long m_back, m_save;
    asm("v_add_u32 %[n], 1, %[n]\n\t"                          /* one more sample taken */
        "v_min_f32 %[a], %[s], %[smax]\n\t"
        "v_sub_f32 %[c], %[t], %[tp]\n\t"
        "v_max_f32 %[b], 0, %[a]\n\t"
        "v_add_f32 %[b], %[b], %[chk]\n\t"
        "v_cmp_lt_f32_e64 %[mb], %[b], %[c]\n\t"             /* back */
        "v_mov_b32 %[b], %[eps]\n\t"
        "v_fma_f32 %[b], %[t], %[cone], %[b]\n\t"
        "v_cmp_lt_f32_e32 vcc, %[s], %[b]\n\t"
        "s_andn2_b64 vcc, vcc, %[mb]\n\t"                      /* hit */
        "s_and_saveexec_b64 %[ms], vcc\n\t"                    /* the lanes that hit */
        "v_mov_b32 %[sh], %[s]\n\t"
        "v_mov_b32 %[te], 0xff800000\n\t"
        "s_andn2_b64 exec, %[ms], vcc\n\t"                     /* the lanes that go on */
        "v_mul_f32 %[d], 0x3f4ccccd, %[sp]\n\t"                /* s_gate */
        "v_cndmask_b32_e64 %[tp], %[t], %[tp], %[mb]\n\t"      /* from, in place */
        "v_cndmask_b32_e64 %[sp], %[a], %[sp], %[mb]\n\t"      /* radius, in place */
        "v_fma_f32 %[b], %[tp], %[cone], %[bmin]\n\t"
        "v_cndmask_b32_e64 %[c], %[leap], 0, %[mb]\n\t"
        "v_max_f32 %[b], %[b], %[c]\n\t"                       /* adv_min */
        "v_max_f32 %[c], %[sp], %[b]\n\t"                      /* plain */
        "v_mul_f32 %[a], %[krelax], %[s]\n\t"
        "v_min_f32 %[a], %[a], %[smaxr]\n\t"
        "v_max_f32 %[a], %[a], %[b]\n\t"                       /* stretched */
        "v_add_f32 %[b], %[t], %[a]\n\t"                       /* t + stretched */
        "s_mov_b32 vcc_lo, %[always]\n\t"
        "s_mov_b32 vcc_hi, %[always]\n\t"
        "v_mov_b32 %[chk], 0x7f800000\n\t"
        "v_add_u32 %[i], 1, %[i]\n\t"
        "v_cndmask_b32_e32 %[t], %[c], %[a], vcc\n\t"          /* the step unless relaxed: plain (k_relax < 1: the scaled one) */
        "v_add_f32 %[t], %[tp], %[t]\n\t"
        "s_andn2_b64 exec, exec, %[mb]\n\t"                    /* relax = !back ... */
        "v_cmpx_gt_f32_e32 vcc, %[a], %[c]\n\t"                /* ... && stretched > plain */
        "v_cmpx_ge_f32_e32 vcc, %[s], %[d]\n\t"                /* ... && s >= s_gate */
        "v_cmpx_le_f32_e32 vcc, %[b], %[rtend]\n\t"            /* ... && t + stretched <= R.t_end */
        "v_mov_b32 %[chk], %[sp]\n\t"
        "v_add_f32 %[t], %[tp], %[a]\n\t"
        "s_mov_b64 exec, %[ms]"
        : [t] "+v"(t), [tp] "+v"(t_prev), [sp] "+v"(s_prev), [chk] "+v"(chk), [i] "+v"(i), [te] "+v"(t_end), [sh] "+v"(s_hit), [n] "+v"(steps),
          [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [d] "=&v"(d), [mb] "=&s"(m_back), [ms] "=&s"(m_save)
        : [s] "v"(s), [leap] "v"(leap), [smax] "v"(R.smax), [bmin] "v"(R.base_min), [smaxr] "v"(R.smax_relax), [rtend] "v"(R.t_end),
          [cone] "s"(uniform(F.cone_eps)), [eps] "s"(uniform(F.eps_hit)), [krelax] "s"(uniform(F.k_relax)),
          [always] "s"(relax_always)
        : "vcc", "scc");
}

/* ... and with a position whose table entry shows no active cell within the leap (no tap read, no sample counted):
 *     t_prev = t;  s_prev = smax;  chk = +inf;  t = t + max(fma(t, cone_eps, base_min), leap);  i++
 * The sample counter is named (and left alone) so that both arms of the branch define it in place: otherwise the compiler carries it
 * through the join with three register copies per position. */
__device__ __forceinline__ void step_over_empty_space(const DFrame& F, const RaySeg& R, float leap, float& t, float& t_prev, float& s_prev,
                                                      float& chk, int& i, unsigned& steps) {
    float b;
    asm("v_mov_b32 %[tp], %[t]\n\t"
        "v_fma_f32 %[b], %[t], %[cone], %[bmin]\n\t"
        "v_max_f32 %[b], %[b], %[leap]\n\t"
        "v_mov_b32 %[sp], %[smax]\n\t"
        "v_mov_b32 %[chk], 0x7f800000\n\t"
        "v_add_f32 %[t], %[t], %[b]\n\t"
        "v_add_u32 %[i], 1, %[i]"
        : [t] "+v"(t), [tp] "+v"(t_prev), [sp] "+v"(s_prev), [chk] "+v"(chk), [i] "+v"(i), [n] "+v"(steps), [b] "=&v"(b)
        : [leap] "v"(leap), [smax] "v"(R.smax), [bmin] "v"(R.base_min), [cone] "s"(uniform(F.cone_eps)));
}

/* Per-lane march state.  The per-lane kernels run it from start to end; the hybrid (LDS) march stops it after its head
 * and hands it to the wave-cooperative tail. */
struct MarchState {
    float t, t_prev, s_prev, s_hit;
    int i;                 /* iterations so far: positions visited, sampled or skipped */
    bool hit;
    float chk;             /* over-relaxation (k_relax > 1): s_prev when the step that led to the current position was a stretched
                              one (the next sample checks the overlap of the two empty spheres), else +inf */
    Cell c;
};

/*
 * The march loop of one lane on data path DP, from state `st` until a hit, the end of the interval, or `limit`
 * iterations.  One iteration = one position: either skipped (the two-level table shows no active cell within the
 * leap: the ray advances, no tap is read, no sample counted) or sampled.
 * Table words are re-read only when the ray changes brick.  (Two refinements were measured and removed — speculative taps
 * for the next brick, a cell's taps kept in registers across samples: profiles/r02_ab_march_variants.txt.)  The loop has ONE
 * exit test, and the sample path is written with selects: every instruction taken out of it is ~0.4 % of the frame.
 * Same positions, same values, same counters as the oracle's loop.
 */
template <int DP, bool DIAG, bool TABLES, bool RELOAD>
__device__ __forceinline__ void march_lane_on(const DFrame& F, const VolRef& V, const RaySeg& R, MarchState& st, int limit, unsigned& steps,
                                              DiagAcc* dg) {
    float t = st.t, t_prev = st.t_prev, s_prev = st.s_prev, chk = st.chk;
    /* the lane's own end of the interval: -inf once it has hit, so that the loop has ONE exit test and no break (a divergent
       break costs a handful of mask operations in every iteration of every lane) */
    float t_end = R.t_end;
    int i = st.i;
    Cell c = st.c;
    constexpr bool tables = TABLES;
    unsigned last_brick = 0xffffffffu, nibw = 0u;
    float s_hit = st.s_hit;
    /* k_relax < 1 (wave-uniform): every distance-driven step is scaled down, never stretched */
    /* ... as a scalar mask word; in integer arithmetic (k_relax > 0: its bits order like its value), so that it stays in a scalar register */
    const int relax_always = (__builtin_bit_cast(int, F.k_relax) - 0x3f800000) >> 31;
    while (i < limit && !(t > t_end)) {
        unsigned long long st0 = 0, st1 = 0;
        if constexpr (DIAG) st0 = stamp();
        c = cell_at(R, t);
        if constexpr (DIAG) {
            asm volatile("" ::"v"(c.cx), "v"(c.cy), "v"(c.cz), "v"(c.fx), "v"(c.fy), "v"(c.fz));
            st1 = stamp();
        }
        const unsigned brick = brick_index(V, c.cx, c.cy, c.cz);
        float leap = 0.0f;
        bool skip = false;
        if constexpr (tables) {
            if constexpr (RELOAD) {
                /* the kernels that walk a BVH are short of registers (72, 7 waves per SIMD, spills): there the word is read in
                   every trip and the "same brick?" test with its two registers goes (config 5 +1.6 %; the single-volume kernel
                   loses 0.5-2 % that way: profiles/r05_ab_step_asm.txt (e)) */
                nibw = table_word(V, brick);
            } else if (brick != last_brick) { /* the table word is re-read only when the ray changes brick */
                nibw = table_word(V, brick);
                last_brick = brick;
            }
            leap = leap_of(R, c, nibw);
            /* no active cell within the leap, and the hit threshold still below half the clamp: the sample could neither hit
               nor shorten the step (oracle: same condition, same advance; smax > 0 wherever there are tables) */
            skip = leap >= R.smax && t <= R.t_skip_end;
        }
        if (skip) {
            step_over_empty_space(F, R, leap, t, t_prev, s_prev, chk, i, steps);
            if constexpr (DIAG) dg->iters++;
        } else {
            const Taps taps = fetch8_at<DP>(V, brick, c.cx, c.cy, c.cz);
            if constexpr (DIAG) {
                asm volatile("s_waitcnt vmcnt(0)" ::"v"(taps.y00a), "v"(taps.y00b), "v"(taps.y01a), "v"(taps.y01b), "v"(taps.y10a),
                             "v"(taps.y10b), "v"(taps.y11a), "v"(taps.y11b));
                const unsigned long long lat = stamp() - st1;
                dg->mem += lat; /* address arithmetic + table words + 4 loads until the data is back */
                dg->fetches += lat < 450 ? 1u : 0u;
                dg->iters++;
            }
            const float s = lerp8(taps, c.fx, c.fy, c.fz) * R.ds;
            step_from_sample(F, R, s, leap, relax_always, t, t_prev, s_prev, chk, i, t_end, s_hit, steps);
        }
        if constexpr (DIAG) dg->loop += stamp() - st0;
    }
    st.hit = t_end < R.t_end;
    st.s_hit = s_hit;
    st.t = t;
    st.t_prev = t_prev;
    st.s_prev = s_prev;
    st.i = i;
    st.c = c;
    st.chk = chk;
}
/* The loop exists twice, with and without the empty-space tables: tested inside it, the (wave-uniform, loop-invariant) question costs
 * every position two scalar instructions and a taken branch. */
template <int DP, bool DIAG, bool RELOAD = false>
__device__ __forceinline__ void march_lane(const DFrame& F, const VolRef& V, const RaySeg& R, MarchState& st, int limit, unsigned& steps,
                                           DiagAcc* dg) {
    if (V.skip != nullptr) march_lane_on<DP, DIAG, true, RELOAD>(F, V, R, st, limit, steps, dg);
    else march_lane_on<DP, DIAG, false, false>(F, V, R, st, limit, steps, dg);
}

/*
 * Sphere-trace one instance (per-lane control flow).  o,d: world-space ray (d normalised).  Returns
 * true on hit and the ray parameter (shared by world and object space — the object-space direction
 * is not re-normalised, DXR semantics).  NORMAL: also produce the world-space normal.
 */
template <int PATH, int NORMAL /* 0 none, 1 fast length, 2 exact length */, bool DIAG = false, bool DIR_SHADOW = false,
          bool REF = false /* the kernel honours the reference-artefact flags (DFrame::zero_outside); false: compiled out */,
          bool MULTI = false /* called from a BVH walk: the march trades the table word's register pair for a load per trip */>
__device__ __forceinline__ bool march_instance(const DFrame& F, const DInstance* __restrict__ I,
                                               const DVolume* __restrict__ Vd, F3 o, F3 d, float t_cur, float t_base,
                                               float& t_hit, F3& n_world, unsigned& steps, unsigned& ex, DiagAcc* dg = nullptr) {
    if constexpr (PATH == kPathCube || PATH == kPathCube16) {
        return march_cube<NORMAL, PATH == kPathCube16>(F, I, Vd, o, d, t_cur, t_hit, n_world, steps, ex);
    } else {
    constexpr int DP = data_path<PATH>(); /* where the taps come from */
    const VolRef V = load_vol<DP>(Vd);
    RaySeg R;
    if constexpr (DIR_SHADOW) { /* d is the scene's directional light: its per-instance constants come precomputed */
        if (!setup_shadow_ray(F, I, V, o, t_cur, t_base, R)) return false;
    } else {
        if (!setup_ray(F, I, V, o, d, t_cur, t_base, R)) return false;
    }
    MarchState st;
    st.t = st.t_prev = R.t0;
    st.s_prev = st.s_hit = 0.0f;
    st.i = 0;
    st.hit = false;
    st.chk = __builtin_inff();
    st.c = Cell{0, 0, 0, 0.0f, 0.0f, 0.0f};
    march_lane<DP, DIAG, MULTI>(F, V, R, st, F.max_steps, steps, dg);
    if (!st.hit) {
        if (F.max_steps > 0 && st.i >= F.max_steps && !(st.t > R.t_end)) ex += kExhaustedOne; /* budget ran out inside the volume: reported, treated as a miss */
        return false;
    }
    float t = st.t;
    Cell c = st.c;
    if (st.s_hit < 0.0f && st.i > 0) t = refine_hit<DP>(V, R, st.t_prev, st.s_prev, t, st.s_hit, c, steps);
    else if (NORMAL != 0 && F.polish > 0 && st.i > 0) t = polish_hit<DP>(F, V, R, t, st.s_hit, st.t_prev, st.s_prev, F.polish, c, steps);
    t_hit = t;
    if constexpr (NORMAL == 1) n_world = hit_normal<DP, false, REF>(I, V, R, c, st.i, REF && F.zero_outside != 0);
    if constexpr (NORMAL == 2) n_world = hit_normal<DP, true, REF>(I, V, R, c, st.i, REF && F.zero_outside != 0);
    return true;
    }
}

/* Closest hit over the scene.  SINGLE: exactly one instance, no BVH, all scene data wave-uniform. */
template <int PATH, bool SINGLE, bool DIAG = false, int NORMAL = 1, bool REF = false>
__device__ __forceinline__ bool trace_closest(const DFrame& F, F3 o, F3 d, float t_max, float t_base, float& t_best,
                                              int& inst_best, F3& n_best, unsigned& steps, unsigned& ex, DiagAcc* dg = nullptr) {
    if constexpr (SINGLE) {
        float t;
        F3 n;
        if (march_instance<PATH, NORMAL, DIAG, false, REF>(F, F.inst, F.vol0, o, d, t_max, t_base, t, n, steps, ex, dg)) {
            t_best = t;
            inst_best = 0;
            n_best = n;
            return true;
        }
        return false;
    } else {
        bool any = false;
        float best = t_max;
        /* The WAVE walks the tree (DBvhNode::right is every node's successor when its subtree is left out, the left child of
           an inner node is the node behind it): a node is visited while ANY lane's ray meets its box, so the node, the
           instance and its volume are wave-uniform — scalar registers and scalar loads — and a lane marches an instance only
           when its own ray meets the box.  Per-lane traversal kept a stack and every instance / volume field in vector
           registers (106 instead of 62).  Same visiting order for every lane as before, and the result does not depend on it. */
        int ni = 0;
        while (ni < F.n_nodes) {
            ni = __builtin_amdgcn_readfirstlane(ni);
            const DBvhNode nd = F.nodes[ni];
            const bool in = slab_box(o, d, nd, best);
            if (__ballot(in) == 0ull) {
                ni = nd.right;
                continue;
            }
            if (nd.left < 0) {
                const int ii = -nd.left - 1;
                const DInstance* I = F.inst + ii;
                if (in) {
                    float t;
                    F3 n;
                    /* a sphere-trace covers the instance's whole interval whatever was hit before (where its stretched steps
                       fall depends on the interval's end: cut at `best` the result would depend on the visiting order); the
                       cell walk of the Cube modes has no such state and stops at the closest hit so far */
                    constexpr bool kCube = PATH == kPathCube || PATH == kPathCube16;
                    if (march_instance<PATH, NORMAL, DIAG, false, REF, true>(F, I, F.vols + I->slot, o, d, kCube ? best : t_max, t_base, t, n, steps, ex, dg)) {
                        if (!any || t < best || (t == best && ii < inst_best)) {
                            any = true;
                            best = t;
                            t_best = t;
                            inst_best = ii;
                            n_best = n;
                        }
                    }
                }
                ni = nd.right;
            } else {
                ni = ni + 1;
            }
        }
        return any;
    }
}

template <int PATH, bool SINGLE, bool DIAG = false, bool DIR_SHADOW = false>
__device__ __forceinline__ bool trace_any(const DFrame& F, F3 o, F3 d, float t_max, float t_base, unsigned& steps, unsigned& ex,
                                          DiagAcc* dg = nullptr) {
    float t;
    F3 n;
    if constexpr (SINGLE) {
        return march_instance<PATH, 0, DIAG, DIR_SHADOW>(F, F.inst, F.vol0, o, d, t_max, t_base, t, n, steps, ex, dg);
    } else {
        bool found = false; /* this lane's ray is blocked: it takes no further part, the wave goes on for the others */
        int ni = 0;
        while (ni < F.n_nodes) {
            if (__ballot(!found) == 0ull) break;
            ni = __builtin_amdgcn_readfirstlane(ni);
            const DBvhNode nd = F.nodes[ni];
            const bool in = !found && slab_box(o, d, nd, t_max);
            if (__ballot(in) == 0ull) {
                ni = nd.right;
                continue;
            }
            if (nd.left < 0) {
                const DInstance* I = F.inst + (-nd.left - 1);
                if (in && march_instance<PATH, 0, DIAG, DIR_SHADOW, false, true>(F, I, F.vols + I->slot, o, d, t_max, t_base, t, n, steps, ex, dg)) found = true;
                ni = nd.right;
            } else {
                ni = ni + 1;
            }
        }
        return found;
    }
}

/* Cube-map point lookup with SampleLevel(dir.xzy) (Raytracing.hlsl:444-449). */
__device__ __forceinline__ unsigned env_fetch(const uint8_t* __restrict__ env, int S, F3 dir) {
    if (env == nullptr) return 0u;
    float vx = dir.x, vy = dir.z, vz = dir.y;
    float ax = fabsf(vx), ay = fabsf(vy), az = fabsf(vz);
    int face;
    float sc, tc, ma;
    if (ax >= ay && ax >= az) {
        ma = ax;
        if (vx >= 0.0f) { face = 0; sc = -vz; tc = -vy; }
        else            { face = 1; sc = vz;  tc = -vy; }
    } else if (ay >= az) {
        ma = ay;
        if (vy >= 0.0f) { face = 2; sc = vx; tc = vz; }
        else            { face = 3; sc = vx; tc = -vz; }
    } else {
        ma = az;
        if (vz >= 0.0f) { face = 4; sc = vx;  tc = -vy; }
        else            { face = 5; sc = -vx; tc = -vy; }
    }
    const float inv_ma = 1.0f / ma; /* one correctly rounded division, two products (kernel and oracle alike) */
    float u = (sc * inv_ma + 1.0f) * 0.5f;
    float v = (tc * inv_ma + 1.0f) * 0.5f;
    int ix = (int)floorf(u * (float)S);
    int iy = (int)floorf(v * (float)S);
    ix = ix < 0 ? 0 : (ix > S - 1 ? S - 1 : ix);
    iy = iy < 0 ? 0 : (iy > S - 1 ? S - 1 : iy);
    return ((guint_p)env)[(face * S + iy) * S + ix]; /* little-endian R,G,B,A bytes */
}
__device__ __forceinline__ F3 env_decode(unsigned px) {
    const float k = 1.0f / 255.0f;
    return f3((float)(px & 0xffu) * k, (float)((px >> 8) & 0xffu) * k, (float)((px >> 16) & 0xffu) * k);
}
__device__ __forceinline__ F3 env_lookup(const uint8_t* __restrict__ env, int S, F3 dir) { return env_decode(env_fetch(env, S, dir)); }

/* Radiance(), Lighting.hlsli:50-101 (F enters twice, PI = 3.141592f as in Constants.hlsli). */
/* BRDF(wi, wo) of Lighting.hlsli:77-96; Radiance = (BRDF * Li) * dot(n, wi)  (:98-101). */
__device__ __forceinline__ F3 brdf_eval(F3 wi, F3 wo, F3 n, F3 albedo, float rough, float metal, float k) {
    const float PI_REF = 3.141592f;
    const float INV_PI_REF = 1.0f / 3.141592f;
    F3 hv = wi + wo;
    /* the half vector's length is the correctly rounded one: n.h feeds c = (n.h)^2 (a2 - 1) + 1, which cancels to 1e-3 for smooth
       materials near the highlight — a 1-ulp reciprocal square root there became 2e-4 of D and, through the tone map of a dark
       channel, 4e-4 of a pixel (soak of 10 000 random scenes) */
    F3 h = normalize3(hv);
    F3 f0 = f3(0.04f + (albedo.x - 0.04f) * metal, 0.04f + (albedo.y - 0.04f) * metal,
               0.04f + (albedo.z - 0.04f) * metal);
    float a2 = rough * rough;
    float ndoth = maxf_(dot3(n, h), 0.0f);
    float c = (ndoth * ndoth) * (a2 - 1.0f) + 1.0f;
    float D = a2 * fast_rcp(maxf_((PI_REF * c) * c, 0.001f));
    float wdoth = maxf_(dot3(wo, h), 0.0f);
    float m = maxf_(-wdoth + 1.0f, 0.0f);
    float m2 = m * m;
    float m5 = (m2 * m2) * m;
    F3 Fr = f3(f0.x + (-f0.x + 1.0f) * m5, f0.y + (-f0.y + 1.0f) * m5, f0.z + (-f0.z + 1.0f) * m5);
    float dwo = maxf_(dot3(n, wo), 0.0f);
    float dwi = maxf_(dot3(n, wi), 0.0f);
    float G = (dwo * fast_rcp(dwo * (1.0f - k) + k)) * (dwi * fast_rcp(dwi * (1.0f - k) + k));
    float dg = (D * G) * fast_rcp(maxf_((4.0f * dwo) * dwi, 0.0001f));
    F3 cook = f3(dg * Fr.x, dg * Fr.y, dg * Fr.z);
    float km = 1.0f - metal;
    F3 kd = f3((1.0f - Fr.x) * km, (1.0f - Fr.y) * km, (1.0f - Fr.z) * km);
    return f3((albedo.x * INV_PI_REF) * kd.x + cook.x * Fr.x, (albedo.y * INV_PI_REF) * kd.y + cook.y * Fr.y,
              (albedo.z * INV_PI_REF) * kd.z + cook.z * Fr.z);
}

__device__ __forceinline__ F3 radiance(F3 Li, F3 wi, F3 wo, F3 n, F3 albedo, float rough, float metal, float k) {
    const F3 brdf = brdf_eval(wi, wo, n, albedo, rough, metal, k);
    const float ndwi = dot3(n, wi);
    return f3((brdf.x * Li.x) * ndwi, (brdf.y * Li.y) * ndwi, (brdf.z * Li.z) * ndwi);
}

__device__ __forceinline__ float tonemap(float c) {
    c = c > 0.0f ? c : 0.0f;          /* negative / NaN → 0, what the UNORM render target keeps */
    c = c * fast_rcp(c + 1.0f);        /* Reinhard, Raytracing.hlsl:35 */
    return fast_pow(c, 1.0f / 2.2f);   /* gamma, :36 (pow(0, e) = exp2(-inf) = 0) */
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

/*
 * Primary-ray kernel.  Four consecutive one-wave workgroups of an XCD = one 16x16-pixel tile (block_and_wave below).  blockIdx → tile
 * map (F.tile_map):
 *   SUPERTILE (default): tiles are grouped into 4x4-tile supertiles (64x64 pixels); supertile s
 *     goes to XCD s % 8 (workgroups are dealt round-robin over the 8 XCDs, so the blocks
 *     b ≡ k (mod 8) share XCD k's L2).  An XCD's 16 consecutive blocks cover one supertile, so its
 *     L2 sees the bricks of a compact screen region, while heavy (object) and cheap (sky)
 *     regions are spread over all 8 XCDs.
 *   BAND: XCD k gets the k-th contiguous eighth of the tiles (best L2 locality, worst balance).
 *   LINEAR: tile = blockIdx (consecutive tiles on different XCDs).
 * Placement only affects speed, never results.
 */
/* blockIdx → 16x16-pixel tile under F.tile_map (see the kernel comment). */
/* Which residue class of supertiles an XCD renders moves on by one with every frame of a block (blockIdx.y).  The 8 XCDs work through
 * a launch independently, each through its own eighth of the workgroups, and the launch ends with the slowest: with a fixed map one
 * XCD carries 10 % (config 3) to 18 % (config 5) more marching than the mean, frame after frame (an orbiting camera moves the object
 * slowly); rotated, every XCD has rendered every class after 8 frames.  Config 3: 61.0 -> 64.4 Grays/s (per frame; 63.8 with 8 phases
 * per block, which keeps more of an XCD's L2 warm — balance matters more). */
__device__ __forceinline__ void tile_of_block(const DFrame& F, int b, int nblk, int& tile_x, int& tile_y) {
    if (F.tile_map == kMapSupertile) {
        const int rot = (int)blockIdx.y;
        const int xcd = (b + rot) & 7, q = b >> 3;
        const int st = (q >> 4) * 8 + xcd;      /* supertile index, row-major over st_x columns */
        const int within = q & 15;
        const int st_x = (F.tiles_x + 3) >> 2;
        tile_x = (st % st_x) * 4 + (within & 3);
        tile_y = (st / st_x) * 4 + (within >> 2);
    } else if (F.tile_map == kMapBand) {
        const int xcd = b & 7, q = b >> 3;
        const int per = nblk >> 3, rem = nblk & 7;
        const int L = (xcd < rem) ? xcd * (per + 1) + q : rem * (per + 1) + (xcd - rem) * per + q;
        tile_x = L % F.tiles_x;
        tile_y = L / F.tiles_x;
    } else {
        tile_x = b % F.tiles_x;
        tile_y = b / F.tiles_x;
    }
}

/* Workgroup shape of the per-lane march kernels: one wave = one 8x8-pixel tile per workgroup; four consecutive workgroups OF
 * THE SAME XCD (blockIdx % 8) cover a 16x16 tile, so the tile -> XCD map is the one described above while the dispatcher
 * refills single wave slots (waves of a tile retire at very different times).  (Round 1's four waves per workgroup measured
 * 3.5 % (two frames in flight) to 5 % (one) slower, profiles/r02_ab_march_variants.txt.) */
constexpr int kMarchThreads = 64;
constexpr int kMarchGridMul = 4;
__device__ __forceinline__ void block_and_wave(int& b, int& wave) {
    const int raw = (int)blockIdx.x, xcd = raw & 7, q = raw >> 3;
    wave = q & 3;
    b = ((q >> 2) << 3) | xcd;
}

/* Pixel of a lane: wave w of a 16x16-pixel tile renders the 8x8 pixels at ((w & 1) * 8, (w >> 1) * 8), lane l the pixel (l & 7, l >> 3) of
 * them (row-major).  Placement only: results do not depend on it.  (Measured in round 5, profiles/r05_ab_placement.txt: Morton order inside
 * the tile -0.4 %, a wave of 16x4 pixels -1.2 %, of 4x16 -3.6 %; cache lines per wave-level load 19.3 / 19.6 / 19.6 / 18.0 — the lanes of a
 * marching wave are spread in DEPTH, not over the screen.) */
__device__ __forceinline__ void pixel_of_lane(int tile_x, int tile_y, int wave, int lane, int& px, int& pyl) {
    px = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
    pyl = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
}

/* False for a whole wave when none of its pixels lies inside the frame's cull rectangle (DFrame::cull_*): its primary rays
 * cannot reach any instance, so the scene is never looked at — no instance / volume record loaded, no slab test.  Four out
 * of five waves of the benchmark frame.  Wave-uniform on purpose: a wave that straddles the rectangle marches all its rays. */
__device__ __forceinline__ bool wave_can_reach(const DCam& C, bool valid, int px, int py) {
    const int x0 = (int)(C.cull_lo & 0xffffu), y0 = (int)(C.cull_lo >> 16), x1 = (int)(C.cull_hi & 0xffffu), y1 = (int)(C.cull_hi >> 16);
    const bool inside = px >= x0 && px <= x1 && py >= y0 && py <= y1;
    return __ballot(valid && inside) != 0ull;
}

/* The frame of the launch this workgroup belongs to (blockIdx.y) and its camera record: one scalar load of 64 bytes at a
 * wave-uniform offset, from the kernarg segment or — launches of more than kMaxBlockFrames frames — from device memory. */
__device__ __forceinline__ DCam load_cam(const DBlock& B, int frame) {
    typedef unsigned u16v __attribute__((ext_vector_type(16)));
    u16v w;
    if (B.f.cams != nullptr) w = *reinterpret_cast<const u16v __attribute__((address_space(4)))*>((const __attribute__((address_space(4))) DCam*)B.f.cams + frame);
    else w = *reinterpret_cast<const u16v*>(&B.cam[frame]);
    DCam C;
    C.cam_o[0] = __uint_as_float(w[0]); C.cam_o[1] = __uint_as_float(w[1]); C.cam_o[2] = __uint_as_float(w[2]);
    C.r0[0] = __uint_as_float(w[3]); C.r0[1] = __uint_as_float(w[4]); C.r0[2] = __uint_as_float(w[5]);
    C.r1[0] = __uint_as_float(w[6]); C.r1[1] = __uint_as_float(w[7]); C.r1[2] = __uint_as_float(w[8]);
    C.r2[0] = __uint_as_float(w[9]); C.r2[1] = __uint_as_float(w[10]); C.r2[2] = __uint_as_float(w[11]);
    C.cx = __uint_as_float(w[12]); C.cy = __uint_as_float(w[13]);
    C.cull_lo = w[14]; C.cull_hi = w[15];
    return C;
}

/* The scene a frame of the launch renders.  Static launches: the kernarg's own (no copy).  DYN instantiations (vrt_block::scenes —
 * objects and lights that move from frame to frame, as in the reference's demo, RendererEngineInstance.cpp:111-130): the kernarg's
 * struct with the frame's light, counts and array pointers from its section of DFrame::dyn — one scalar load of 64 bytes at a
 * wave-uniform address plus four pointer additions; everything stays in scalar registers. */
template <bool DYN>
__device__ __forceinline__ const DFrame& frame_view(const DBlock& B, int frame, DFrame& Fd) {
    if constexpr (!DYN) {
        (void)frame;
        (void)Fd;
        return B.f;
    } else {
        typedef unsigned u16v __attribute__((ext_vector_type(16)));
        Fd = B.f;
        const char* sec = static_cast<const char*>(B.f.dyn) + (size_t)(unsigned)frame * kDynStride;
        const u16v w = *reinterpret_cast<const u16v __attribute__((address_space(4)))*>((const __attribute__((address_space(4))) char*)sec);
        Fd.light_dir[0] = __uint_as_float(w[0]);
        Fd.light_dir[1] = __uint_as_float(w[1]);
        Fd.light_dir[2] = __uint_as_float(w[2]);
        Fd.light_strength = __uint_as_float(w[3]);
        Fd.n_inst = (int)w[4];
        Fd.n_nodes = (int)w[5];
        Fd.n_point = (int)w[6];
        Fd.n_spot = (int)w[7];
        Fd.vol0 = B.f.vols + (int)w[8];
        Fd.inst = reinterpret_cast<const DInstance*>(sec + kDynInstOff);
        Fd.nodes = reinterpret_cast<const DBvhNode*>(sec + kDynNodesOff);
        Fd.point = reinterpret_cast<const DPointLight*>(sec + kDynPointOff);
        Fd.spot = reinterpret_cast<const DSpotLight*>(sec + kDynSpotOff);
        return Fd;
    }
}

/* Camera ray of pixel (px,py) (Ray.hlsli:36-48, then normalised). */
__device__ __forceinline__ void camera_ray(const DFrame& F, const DCam& C, int px, int py, F3& o, F3& d) {
    float sx = (((float)px + 0.5f) * F.inv_w) * 2.0f - 1.0f; /* 1/width, 1/height from the host (oracle: the same two products) */
    float sy = (((float)py + 0.5f) * F.inv_h) * 2.0f - 1.0f;
    float tx = sx * C.cx;
    float ty = (-sy) * C.cy;
    d = normalize3(f3((tx * C.r0[0] + ty * C.r1[0]) - C.r2[0], (tx * C.r0[1] + ty * C.r1[1]) - C.r2[1],
                      (tx * C.r0[2] + ty * C.r1[2]) - C.r2[2]));
    o = f3(C.cam_o[0], C.cam_o[1], C.cam_o[2]);
}

/* Length of the camera direction of pixel (px,py) BEFORE its normalisation: what the reference's WorldRayDirection() is long
   (VRT_FLAG_REFERENCE_VIEW_VECTOR; oracle: camera_ray's len).  Recomputed behind the march rather than kept in a register across it. */
__device__ __forceinline__ float camera_len(const DFrame& F, const DCam& C, int px, int py) {
    float sx = (((float)px + 0.5f) * F.inv_w) * 2.0f - 1.0f;
    float sy = (((float)py + 0.5f) * F.inv_h) * 2.0f - 1.0f;
    float tx = sx * C.cx;
    float ty = (-sy) * C.cy;
    const F3 a = f3((tx * C.r0[0] + ty * C.r1[0]) - C.r2[0], (tx * C.r0[1] + ty * C.r1[1]) - C.r2[1], (tx * C.r0[2] + ty * C.r1[2]) - C.r2[2]);
    return sqrtf(dot3(a, a));
}

struct Counters {
    unsigned n_primary = 0, n_shadow = 0, n_bounce = 0, s_primary = 0, s_shadow = 0, n_hits = 0;
};

/*
 * Statistics (algorithmic-byte accounting): wave shuffle-reduce, then one 32-byte record
 * per WAVE with plain stores.  No atomics (6 same-address atomics per wave serialise at ~12 ns each
 * at the memory side and cost more than the march itself) and no workgroup barrier (it would pin
 * the three fast waves of a tile until its slowest wave retires).
 */
/* UNIT: every ray / hit counter of a lane is 0 or 1 (the lean kernels): one ballot + popcount each on the scalar unit instead of
   a 12-instruction shuffle reduction. */
template <bool DIAG, bool UNIT = false, bool ADD = false /* a later pass of a launch adds to the record an earlier one wrote */>
__device__ __forceinline__ void write_records(const DFrame& F, int frame, int b, int wave, int lane, Counters k, const DiagAcc& dg,
                                              unsigned long long t_start) {
    const size_t frame_words = (size_t)(unsigned)frame * F.stats_stride; /* this frame's records within the launch's buffers */
    const unsigned ex_lane = k.n_hits >> kExhaustedShift;
    const unsigned exhausted = __ballot(ex_lane != 0u) == 0ull ? 0u : wave_sum(ex_lane); /* practically never set */
    k.n_hits &= kExhaustedOne - 1u;
    const unsigned s_primary_lane = k.s_primary, s_shadow_lane = k.s_shadow;
    if constexpr (UNIT) {
        k.n_primary = (unsigned)__builtin_popcountll(__ballot(k.n_primary != 0u));
        k.n_shadow = (unsigned)__builtin_popcountll(__ballot(k.n_shadow != 0u));
        k.n_bounce = 0u;
        k.n_hits = (unsigned)__builtin_popcountll(__ballot(k.n_hits != 0u));
        k.s_primary = __ballot(k.s_primary != 0u) == 0ull ? 0u : wave_sum(k.s_primary); /* sky waves take no sample */
        k.s_shadow = __ballot(k.s_shadow != 0u) == 0ull ? 0u : wave_sum(k.s_shadow);
    } else {
        k.n_primary = wave_sum(k.n_primary);
        k.n_shadow = wave_sum(k.n_shadow);
        k.n_bounce = wave_sum(k.n_bounce);
        k.s_primary = wave_sum(k.s_primary);
        k.s_shadow = wave_sum(k.s_shadow);
        k.n_hits = wave_sum(k.n_hits);
    }
    if (F.stats != nullptr && lane < 8) {
        unsigned v = lane == 0 ? k.n_primary : lane == 1 ? k.n_shadow : lane == 2 ? k.n_bounce : lane == 3 ? k.s_primary
                   : lane == 4 ? k.s_shadow : lane == 5 ? k.n_hits : lane == 6 ? exhausted : 0u;
        unsigned* const rec = F.stats + (frame_words + ((size_t)b * 4 + wave) * kStatRecord + lane);
        if constexpr (ADD) v += *rec;
        *rec = v;
    }
    if constexpr (DIAG) {
        /* diagnostic timeline record: where and when this wave ran and where its march cycles went
           (never in the production kernel; stamp values leave only through this buffer).  The
           accumulators are per lane: report the lane with the longest chain of dependent samples,
           i.e. the wave's critical path */
        unsigned max_iter = s_primary_lane + s_shadow_lane;
        unsigned long long key = ((unsigned long long)dg.iters << 8) | (unsigned)lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            unsigned other = __shfl_xor(max_iter, o);
            max_iter = other > max_iter ? other : max_iter;
            unsigned long long ok = __shfl_xor(key, o);
            key = ok > key ? ok : key;
        }
        const int src = (int)(key & 0xff);
        const unsigned d_mem = __shfl((unsigned)dg.mem, src);
        const unsigned d_loop = __shfl((unsigned)dg.loop, src);
        const unsigned d_iters = __shfl(dg.iters, src);
        const unsigned d_fetches = __shfl(dg.fetches, src);
        if (F.diag_buf != nullptr && lane < kDiagRecord) {
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
            const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4); /* HW_REG_HW_ID */
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   /* HW_REG_XCC_ID[3:0] */
            unsigned v = 0;
            switch (lane) {
                case 0: v = (unsigned)t_start; break;
                case 1: v = (unsigned)t_end; break;
                case 2: v = d_fetches; break;
                case 3: v = xcc | (hw_id << 4); break;
                case 4: v = max_iter; break;
                case 5: v = d_mem; break;
                case 6: v = d_loop; break;
                case 7: v = d_iters; break;
                default: break;
            }
            F.diag_buf[frame_words + ((size_t)b * 4 + wave) * kDiagRecord + lane] = v;
        }
    }
}

/* Directional-light shading of a hit (NoTex closest hit, Raytracing_NoTex.hlsl:41-94). */
__device__ __forceinline__ F3 shade_hit(const DFrame& F, const DVolume* __restrict__ V, F3 d, F3 n, bool shadowed) {
    F3 albedo = f3(V->tint[0], V->tint[1], V->tint[2]);
    if (F.unlit) return albedo;
    F3 color = f3(0.0f, 0.0f, 0.0f); /* SHADOW_BRIGHTNESS */
    if (!shadowed) {
        F3 wo = f3(-d.x, -d.y, -d.z);
        F3 ld = f3(F.light_dir[0], F.light_dir[1], F.light_dir[2]);
        F3 Li = f3(F.light_strength, F.light_strength, F.light_strength);
        color = color + radiance(Li, ld, wo, n, albedo, V->roughness, V->metallic, V->k);
    }
    return color;
}
/* ... with the surface given (constant material textures folded in) and the reference's view vector wo = -vs d. */
__device__ __forceinline__ F3 shade_hit_surface(const DFrame& F, const DVolume* __restrict__ V, F3 d, float vs, F3 n, F3 albedo, float rough, float metal,
                                                bool shadowed) {
    if (F.unlit) return albedo;
    F3 color = f3(0.0f, 0.0f, 0.0f);
    if (!shadowed) {
        F3 wo = f3(-d.x * vs, -d.y * vs, -d.z * vs);
        F3 ld = f3(F.light_dir[0], F.light_dir[1], F.light_dir[2]);
        F3 Li = f3(F.light_strength, F.light_strength, F.light_strength);
        color = color + radiance(Li, ld, wo, n, albedo, rough, metal, V->k);
    }
    return color;
}

/* Shadow-ray origin: the hit point pulled 0.1 back along the ray (Raytracing.hlsl:51-52); 0.2 in the Cube
   modes (Raytracing_Cube.hlsl:52). */
__device__ __forceinline__ F3 shadow_origin(const DFrame& F, F3 o, F3 d, float t_hit) {
    F3 hp = f3(__builtin_fmaf(d.x, t_hit, o.x), __builtin_fmaf(d.y, t_hit, o.y), __builtin_fmaf(d.z, t_hit, o.z));
    return f3(hp.x - d.x * F.back, hp.y - d.y * F.back, hp.z - d.z * F.back);
}
/* ... vs x as far back: the reference's offset is in units of its un-normalised ray direction (VRT_FLAG_REFERENCE_VIEW_VECTOR) */
__device__ __forceinline__ F3 shadow_origin(const DFrame& F, F3 o, F3 d, float t_hit, float vs) {
    F3 hp = f3(__builtin_fmaf(d.x, t_hit, o.x), __builtin_fmaf(d.y, t_hit, o.y), __builtin_fmaf(d.z, t_hit, o.z));
    const float back = F.back * vs;
    return f3(hp.x - d.x * back, hp.y - d.y * back, hp.z - d.z * back);
}

/* UNORM8 of a tone-mapped channel in [0,1]: round to nearest, the D3D float→UNORM rule (the reference's
   render target is 8-bit UNORM, DXConstants.cpp:21).  Plain mul + add, no fma, so numpy restates it exactly. */
__device__ __forceinline__ unsigned unorm8(float c) {
    return (unsigned)(fminf(c, 1.0f) * 255.0f + 0.5f);
}

__device__ __forceinline__ void store_pixel(const DFrame& F, int frame, unsigned pix /* pyl * width + px */, F3 color) {
    char* const out = reinterpret_cast<char*>(F.out) + (size_t)(unsigned)frame * F.frame_stride;
    const float r = tonemap(color.x), g = tonemap(color.y), b = tonemap(color.z);
    /* streaming stores: a frame writes as many bytes as an XCD's whole L2 holds; they must not push the bricks out */
    if (F.rgba8) {
        /* R8G8B8A8, or (2) the reference's own back-buffer order B8G8R8A8 (DXConstants.cpp:21): which channel goes low, which third */
        const bool bgra = F.rgba8 == 2;
        __builtin_nontemporal_store(unorm8(bgra ? b : r) | unorm8(g) << 8 | unorm8(bgra ? r : b) << 16 | 0xff000000u,
                                    reinterpret_cast<unsigned*>(out) + pix);
    } else {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = {r, g, b, 1.0f};
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out) + pix);
    }
}

/* Frame row of local row pyl of this launch's tile (contiguous rows, or interleaved strips). */
__device__ __forceinline__ int frame_row(const DFrame& F, int pyl) {
    if (F.strip_rows > 0) {
        const int s = pyl / F.strip_rows;
        return (s * F.strip_stride + F.strip_first) * F.strip_rows + (pyl - s * F.strip_rows);
    }
    return F.row0 + pyl;
}

/* ---- tri-planar material textures (SH/Include/Textures.hlsli:16-59, Quaternion.hlsli:18-82) ----------
 * Restated line by line in the oracle (textured_surface); everything here feeds discrete decisions
 * (texel choice, the bounce direction), so divisions and square roots are the correctly rounded ones. */

/* Geometry sampler: point filter, wrap addressing (RDXScene.cpp:262-270). */
__device__ __forceinline__ F3 tex_point_wrap(const uint8_t* __restrict__ px, int W, int H, float u, float v) {
    const float fu = u - floorf(u), fv = v - floorf(v);
    int x = (int)(fu * (float)W), y = (int)(fv * (float)H);
    x = x > W - 1 ? W - 1 : (x < 0 ? 0 : x);
    y = y > H - 1 ? H - 1 : (y < 0 ? 0 : y);
    const unsigned t = *(const unsigned __attribute__((address_space(1)))*)(px + ((size_t)y * (size_t)W + (size_t)x) * 4);
    return f3((float)(t & 0xffu) / 255.0f, (float)((t >> 8) & 0xffu) / 255.0f, (float)((t >> 16) & 0xffu) / 255.0f);
}

/* CONST_ONLY: every bound texture of the launch is a 1x1 image (DVolume::tex_const_mask), e.g. the reference's default normal texel:
   no fetch is compiled in.  A constant slot's three planar samples are the one texel; the arithmetic behind them is the fetch path's. */
template <bool CONST_ONLY>
__device__ __forceinline__ F3 tri_sample(const DVolume* __restrict__ V, int which, F3 op, F3 blend, bool as_normal) {
    F3 tx, ty, tz;
    if (CONST_ONLY || ((V->tex_const_mask >> which) & 1) != 0) {
        tx = ty = tz = f3(V->tex_const[3 * which], V->tex_const[3 * which + 1], V->tex_const[3 * which + 2]);
    } else {
        const uint8_t* px = V->tex_px[which];
        const int W = V->tex_w[which], H = V->tex_h[which];
        const float su = V->tex_scale[0], sv = V->tex_scale[1];
        tx = tex_point_wrap(px, W, H, op.z / su, op.y / sv);
        ty = tex_point_wrap(px, W, H, op.x / su, op.z / sv);
        tz = tex_point_wrap(px, W, H, op.x / su, op.y / sv);
    }
    if (as_normal) {
        tx = f3(tx.x * 2.0f - 1.0f, tx.y * 2.0f - 1.0f, tx.z * 2.0f - 1.0f);
        ty = f3(ty.x * 2.0f - 1.0f, ty.y * 2.0f - 1.0f, ty.z * 2.0f - 1.0f);
        tz = f3(tz.x * 2.0f - 1.0f, tz.y * 2.0f - 1.0f, tz.z * 2.0f - 1.0f);
    }
    return f3((tx.x * blend.x + ty.x * blend.y) + tz.x * blend.z, (tx.y * blend.x + ty.y * blend.y) + tz.y * blend.z,
              (tx.z * blend.x + ty.z * blend.y) + tz.z * blend.z);
}

struct Q4 {
    float x, y, z, w;
};
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
    Q4 r;
    r.x = (b.x * a.w + a.x * b.w) + (a.y * b.z - a.z * b.y);
    r.y = (b.y * a.w + a.y * b.w) + (a.z * b.x - a.x * b.z);
    r.z = (b.z * a.w + a.z * b.w) + (a.x * b.y - a.y * b.x);
    r.w = a.w * b.w - ((a.x * b.x + a.y * b.y) + a.z * b.z);
    return r;
}
__device__ __forceinline__ Q4 quat_from_x(F3 n) {
    const float d = n.x;
    Q4 q;
    if (d < -0.999999f) {
        q.x = 0.0f; q.y = 0.0f; q.z = -1.0f; q.w = -4.371139e-08f;
    } else if (d > 0.999999f) {
        q.x = 0.0f; q.y = 0.0f; q.z = 0.0f; q.w = 1.0f;
    } else {
        q.x = 0.0f; q.y = -n.z; q.z = n.y; q.w = 1.0f + d;
        const float inv = 1.0f / sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
        q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv;
    }
    return q;
}
__device__ __forceinline__ F3 rotate_vector(F3 v, Q4 r) {
    Q4 rc = {-r.x, -r.y, -r.z, r.w};
    Q4 vq = {v.x, v.y, v.z, 0.0f};
    const Q4 t = qmul(r, qmul(vq, rc));
    return f3(t.x, t.y, t.z);
}

/* Material at a hit in the textured modes (unbound slots are exact identities). */
template <bool CONST_ONLY = false>
__device__ __forceinline__ void textured_surface(const DVolume* __restrict__ V, const DInstance* __restrict__ I, F3 hit_world,
                                                 F3& albedo, F3& n, float& rough, float& metal) {
    const F3 op = mul33(I->w2o, f3(hit_world.x - I->pos[0], hit_world.y - I->pos[1], hit_world.z - I->pos[2]));
    const F3 no = mul33(I->w2o, n);
    const F3 an = f3(fabsf(no.x), fabsf(no.y), fabsf(no.z));
    const float sum = (an.x + an.y) + an.z;
    const F3 blend = f3(an.x / sum, an.y / sum, an.z / sum);
    if (V->tex_px[0] != nullptr) {
        const F3 t = tri_sample<CONST_ONLY>(V, 0, op, blend, false);
        albedo = f3(albedo.x * t.x, albedo.y * t.y, albedo.z * t.z);
    }
    if (V->tex_px[2] != nullptr) {
        const F3 t = tri_sample<CONST_ONLY>(V, 2, op, blend, false);
        rough = minf_(maxf_(V->roughness_raw * t.x, 0.0f), 1.0f);
        metal = minf_(maxf_(V->metallic_raw * t.y, 0.0f), 1.0f);
    }
    if (V->tex_px[1] != nullptr) {
        F3 t = tri_sample<CONST_ONLY>(V, 1, op, blend, true);
        t = normalize3(t);
        n = mul33(I->o2w, rotate_vector(f3(t.z, t.x, t.y), quat_from_x(no)));
    }
}

/* (Round 3 experiment, removed: ONE wave rendering a whole 16x16 sky tile — four pixels per lane — while the tile's other three
 * waves end at once.  Letting three out of four sky waves end at once changes nothing (41.1 against 41.0 us per frame: the sky
 * waves only fill wave slots the marching waves leave empty), and the four-pixel sky wave made the frame 20 % slower:
 * profiles/r03_sky_tile_and_occupancy_experiments.txt.) */
template <int PATH, bool SINGLE, bool DIAG, bool DYN = false,
          bool REF = false /* "what the DXR backend renders": constant (1x1) material textures in the textured modes, the reference's view
                              vector and boundary texels (DFrame::textured / view_vec / zero_outside).  A separate instantiation, so that the
                              plain kernel's code and registers stay what they are; launched when a frame needs any of the three */>
/* The multi-instance (BVH) instantiations on brick / cell-record paths are asked to fit 7 waves per SIMD (72 VGPRs instead of the 78 the
 * register allocator settles for; 3 registers and 10 scalars spilled outside the march loop): config 5 68.6 -> 70.9 Grays/s.  The
 * single-instance ones sit at the hardware's 8 waves per SIMD anyway. */
#ifndef VRT_BVH_WAVES
#define VRT_BVH_WAVES 7
#endif
__global__ __launch_bounds__(kMarchThreads) __attribute__((amdgpu_waves_per_eu((!SINGLE && !DIAG && PATH != VRT_PATH_DENSE) ? VRT_BVH_WAVES : (REF && SINGLE) ? 8 : 1))) void march_kernel(const DBlock B) {
    unsigned long long t_start = 0;
    if constexpr (DIAG) t_start = __builtin_amdgcn_s_memrealtime(); /* 100 MHz; diagnostic build only */
    const int frame = (int)blockIdx.y;
    DFrame Fd;
    const DFrame& F = frame_view<DYN>(B, frame, Fd);
    const DCam C = load_cam(B, frame);
    int b, wave;
    block_and_wave(b, wave);
    int tile_x, tile_y;
    tile_of_block(F, b, (int)gridDim.x / kMarchGridMul, tile_x, tile_y);
    const int lane = (int)threadIdx.x & 63;
    int px, pyl;
    pixel_of_lane(tile_x, tile_y, wave, lane, px, pyl);
    const int py = frame_row(F, pyl);
    const bool valid = tile_x < F.tiles_x && tile_y < F.tiles_y && px < F.width && pyl < F.rows && py < F.height;

    Counters k;
    DiagAcc dg;
    const bool reach = wave_can_reach(C, valid, px, py);

    if (valid) {
        F3 o, d;
        camera_ray(F, C, px, py, o, d);
        k.n_primary = 1;
        float t_hit = 0.0f;
        int inst = 0;
        F3 n = f3(0.0f, 0.0f, 0.0f);
        F3 color;
        /* the sky texel is asked for before the march (one register across it): four out of five waves of a frame see only
           sky, and for them it is the last link of a chain of dependent loads (kernarg -> instance / volume -> texel -> store) */
        const unsigned sky = env_fetch(F.env, F.env_size, d);
        /* the normal's length is the correctly rounded one: its dot product with the light decides whether a shadow ray is cast */
        if (reach && trace_closest<PATH, SINGLE, DIAG, 2, REF>(F, o, d, 10000.0f, 0.0f, t_hit, inst, n, k.s_primary, k.n_hits, &dg)) {
            k.n_hits += 1; /* (its upper bits count exhausted marches) */
            bool shadowed = false;
            const F3 ld = f3(F.light_dir[0], F.light_dir[1], F.light_dir[2]);
            if constexpr (REF) {
                const DVolume* V = SINGLE ? F.vol0 : F.vols + F.inst[inst].slot;
                /* the hit's surface first (a constant normal texel tilts the normal the shadow decision looks at), then as below */
                F3 albedo = f3(V->tint[0], V->tint[1], V->tint[2]);
                float rough = V->roughness, metal = V->metallic;
                if (F.textured) {
                    const F3 hp = f3(__builtin_fmaf(d.x, t_hit, o.x), __builtin_fmaf(d.y, t_hit, o.y), __builtin_fmaf(d.z, t_hit, o.z));
                    textured_surface<true>(V, SINGLE ? F.inst : F.inst + inst, hp, albedo, n, rough, metal);
                }
                if (F.shadow && !F.unlit && dot3(n, ld) > 0.0f) {
                    k.n_shadow = 1;
                    const float vs = F.view_vec ? camera_len(F, C, px, py) : 1.0f;
                    shadowed = trace_any<PATH, SINGLE, DIAG, true>(F, shadow_origin(F, o, d, t_hit, vs), ld, 5000.0f, t_hit, k.s_shadow, k.n_hits, &dg);
                }
                const float vs = F.view_vec ? camera_len(F, C, px, py) : 1.0f;
                color = shade_hit_surface(F, V, d, vs, n, albedo, rough, metal, shadowed);
            } else {
            /* A surface facing away from the light gets a contribution <= 0 from it, blocked or not, and this kernel has no
               other term: the tone-map clamps the pixel to 0 either way, so that shadow ray is not cast (oracle: same rule) */
            if (F.shadow && !F.unlit && dot3(n, ld) > 0.0f) {
                k.n_shadow = 1;
                shadowed = trace_any<PATH, SINGLE, DIAG, true>(F, shadow_origin(F, o, d, t_hit), ld, 5000.0f, t_hit, k.s_shadow, k.n_hits, &dg);
            }
            color = shade_hit(F, SINGLE ? F.vol0 : F.vols + F.inst[inst].slot, d, n, shadowed);
            }
        } else {
            color = env_decode(sky);
        }
        /* (The pixel's offset derived again from the lane id (mbcnt) behind the march, so that no register holds pixel coordinates
           across it, gave 60 instead of 64 VGPRs and measured 5 % SLOWER: profiles/r03_ab_fused_variants.txt.) */
        store_pixel(F, frame, (unsigned)pyl * (unsigned)F.width + (unsigned)px, color);
    }
    write_records<DIAG, true>(F, frame, b, wave, lane, k, dg, t_start);
}

/*
 * Full closest hit (SH/Raytracing_NoTex.hlsl:41-139): directional + ≤5 point + ≤5 spot lights, each
 * with its own shadow ray, and the mirror bounce of smooth materials (roughness < 0.3) down to
 * MAX_RAY_RECURSION_DEPTH = 3 levels.  The reference recurses (TraceRay inside the closest-hit
 * shader); here the ≤3 levels are a loop that records, per level, the direct light and the BRDF
 * factor of the bounce, and folds them back to front:  colour_L = direct_L + (BRDF_L · fade(colour_L+1)) · n·wi.
 * At level 3 the shadow rays are skipped and count as unoccluded (TraceShadowRay's recursion guard,
 * SH/Include/Ray.hlsli:83-86).
 */
constexpr int kMaxDepth = 3;

#ifndef VRT_FULL_WAVES
#define VRT_FULL_WAVES 1
#endif
/* The record of this wave's lane in the three-pass form (DFrame::hit_rec / hit_aux), and the wave's word of hit_mask. */
__device__ __forceinline__ size_t pass_record(const DFrame& F, int frame, int b, int wave, int lane) {
    return (size_t)(unsigned)frame * F.rec_stride + (size_t)(((unsigned)b * 4u + (unsigned)wave) * 64u + (unsigned)lane);
}
__device__ __forceinline__ size_t pass_wave(const DFrame& F, int frame, int b, int wave) {
    return ((size_t)(unsigned)frame * F.rec_stride >> 6) + (size_t)((unsigned)b * 4u + (unsigned)wave);
}
constexpr unsigned kShadowBitDir = 1u << 16, kShadowBitPoint = 2u << 16, kShadowBitSpot = 64u << 16;

/* The direct light of a hit whose shadow rays have been cast already (their verdicts in `bits`): the three light loops of
 * march_kernel_full below, operation for operation, with a bit test where they march. */
__device__ __forceinline__ F3 direct_light_from_bits(const DFrame& F, F3 so, F3 wo, F3 n, F3 albedo, float rough, float metal, float kk,
                                                     bool shadows, bool bounce, unsigned bits) {
    F3 sum = f3(0.0f, 0.0f, 0.0f);
    {
        const F3 ld = f3(F.light_dir[0], F.light_dir[1], F.light_dir[2]);
        const bool lone_backfacing = F.n_point == 0 && F.n_spot == 0 && !bounce && !(dot3(n, ld) > 0.0f);
        const bool sh = shadows && !lone_backfacing && (bits & kShadowBitDir) != 0u;
        if (!sh) sum = sum + radiance(f3(F.light_strength, F.light_strength, F.light_strength), ld, wo, n, albedo, rough, metal, kk);
    }
    for (int i = 0; i < F.n_point; i++) {
        const DPointLight L = F.point[i];
        const F3 dl = f3(L.pos[0] - so.x, L.pos[1] - so.y, L.pos[2] - so.z);
        const float dist = sqrtf(dot3(dl, dl));
        const float inten = L.intensity / ((1.0f + L.att_linear * dist) + (L.att_exp * dist) * dist);
        if (inten > 0.005f) {
            const F3 ld = dl * (1.0f / dist);
            const bool sh = shadows && (bits & (kShadowBitPoint << i)) != 0u;
            if (!sh) sum = sum + radiance(f3(L.color[0] * inten, L.color[1] * inten, L.color[2] * inten), ld, wo, n, albedo, rough, metal, kk);
        }
    }
    for (int i = 0; i < F.n_spot; i++) {
        const DSpotLight L = F.spot[i];
        const F3 dl = f3(L.pos[0] - so.x, L.pos[1] - so.y, L.pos[2] - so.z);
        const float dist = sqrtf(dot3(dl, dl));
        const F3 sd = f3(-dl.x, -dl.y, -dl.z) * (1.0f / dist);
        const float cs = dot3(f3(L.fwd[0], L.fwd[1], L.fwd[2]), sd);
        float inten = 0.0f;
        if (cs >= 0.0f && cs > L.cos_angle) {
            const float delta = (cs - L.cos_angle) / (L.cos_falloff - L.cos_angle);
            const float base = L.intensity * minf_(delta, 1.0f);
            inten = base / ((1.0f + L.att_linear * dist) + (L.att_exp * dist) * dist);
        }
        if (inten > 0.01f) {
            const F3 ld = dl * (1.0f / dist);
            const bool sh = shadows && (bits & (kShadowBitSpot << i)) != 0u;
            if (!sh) sum = sum + radiance(f3(L.color[0] * inten, L.color[1] * inten, L.color[2] * inten), ld, wo, n, albedo, rough, metal, kk);
        }
    }
    return sum;
}

/* SHADE_PASS: third pass of the three-pass form (below): the camera ray's hit comes from the first pass's record, the verdicts of its
 * light shadow rays from the second's bits; everything behind a mirror bounce is traced here as in the one-kernel form. */
template <int PATH, bool SINGLE, bool SHADE_PASS = false, bool DYN = false>
__global__ __launch_bounds__(kMarchThreads) __attribute__((amdgpu_waves_per_eu(VRT_FULL_WAVES))) void march_kernel_full(const DBlock B) {
    const int frame = (int)blockIdx.y;
    DFrame Fd;
    const DFrame& F = frame_view<DYN>(B, frame, Fd);
    int b, wave;
    block_and_wave(b, wave);
    const int lane = (int)threadIdx.x & 63;
    bool rec_hit = false;
    if constexpr (SHADE_PASS) {
        const unsigned long long m = F.hit_mask[pass_wave(F, frame, b, wave)];
        if (m == 0ull) return; /* sky: the first pass stored these pixels and their records */
        rec_hit = ((m >> lane) & 1ull) != 0ull;
    }
    const DCam C = load_cam(B, frame);
    int tile_x, tile_y;
    tile_of_block(F, b, (int)gridDim.x / kMarchGridMul, tile_x, tile_y);
    int px, pyl;
    pixel_of_lane(tile_x, tile_y, wave, lane, px, pyl);
    const int py = frame_row(F, pyl);
    const bool valid = SHADE_PASS ? rec_hit : (tile_x < F.tiles_x && tile_y < F.tiles_y && px < F.width && pyl < F.rows && py < F.height);

    Counters k;
    DiagAcc dg;
    const bool reach = SHADE_PASS ? true : wave_can_reach(C, valid, px, py);
    if (valid) {
        F3 o, d;
        camera_ray(F, C, px, py, o, d);
        if constexpr (!SHADE_PASS) k.n_primary = 1;
        F3 direct[kMaxDepth - 1], brdf[kMaxDepth - 1];
        float ndwi[kMaxDepth - 1], fade[kMaxDepth - 1];
        int pending = 0;
        float t_base = 0.0f;
        F3 color = f3(0.0f, 0.0f, 0.0f);
        unsigned shadow_bits = 0u;
#pragma unroll 1
        for (int level = 1; level <= kMaxDepth; level++) {
            float t_hit = 0.0f;
            int inst = 0;
            F3 n = f3(0.0f, 0.0f, 0.0f);
            if (SHADE_PASS && level == 1) {
                const size_t r = pass_record(F, frame, b, wave, lane);
                const HitRecord h = F.hit_rec[r];
                const unsigned aux = F.hit_aux[r];
                t_hit = h.t;
                n = f3(h.nx, h.ny, h.nz);
                inst = (int)(aux & 0xffffu);
                shadow_bits = aux;
            } else {
                if ((level == 1 && !reach) || !trace_closest<PATH, SINGLE, false, 2, true>(F, o, d, 10000.0f, t_base, t_hit, inst, n, k.s_primary, k.n_hits)) {
                    color = env_lookup(F.env, F.env_size, d);
                    break;
                }
                k.n_hits++;
            }
            const bool from_bits = SHADE_PASS && level == 1; /* this level's shadow rays were cast (and counted) by the second pass */
            const DVolume* V = SINGLE ? F.vol0 : F.vols + F.inst[inst].slot;
            F3 albedo = f3(V->tint[0], V->tint[1], V->tint[2]);
            float rough = V->roughness, metal = V->metallic;
            if (F.textured) {
                const F3 hp = f3(__builtin_fmaf(d.x, t_hit, o.x), __builtin_fmaf(d.y, t_hit, o.y), __builtin_fmaf(d.z, t_hit, o.z));
                textured_surface(V, SINGLE ? F.inst : F.inst + inst, hp, albedo, n, rough, metal);
            }
            if (F.unlit) {
                color = albedo;
                break;
            }
            const float kk = V->k;
            /* the camera ray is the one ray whose direction the reference leaves un-normalised (VRT_FLAG_REFERENCE_VIEW_VECTOR) */
            const float vs = (level == 1 && F.view_vec) ? camera_len(F, C, px, py) : 1.0f;
            const F3 so = shadow_origin(F, o, d, t_hit, vs);
            const F3 wo = f3(-d.x * vs, -d.y * vs, -d.z * vs);
            const float tb = t_base + t_hit;
            const bool shadows = F.shadow && level < kMaxDepth;
            const bool bounce = rough < 0.3f && level <= F.max_bounces && level < kMaxDepth;
            F3 sum = f3(0.0f, 0.0f, 0.0f);
            if (from_bits) {
                sum = direct_light_from_bits(F, so, wo, n, albedo, rough, metal, kk, shadows, bounce, shadow_bits);
            } else {
            {   /* directional light */
                const F3 ld = f3(F.light_dir[0], F.light_dir[1], F.light_dir[2]);
                bool sh = false;
                /* not cast where it cannot change the pixel: the surface faces away from the light (contribution <= 0
                   blocked or not) and nothing else is added at this hit (no point / spot light, no mirror bounce) */
                const bool lone_backfacing = F.n_point == 0 && F.n_spot == 0 && !bounce && !(dot3(n, ld) > 0.0f);
                if (shadows && !lone_backfacing) {
                    k.n_shadow++;
                    sh = trace_any<PATH, SINGLE, false, true>(F, so, ld, 5000.0f, tb, k.s_shadow, k.n_hits);
                }
                if (!sh) sum = sum + radiance(f3(F.light_strength, F.light_strength, F.light_strength), ld, wo, n, albedo, rough, metal, kk);
            }
            for (int i = 0; i < F.n_point; i++) { /* ComputePointLightIntensity, Lighting.hlsli:17-20 */
                const DPointLight L = F.point[i];
                const F3 dl = f3(L.pos[0] - so.x, L.pos[1] - so.y, L.pos[2] - so.z);
                const float dist = sqrtf(dot3(dl, dl));
                const float inten = L.intensity / ((1.0f + L.att_linear * dist) + (L.att_exp * dist) * dist);
                if (inten > 0.005f) {
                    const F3 ld = dl * (1.0f / dist);
                    bool sh = false;
                    if (shadows) {
                        k.n_shadow++;
                        sh = trace_any<PATH, SINGLE>(F, so, ld, dist, tb, k.s_shadow, k.n_hits);
                    }
                    if (!sh) sum = sum + radiance(f3(L.color[0] * inten, L.color[1] * inten, L.color[2] * inten), ld, wo, n, albedo, rough, metal, kk);
                }
            }
            for (int i = 0; i < F.n_spot; i++) { /* ComputeSpotLightIntensity, Lighting.hlsli:30-48 */
                const DSpotLight L = F.spot[i];
                const F3 dl = f3(L.pos[0] - so.x, L.pos[1] - so.y, L.pos[2] - so.z);
                const float dist = sqrtf(dot3(dl, dl));
                const F3 sd = f3(-dl.x, -dl.y, -dl.z) * (1.0f / dist);
                const float cs = dot3(f3(L.fwd[0], L.fwd[1], L.fwd[2]), sd);
                float inten = 0.0f;
                if (cs >= 0.0f && cs > L.cos_angle) {
                    const float delta = (cs - L.cos_angle) / (L.cos_falloff - L.cos_angle);
                    const float base = L.intensity * minf_(delta, 1.0f);
                    inten = base / ((1.0f + L.att_linear * dist) + (L.att_exp * dist) * dist);
                }
                if (inten > 0.01f) {
                    const F3 ld = dl * (1.0f / dist);
                    bool sh = false;
                    if (shadows) {
                        k.n_shadow++;
                        sh = trace_any<PATH, SINGLE>(F, so, ld, dist, tb, k.s_shadow, k.n_hits);
                    }
                    if (!sh) sum = sum + radiance(f3(L.color[0] * inten, L.color[1] * inten, L.color[2] * inten), ld, wo, n, albedo, rough, metal, kk);
                }
            }
            }
            if (bounce) {
                /* mirror bounce: continue along the reflected ray, fold this level in afterwards */
                const float dn = dot3(d, n);
                const F3 rd = normalize3(f3(d.x - (2.0f * dn) * n.x, d.y - (2.0f * dn) * n.y, d.z - (2.0f * dn) * n.z));
                direct[pending] = sum;
                brdf[pending] = brdf_eval(rd, wo, n, albedo, rough, metal, kk);
                ndwi[pending] = dot3(n, rd);
                fade[pending] = rough * 2.2f;
                pending++;
                k.n_bounce++;
                o = so;
                d = rd;
                t_base = tb;
                color = f3(0.0f, 0.0f, 0.0f); /* level kMaxDepth+1 would return black (Ray.hlsli:62-65) */
                continue;
            }
            color = sum;
            break;
        }
        for (int j = pending - 1; j >= 0; j--) {
            const F3 rc = f3(maxf_(0.0f, color.x + (0.0f - color.x) * fade[j]), maxf_(0.0f, color.y + (0.0f - color.y) * fade[j]),
                             maxf_(0.0f, color.z + (0.0f - color.z) * fade[j]));
            const F3 r = f3((brdf[j].x * rc.x) * ndwi[j], (brdf[j].y * rc.y) * ndwi[j], (brdf[j].z * rc.z) * ndwi[j]);
            color = direct[j] + r;
        }
        store_pixel(F, frame, (unsigned)pyl * (unsigned)F.width + (unsigned)px, color);
    }
    write_records<false, false, SHADE_PASS>(F, frame, b, wave, lane, k, dg, 0ull);
}

/*
 * The full closest hit in passes — what vrt_render_block launches for a block of frames (a lone frame keeps the one kernel
 * above: every further launch pays a launch's latency-bound tail again).  The one-kernel form holds a level's whole shading state
 * across every march (109-123 VGPRs: 4 waves per SIMD, and the march lives on occupancy,
 * profiles/r03_sky_tile_and_occupancy_experiments.txt); here every march of the first level — the camera rays and the shadow rays
 * of their hits, nearly all the rays of a frame — runs in a kernel that keeps almost nothing else alive (58 VGPRs, 8 waves):
 *   1. primary_pass_kernel: the lean kernel's camera-ray march; a miss stores its sky pixel, a hit writes {normal, t} and its
 *      instance (hit_rec / hit_aux) and the wave its lane mask (hit_mask);
 *   2. light_pass_kernel: per hit, one any-hit march per light that the one-kernel form would cast, in its order and by its
 *      rules, the verdicts collected as bits; then the hit's shading from the record and the bits, and the pixel — unless the
 *      material mirrors: that lane's bits go to hit_aux, and the wave's mask becomes the mask of its bouncing lanes;
 *   3. march_kernel_full<.., SHADE_PASS> (only frames that can bounce): level 1 from the record and the bits, everything behind
 *      the mirror as in the one-kernel form.
 * Waves whose mask is zero (four out of five) end after one scalar load in passes 2 and 3.  Same rays, same arithmetic, same
 * counters (passes 2 and 3 add to the per-wave record pass 1 wrote): bit-identical to the one-kernel form, which the tests check.
 */
#ifndef VRT_PRIMARY_PASS_WAVES
#define VRT_PRIMARY_PASS_WAVES 7
#endif
#ifndef VRT_SHADOW_PASS_WAVES
#define VRT_SHADOW_PASS_WAVES 7
#endif
template <int PATH, bool SINGLE, bool DYN = false>
__global__ __launch_bounds__(kMarchThreads) __attribute__((amdgpu_waves_per_eu((!SINGLE && PATH != VRT_PATH_DENSE) ? VRT_PRIMARY_PASS_WAVES : 1)))
void primary_pass_kernel(const DBlock B) {
    const int frame = (int)blockIdx.y;
    DFrame Fd;
    const DFrame& F = frame_view<DYN>(B, frame, Fd);
    const DCam C = load_cam(B, frame);
    int b, wave;
    block_and_wave(b, wave);
    int tile_x, tile_y;
    tile_of_block(F, b, (int)gridDim.x / kMarchGridMul, tile_x, tile_y);
    const int lane = (int)threadIdx.x & 63;
    int px, pyl;
    pixel_of_lane(tile_x, tile_y, wave, lane, px, pyl);
    const int py = frame_row(F, pyl);
    const bool valid = tile_x < F.tiles_x && tile_y < F.tiles_y && px < F.width && pyl < F.rows && py < F.height;

    Counters k;
    DiagAcc dg;
    const bool reach = wave_can_reach(C, valid, px, py);
    bool hit = false;
    if (valid) {
        F3 o, d;
        camera_ray(F, C, px, py, o, d);
        k.n_primary = 1;
        float t_hit = 0.0f;
        int inst = 0;
        F3 n = f3(0.0f, 0.0f, 0.0f);
        const unsigned sky = env_fetch(F.env, F.env_size, d);
        if (reach && trace_closest<PATH, SINGLE, false, 2, true>(F, o, d, 10000.0f, 0.0f, t_hit, inst, n, k.s_primary, k.n_hits)) {
            k.n_hits += 1;
            hit = true;
            const size_t r = pass_record(F, frame, b, wave, lane);
            F.hit_rec[r] = HitRecord{n.x, n.y, n.z, t_hit};
            F.hit_aux[r] = (unsigned)inst;
        } else {
            store_pixel(F, frame, (unsigned)pyl * (unsigned)F.width + (unsigned)px, env_decode(sky));
        }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) F.hit_mask[pass_wave(F, frame, b, wave)] = m;
    write_records<false, true>(F, frame, b, wave, lane, k, dg, 0ull);
}

template <int PATH, bool SINGLE, bool DYN = false>
__global__ __launch_bounds__(kMarchThreads) __attribute__((amdgpu_waves_per_eu(SINGLE ? 8 : VRT_SHADOW_PASS_WAVES))) void light_pass_kernel(const DBlock B) {
    const int frame = (int)blockIdx.y;
    DFrame Fd;
    const DFrame& F = frame_view<DYN>(B, frame, Fd);
    int b, wave;
    block_and_wave(b, wave);
    const int lane = (int)threadIdx.x & 63;
    const unsigned long long m = F.hit_mask[pass_wave(F, frame, b, wave)];
    if (m == 0ull) return;
    const DCam C = load_cam(B, frame);
    int tile_x, tile_y;
    tile_of_block(F, b, (int)gridDim.x / kMarchGridMul, tile_x, tile_y);
    int px, pyl;
    pixel_of_lane(tile_x, tile_y, wave, lane, px, pyl);
    const int py = frame_row(F, pyl);
    Counters k;
    DiagAcc dg;
    bool bounces = false;
    if (((m >> lane) & 1ull) != 0ull) {
        F3 o, d;
        camera_ray(F, C, px, py, o, d);
        const size_t r = pass_record(F, frame, b, wave, lane);
        unsigned bits = 0u;
        if (F.shadow && !F.unlit) {
            const HitRecord h = F.hit_rec[r];
            const F3 so = shadow_origin(F, o, d, h.t, F.view_vec ? camera_len(F, C, px, py) : 1.0f);
            const float tb = 0.0f + h.t;
            const F3 sun = f3(F.light_dir[0], F.light_dir[1], F.light_dir[2]);
            /* the directional light's ray is left out where the one-kernel form leaves it out: it is this hit's only term and the
               surface (with its material's normal map, if any) faces away from the light */
            bool cast_sun = true;
            if (F.n_point == 0 && F.n_spot == 0) {
                const int inst = (int)F.hit_aux[r];
                const DVolume* V = SINGLE ? F.vol0 : F.vols + F.inst[inst].slot;
                F3 n = f3(h.nx, h.ny, h.nz);
                F3 albedo = f3(V->tint[0], V->tint[1], V->tint[2]);
                float rough = V->roughness, metal = V->metallic;
                if (F.textured) {
                    const F3 hp = f3(__builtin_fmaf(d.x, h.t, o.x), __builtin_fmaf(d.y, h.t, o.y), __builtin_fmaf(d.z, h.t, o.z));
                    textured_surface(V, SINGLE ? F.inst : F.inst + inst, hp, albedo, n, rough, metal);
                }
                const bool bounce = rough < 0.3f && 1 <= F.max_bounces;
                cast_sun = bounce || dot3(n, sun) > 0.0f;
            }
            if (cast_sun) {
                k.n_shadow++;
                if (trace_any<PATH, SINGLE, false, true>(F, so, sun, 5000.0f, tb, k.s_shadow, k.n_hits)) bits |= kShadowBitDir;
            }
            const int n_lights = F.n_point + F.n_spot;
#pragma unroll 1
            for (int i = 0; i < n_lights; i++) {
                const bool is_point = i < F.n_point;
                const float* pos = is_point ? F.point[i].pos : F.spot[i - F.n_point].pos;
                const F3 dl = f3(pos[0] - so.x, pos[1] - so.y, pos[2] - so.z);
                const float dist = sqrtf(dot3(dl, dl));
                float inten, least;
                if (is_point) { /* ComputePointLightIntensity, Lighting.hlsli:17-20 */
                    const DPointLight L = F.point[i];
                    inten = L.intensity / ((1.0f + L.att_linear * dist) + (L.att_exp * dist) * dist);
                    least = 0.005f;
                } else {        /* ComputeSpotLightIntensity, Lighting.hlsli:30-48 */
                    const DSpotLight L = F.spot[i - F.n_point];
                    const F3 sd = f3(-dl.x, -dl.y, -dl.z) * (1.0f / dist);
                    const float cs = dot3(f3(L.fwd[0], L.fwd[1], L.fwd[2]), sd);
                    inten = 0.0f;
                    if (cs >= 0.0f && cs > L.cos_angle) {
                        const float delta = (cs - L.cos_angle) / (L.cos_falloff - L.cos_angle);
                        const float base = L.intensity * minf_(delta, 1.0f);
                        inten = base / ((1.0f + L.att_linear * dist) + (L.att_exp * dist) * dist);
                    }
                    least = 0.01f;
                }
                if (inten > least) {
                    k.n_shadow++;
                    if (trace_any<PATH, SINGLE>(F, so, dl * (1.0f / dist), dist, tb, k.s_shadow, k.n_hits))
                        bits |= is_point ? (kShadowBitPoint << i) : (kShadowBitSpot << (i - F.n_point));
                }
            }
        }
        /* the hit's shading (level 1 of march_kernel_full) from the record, read again behind the marches so that nothing of it
           is alive across them */
        {
            const HitRecord h = F.hit_rec[r];
            const unsigned aux = F.hit_aux[r];
            const int inst = (int)(aux & 0xffffu);
            const DVolume* V = SINGLE ? F.vol0 : F.vols + F.inst[inst].slot;
            F3 n = f3(h.nx, h.ny, h.nz);
            F3 albedo = f3(V->tint[0], V->tint[1], V->tint[2]);
            float rough = V->roughness, metal = V->metallic;
            if (F.textured) {
                const F3 hp = f3(__builtin_fmaf(d.x, h.t, o.x), __builtin_fmaf(d.y, h.t, o.y), __builtin_fmaf(d.z, h.t, o.z));
                textured_surface(V, SINGLE ? F.inst : F.inst + inst, hp, albedo, n, rough, metal);
            }
            bounces = !F.unlit && rough < 0.3f && 1 <= F.max_bounces;
            if (bounces) {
                F.hit_aux[r] = aux | bits; /* the third pass shades this hit and follows its mirror ray */
            } else {
                F3 color = albedo;
                if (!F.unlit) {
                    const float vs = F.view_vec ? camera_len(F, C, px, py) : 1.0f;
                    color = direct_light_from_bits(F, shadow_origin(F, o, d, h.t, vs), f3(-d.x * vs, -d.y * vs, -d.z * vs), n, albedo, rough, metal, V->k, F.shadow != 0, false, bits);
                }
                store_pixel(F, frame, (unsigned)pyl * (unsigned)F.width + (unsigned)px, color);
            }
        }
    }
    /* the wave's mask now names the lanes that bounce: all the third pass looks at */
    const unsigned long long bm = __ballot(bounces);
    if (lane == 0) F.hit_mask[pass_wave(F, frame, b, wave)] = bm;
    write_records<false, false, true>(F, frame, b, wave, lane, k, dg, 0ull);
}

/* ---- hybrid march: per-lane head, wave-cooperative LDS tail ----------------------------------- */

constexpr int kLdsSlots = 8;                 /* bricks cached per wave: any 2x2x2 brick neighbourhood fits */
constexpr unsigned kTagInvalid = 0xffffffffu;
constexpr int kHeadSteps = 12;               /* samples every ray takes from global memory before the LDS phase */

/*
 * Phase 2 of the hybrid march: the rays of a wave that are still marching after kHeadSteps iterations
 * (rays that graze a surface: 100+ dependent samples, each a full L2/HBM round trip for a lone wave —
 * they set the kernel's tail) continue with their taps served from a per-wave LDS brick cache.
 * ALL 64 lanes run this loop (wave-uniform control flow); `active` marks the lanes that carry a ray.
 * Per iteration a lane first consults the two-level empty-space table (small global tables) and advances
 * without taps where it may; a lane that samples looks up its brick's slot (direct-mapped on the low bit of
 * each brick coordinate), reads the tag and the 8 taps together, and keeps them when the tag matches; lanes
 * that miss vote (__ballot), the first one's brick is fetched by the whole wave with one coalesced 512-B
 * read (64 lanes x 8 B) into its slot, and the lookup repeats.  If the slot is in use by lanes that hit
 * in this very sample, the missing lanes take their taps from global memory instead (no eviction
 * ping-pong).  LDS operations of one wave execute in order: no barrier.  The loop ends as soon as no
 * lane is active (__ballot early-out).  Same positions, same taps, same arithmetic: bit-identical results.
 */
__device__ __forceinline__ void march_tail_lds(const DFrame& F, const VolRef& V, const RaySeg& R, bool active, MarchState& st,
                                               float* __restrict__ slots, unsigned* __restrict__ tags, int lane, unsigned& steps) {
    float t = st.t, t_prev = st.t_prev, s_prev = st.s_prev;
    int i = st.i;
    bool relaxed = st.chk < __builtin_inff(); /* the step that led to the current position was a stretched one (march_lane) */
    const int max_steps = F.max_steps;
    const unsigned nb = (unsigned)V.nb;
    for (;;) {
        if (active && (i >= max_steps || t > R.t_end)) active = false;
        if (__ballot(active) == 0ull) break;
        const Cell c = cell_at(R, t);
        const unsigned bx = (unsigned)c.cx >> 2, by = (unsigned)c.cy >> 2, bz = (unsigned)c.cz >> 2;
        float leap = 0.0f;
        bool need = active;
        if (V.skip != nullptr && active) {
            const unsigned brick = (bx * nb + bz) * nb + by;
            leap = leap_of(R, c, V.nib[brick]);
            if (leap >= R.smax && t <= R.t_skip_end) {
                t_prev = t;
                s_prev = R.smax;
                relaxed = false;
                t = t + __builtin_fmaxf(__builtin_fmaf(t, F.cone_eps, R.base_min), leap);
                i++;
                need = false; /* skipped: no taps this iteration */
            }
        }
        const bool sample = need;
        const unsigned tag = (bx & 0xffu) | ((by & 0xffu) << 8) | ((bz & 0xffu) << 16);
        const unsigned slot = (bx & 1u) | ((by & 1u) << 1) | ((bz & 1u) << 2);
        const float* sp = slots + (slot << 7) + brick_local(c.cx, c.cy, c.cz);
        bool use_global = false;
        Taps taps;
        for (;;) {
            /* speculative: tag and taps travel together, the taps are only meaningful on a tag match */
            const unsigned tg = tags[slot];
            taps.y00a = sp[0];
            taps.y00b = sp[1];
            taps.y01a = sp[5];
            taps.y01b = sp[6];
            taps.y10a = sp[25];
            taps.y10b = sp[26];
            taps.y11a = sp[30];
            taps.y11b = sp[31];
            const bool ok = tg == tag;
            const unsigned long long miss = __ballot(need && !ok);
            if (miss == 0ull) break; /* every lane that needs taps has them */
            const int leader = __builtin_ctzll(miss);
            const unsigned ltag = __builtin_amdgcn_readlane(tag, leader);
            const unsigned lbx = ltag & 0xffu, lby = (ltag >> 8) & 0xffu, lbz = (ltag >> 16) & 0xffu;
            const unsigned lslot = (lbx & 1u) | ((lby & 1u) << 1) | ((lbz & 1u) << 2);
            if (__ballot(need && ok && slot == lslot) != 0ull) {
                /* the slot serves other lanes right now: the leader's brick is read from global memory */
                if (need && tag == ltag) {
                    use_global = true;
                    need = false;
                }
                continue;
            }
            const unsigned brick = (lbx * nb + lbz) * nb + lby;
            const gfloat_p src = (gfloat_p)(V.p + (((size_t)brick << 9) + ((unsigned)lane << 3)));
            const float v0 = src[0], v1 = src[1];
            float* dst = slots + (lslot << 7) + ((unsigned)lane << 1);
            dst[0] = v0;
            dst[1] = v1;
            if (lane == 0) tags[lslot] = ltag;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (__ballot(use_global) != 0ull) {
            if (use_global) taps = fetch8<VRT_PATH_BRICK>(V, c.cx, c.cy, c.cz);
        }
        const float s = lerp8(taps, c.fx, c.fy, c.fz) * R.ds;
        if (sample) {
            steps++;
            st.c = c;
            if (relaxed && __builtin_fmaxf(__builtin_fminf(s, R.smax), 0.0f) + s_prev < t - t_prev) { /* see march_lane */
                relaxed = false;
                t = t_prev + __builtin_fmaxf(s_prev, __builtin_fmaf(t_prev, F.cone_eps, R.base_min));
                i++;
            } else if (s < __builtin_fmaf(t, F.cone_eps, F.eps_hit)) {
                st.hit = true;
                st.s_hit = s;
                active = false;
            } else {
                i++;
                const float s_gate = kRelaxGate * s_prev;
                t_prev = t;
                s_prev = __builtin_fminf(s, R.smax);
                const float adv_min = __builtin_fmaxf(__builtin_fmaf(t, F.cone_eps, R.base_min), leap);
                const float om = F.k_relax;
                const float plain = __builtin_fmaxf(s_prev, adv_min);
                const float stretched = __builtin_fmaxf(__builtin_fminf(s * om, R.smax_relax), adv_min);
                relaxed = stretched > plain && s >= s_gate && t + stretched <= R.t_end;
                t = t + ((relaxed || om < 1.0f) ? stretched : plain);
            }
        }
        if (active) { /* keep the lane's state current: it is read back when the lane retires */
            st.t = t;
            st.t_prev = t_prev;
            st.s_prev = s_prev;
            st.i = i;
        }
    }
    if (!st.hit) {
        st.t = t;
        st.i = i;
    }
}

/*
 * One ray of every lane against the single instance of the scene.  Phase 1: kHeadSteps iterations per
 * lane with taps from the bricks in global memory (most rays finish here).  Phase 2 (only if some
 * lane is still marching): the LDS tail above.  Returns true on hit with t / cell / iteration.
 */
template <bool CLOSEST>
__device__ __forceinline__ bool march_hybrid(const DFrame& F, const VolRef& V, const RaySeg& R, bool act, float* slots,
                                             unsigned* tags, int lane, float& t_hit, Cell& c_hit, int& iter_hit, unsigned& steps, unsigned& ex) {
    MarchState st;
    st.t = st.t_prev = R.t0;
    st.s_prev = st.s_hit = 0.0f;
    st.i = 0;
    st.hit = false;
    st.chk = __builtin_inff();
    st.c = Cell{0, 0, 0, 0.0f, 0.0f, 0.0f};
    const int max_steps = F.max_steps;
    const int head = max_steps < kHeadSteps ? max_steps : kHeadSteps;
    if (act) march_lane<VRT_PATH_BRICK, false>(F, V, R, st, head, steps, nullptr);
    const bool cont = act && !st.hit && st.i < max_steps && !(st.t > R.t_end);
    if (__ballot(cont) != 0ull) march_tail_lds(F, V, R, cont, st, slots, tags, lane, steps);
    if (!st.hit) {
        if (act && max_steps > 0 && st.i >= max_steps && !(st.t > R.t_end)) ex += kExhaustedOne;
        return false;
    }
    float t = st.t;
    Cell c = st.c;
    if (st.s_hit < 0.0f && st.i > 0) t = refine_hit<VRT_PATH_BRICK>(V, R, st.t_prev, st.s_prev, t, st.s_hit, c, steps);
    else if (CLOSEST && F.polish > 0 && st.i > 0) t = polish_hit<VRT_PATH_BRICK>(F, V, R, t, st.s_hit, st.t_prev, st.s_prev, F.polish, c, steps);
    t_hit = t;
    c_hit = c;
    iter_hit = st.i;
    return true;
}

/*
 * Single-instance kernel with the hybrid march (VRT_PATH_BRICK_LDS).  Same tile mapping and per-pixel
 * arithmetic as march_kernel; the skeleton is wave-uniform so that every lane can take part in the
 * brick fills of the LDS phase.
 */
template <bool DIAG>
__global__ __launch_bounds__(kBlockThreads) void march_kernel_coop(const DBlock B) {
    __shared__ float s_slots[4][kLdsSlots * kBrickFloats];
    __shared__ unsigned s_tags[4][kLdsSlots];
    unsigned long long t_start = 0;
    if constexpr (DIAG) t_start = __builtin_amdgcn_s_memrealtime();
    const DFrame& F = B.f;
    const int frame = (int)blockIdx.y;
    const DCam C = load_cam(B, frame);
    const int b = (int)blockIdx.x;
    int tile_x, tile_y;
    tile_of_block(F, b, (int)gridDim.x, tile_x, tile_y);
    const int wave = (int)threadIdx.x >> 6;
    const int lane = (int)threadIdx.x & 63;
    int px, pyl;
    pixel_of_lane(tile_x, tile_y, wave, lane, px, pyl);
    const int py = frame_row(F, pyl);
    const bool valid = tile_x < F.tiles_x && tile_y < F.tiles_y && px < F.width && pyl < F.rows && py < F.height;

    float* slots = s_slots[wave];
    unsigned* tags = s_tags[wave];
    if (lane < kLdsSlots) tags[lane] = kTagInvalid;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    Counters k;
    DiagAcc dg;
    const DInstance* I = F.inst;
    const DVolume* Vd = F.vol0;
    const VolRef V = load_vol<VRT_PATH_BRICK>(Vd);

    F3 o = f3(0.0f, 0.0f, 0.0f), d = f3(1.0f, 0.0f, 0.0f);
    RaySeg R = {};
    bool act = false;
    const bool reach = wave_can_reach(C, valid, px, py);
    if (valid) {
        camera_ray(F, C, px, py, o, d);
        k.n_primary = 1;
        if (reach) act = setup_ray(F, I, V, o, d, 10000.0f, 0.0f, R);
    }
    float t_hit = 0.0f;
    Cell c_hit = {0, 0, 0, 0.0f, 0.0f, 0.0f};
    int iter_hit = 0;
    const bool hit = march_hybrid<true>(F, V, R, act, slots, tags, lane, t_hit, c_hit, iter_hit, k.s_primary, k.n_hits);

    F3 n = f3(0.0f, 0.0f, 0.0f);
    if (hit) {
        k.n_hits += 1;
        n = hit_normal<VRT_PATH_BRICK, true>(I, V, R, c_hit, iter_hit);
    }
    bool shadowed = false;
    /* no shadow ray from a surface that faces away from the light (see march_kernel) */
    const bool want_shadow = hit && F.shadow && !F.unlit && dot3(n, f3(F.light_dir[0], F.light_dir[1], F.light_dir[2])) > 0.0f;
    if (__ballot(want_shadow) != 0ull) {
        RaySeg Rs = {};
        bool act_s = false;
        if (want_shadow) {
            k.n_shadow = 1;
            act_s = setup_shadow_ray(F, I, V, shadow_origin(F, o, d, t_hit), 5000.0f, t_hit, Rs);
        }
        float ts = 0.0f;
        Cell cs = {0, 0, 0, 0.0f, 0.0f, 0.0f};
        int is = 0;
        shadowed = march_hybrid<false>(F, V, Rs, act_s, slots, tags, lane, ts, cs, is, k.s_shadow, k.n_hits);
    }
    if (valid) {
        F3 color = hit ? shade_hit(F, Vd, d, n, shadowed) : env_lookup(F.env, F.env_size, d);
        store_pixel(F, frame, (unsigned)pyl * (unsigned)F.width + (unsigned)px, color);
    }
    write_records<DIAG, true>(F, frame, b, wave, lane, k, dg, t_start);
}

/* dense N^3 grid → 4^3-cell bricks with a one-sample apron (5^3 samples, padded to 128 floats). */
__global__ __launch_bounds__(128) void retile_bricks_kernel(const float* __restrict__ dense, float* __restrict__ bricks,
                                                            int N, int nb) {
    const int brick = (int)blockIdx.x; /* (bx*nb + bz)*nb + by */
    const int by = brick % nb;
    const int bz = (brick / nb) % nb;
    const int bx = brick / (nb * nb);
    const int l = (int)threadIdx.x;
    float v = 0.0f;
    if (l < 125) {
        const int lx = l / 25, lz = (l / 5) % 5, ly = l % 5;
        int x = bx * 4 + lx, y = by * 4 + ly, z = bz * 4 + lz;
        x = x > N - 1 ? N - 1 : x;
        y = y > N - 1 ? N - 1 : y;
        z = z > N - 1 ? N - 1 : z;
        v = dense[((size_t)x * N + z) * N + y];
    }
    bricks[(size_t)brick * kBrickFloats + l] = v;
}

/* The same for VRT_FORMAT_TEXEL16 volumes: the dense grid holds the integer field +-q as floats; bricks of 128 int16. */
__global__ __launch_bounds__(128) void retile_bricks16_kernel(const float* __restrict__ dense, short* __restrict__ bricks, int N, int nb) {
    const int brick = (int)blockIdx.x;
    const int by = brick % nb;
    const int bz = (brick / nb) % nb;
    const int bx = brick / (nb * nb);
    const int l = (int)threadIdx.x;
    short v = 0;
    if (l < 125) {
        const int lx = l / 25, lz = (l / 5) % 5, ly = l % 5;
        int x = bx * 4 + lx, y = by * 4 + ly, z = bz * 4 + lz;
        x = x > N - 1 ? N - 1 : x;
        y = y > N - 1 ? N - 1 : y;
        z = z > N - 1 ? N - 1 : z;
        v = (short)(int)dense[((size_t)x * N + z) * N + y]; /* |value| <= 32767, integer: exact */
    }
    bricks[(size_t)brick * kBrickFloats + l] = v;
}

/* VRT_PATH_CELLS: the 8 corner texels of every cell as one 16-byte record (cells beyond the grid repeat the last sample,
 * like the bricks' apron). */
__global__ __launch_bounds__(64) void retile_cells16_kernel(const float* __restrict__ dense, short* __restrict__ cells, int N, int nb) {
    const int brick = (int)blockIdx.x;
    const int by = brick % nb, bz = (brick / nb) % nb, bx = brick / (nb * nb);
    const int l = (int)threadIdx.x; /* record lx*16 + lz*4 + ly */
    const int lx = l >> 4, lz = (l >> 2) & 3, ly = l & 3;
    short v[8];
    for (int k = 0; k < 8; k++) { /* tap order: (x,z) = 00, 01, 10, 11; y then y+1 */
        int x = bx * 4 + lx + (k >> 2), z = bz * 4 + lz + ((k >> 1) & 1), y = by * 4 + ly + (k & 1);
        x = x > N - 1 ? N - 1 : x;
        y = y > N - 1 ? N - 1 : y;
        z = z > N - 1 ? N - 1 : z;
        v[k] = (short)(int)dense[((size_t)x * N + z) * N + y];
    }
    uint4v w;
    w.x = (unsigned)(unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16);
    w.y = (unsigned)(unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16);
    w.z = (unsigned)(unsigned short)v[4] | ((unsigned)(unsigned short)v[5] << 16);
    w.w = (unsigned)(unsigned short)v[6] | ((unsigned)(unsigned short)v[7] << 16);
    reinterpret_cast<uint4v*>(cells)[(size_t)brick * 64 + l] = w;
}

/* VRT_FORMAT_TEXEL16: a density as the reference's volume texel keeps it — sign + 15-bit trunc(|d| * 100)
 * (VDXVoxelVolume::EncodeVoxel, RDXVoxelVolume.cpp:399-421) — returned as the integer +-q (oracle: texel16_value). */
__device__ __forceinline__ float texel16_value(float d) {
    const float a = fabsf(d) * 100.0f;
    unsigned q = 0u;
    if (a >= 4294967040.0f) q = 0xffffffffu;
    else if (a >= 0.0f) q = (unsigned)a; /* NaN -> 0 */
    q &= 0x7fffu;
    const float v = (float)q;
    return d < 0.0f ? -v : v;
}

__global__ void quantize_field_kernel(float* __restrict__ density, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) density[i] = texel16_value(density[i]);
}

/* The reference's volume texture (N^3 RGBA8 texels, texel (x,y,z) at 4*(z*N*N + y*N + x): R = sign<<7 | q>>8, G = q & 0xff,
 * B = A = material; UpdateVolumeTexture, RDXVoxelVolume.cpp:294-327) -> integer field +-q in the grid's own order + materials. */
__global__ void texels_to_field_kernel(const uchar4* __restrict__ texels, float* __restrict__ density, uint8_t* __restrict__ material, int N) {
    const size_t count = (size_t)N * N * N;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) { /* i = x*N*N + z*N + y */
        const size_t x = i / ((size_t)N * N), z = (i / N) % N, y = i % N;
        const uchar4 t = texels[(z * N + y) * N + x];
        const float q = (float)((((unsigned)t.x & 0x7fu) << 8) | (unsigned)t.y);
        density[i] = (t.x & 0x80u) ? -q : q;
        material[i] = t.z;
    }
}

/* Empty-space table, level 2 (oracle: build_nibble_table).  Step 1: a cell is ACTIVE when one of its 8 corners holds a
 * trustworthy distance below the clamp. */
__global__ void active_cells_kernel(const float* __restrict__ dense, uint8_t* __restrict__ act, int N, float density_scale, float step_max) {
    const int C = N - 1;
    const size_t count = (size_t)C * C * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        const size_t x = i / ((size_t)C * C), z = (i / C) % C, y = i % C;
        bool a = false;
        for (int k = 0; k < 8; k++) {
            const size_t xx = x + (k >> 2), zz = z + ((k >> 1) & 1), yy = y + (k & 1);
            a = a || dense[(xx * N + zz) * N + yy] * density_scale < step_max;
        }
        act[i] = a ? 1 : 0;
    }
}

/* Steps 2-4: separable min-plus passes of the windowed squared Euclidean distance transform between cells (cube-to-cube
 * distance: per axis max(|d|-1, 0)), along y (AXIS 0, from the active flags), z (AXIS 1) and x (AXIS 2).  0xffff = none
 * within the window. */
template <int AXIS>
__global__ void edt_pass_kernel(const uint8_t* __restrict__ act, const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int C) {
    const size_t count = (size_t)C * C * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t pitch = AXIS == 0 ? 1 : (AXIS == 1 ? (size_t)C : (size_t)C * C);
    for (; i < count; i += stride) {
        const int pos = AXIS == 0 ? (int)(i % C) : (AXIS == 1 ? (int)((i / C) % C) : (int)(i / ((size_t)C * C)));
        int best = 0xffff;
        for (int d = -kNibWindow; d <= kNibWindow; d++) {
            const int q = pos + d;
            if (q < 0 || q >= C) continue;
            const size_t j = (size_t)((long long)i + (long long)d * (long long)pitch);
            int g = (d < 0 ? -d : d);
            g = g > 0 ? g - 1 : 0;
            g *= g;
            int v;
            if constexpr (AXIS == 0) v = act[j] ? g : 0xffff;
            else v = g + (int)in[j];
            best = v < best ? v : best;
        }
        out[i] = (uint16_t)(best > 0xffff ? 0xffff : best);
    }
}

/* Step 5: per brick the eight sub-block nibbles: min over the sub-block's cells of floor(sqrt(d2)), capped at 15. */
__global__ __launch_bounds__(64) void nibble_kernel(const uint16_t* __restrict__ d2, unsigned* __restrict__ nib, int C, int nb) {
    const int brick = (int)blockIdx.x;
    const int by = brick % nb, bz = (brick / nb) % nb, bx = brick / (nb * nb);
    const int l = (int)threadIdx.x; /* one lane per cell */
    const int lx = l >> 4, lz = (l >> 2) & 3, ly = l & 3;
    const int x = bx * 4 + lx, y = by * 4 + ly, z = bz * 4 + lz;
    int r = 15;
    if (x < C && y < C && z < C) {
        const int v = d2[((size_t)x * C + z) * C + y];
        r = 0;
        while (r < 15 && (r + 1) * (r + 1) <= v) r++;
    }
    unsigned w = 0;
    for (int k = 0; k < 8; k++) {
        const bool mine = ((lx >> 1) * 4 + (lz >> 1) * 2 + (ly >> 1)) == k;
        int m = mine ? r : 15;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(m, o);
            m = other < m ? other : m;
        }
        w |= (unsigned)m << (4 * k);
    }
    if (l == 0) nib[brick] = w;
}

/* Empty-space table, level 1, step 1: a brick is "near" (0) when any of its 5^3 samples holds a trustworthy
 * distance below the clamp, density*density_scale < step_max (equivalently: when it holds an active cell); everything
 * else starts at 255.  Samples come from the dense grid (whatever the brick format). */
__global__ __launch_bounds__(128) void skip_seed_kernel(const float* __restrict__ dense, uint8_t* __restrict__ table, int N, int nb,
                                                        float density_scale, float step_max) {
    const int brick = (int)blockIdx.x;
    const int l = (int)threadIdx.x;
    const int by = brick % nb, bz = (brick / nb) % nb, bx = brick / (nb * nb);
    bool near = false;
    if (l < 125) {
        const int lx = l / 25, lz = (l / 5) % 5, ly = l % 5;
        int x = bx * 4 + lx, y = by * 4 + ly, z = bz * 4 + lz;
        x = x > N - 1 ? N - 1 : x;
        y = y > N - 1 ? N - 1 : y;
        z = z > N - 1 ? N - 1 : z;
        near = dense[((size_t)x * N + z) * N + y] * density_scale < step_max;
    }
    const unsigned long long any0 = __ballot(near);
    __shared__ int flag[2];
    if ((l & 63) == 0) flag[l >> 6] = any0 != 0ull;
    __syncthreads();
    if (l == 0) table[brick] = (flag[0] || flag[1]) ? 0 : 255;
}

/* Bounding box, in bricks, of the near bricks (distance 0): box = {min x, z, y, max x, z, y}, preset to {nb.., -1..}. */
__global__ void active_box_kernel(const uint8_t* __restrict__ table, int nb, int* __restrict__ box) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= nb * nb * nb || table[i] != 0) return;
    const int by = i % nb, bz = (i / nb) % nb, bx = i / (nb * nb);
    atomicMin(&box[0], bx);
    atomicMin(&box[1], bz);
    atomicMin(&box[2], by);
    atomicMax(&box[3], bx);
    atomicMax(&box[4], bz);
    atomicMax(&box[5], by);
}

/* The march wants the leap count, not the distance: L = max(D-1, 0) (one convert + one multiply per sample). */
__global__ void skip_to_leap_kernel(uint8_t* __restrict__ table, int n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) {
        const uint8_t d = table[i];
        table[i] = d > 1 ? (uint8_t)(d - 1) : (uint8_t)0;
    }
}

/* Cube modes' table, step 1: a brick is a seed (0) when one of its 4^3 cell-origin voxels is solid
 * (density <= 0); voxels beyond cell N-2 do not exist (only at resolutions < 2, where one brick covers
 * the volume). */
__global__ __launch_bounds__(64) void cube_seed_kernel(const float* __restrict__ dense, uint8_t* __restrict__ table, int N, int nb) {
    const int brick = (int)blockIdx.x;
    const int l = (int)threadIdx.x;
    const int by = brick % nb, bz = (brick / nb) % nb, bx = brick / (nb * nb);
    const int lx = l >> 4, lz = (l >> 2) & 3, ly = l & 3;
    const int x = bx * 4 + lx, y = by * 4 + ly, z = bz * 4 + lz;
    bool solid = false;
    if (x <= N - 2 && y <= N - 2 && z <= N - 2) solid = dense[((size_t)x * N + z) * N + y] <= 0.0f;
    const unsigned long long any0 = __ballot(solid);
    if (l == 0) table[brick] = any0 != 0ull ? 0 : 255;
}

/* Step k of the exact Chebyshev distance transform: bricks still at 255 that touch (3x3x3) a brick at
 * distance k-1 get distance k.  Launched for k = 1..nb on ping-pong buffers. */
__global__ void skip_dilate_kernel(const uint8_t* __restrict__ cur, uint8_t* __restrict__ nxt, int nb, int k) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= nb * nb * nb) return;
    const int by = i % nb, bz = (i / nb) % nb, bx = i / (nb * nb);
    uint8_t d = cur[i];
    if (d == 255) {
        bool hit = false;
        for (int dx = -1; dx <= 1; dx++)
            for (int dz = -1; dz <= 1; dz++)
                for (int dy = -1; dy <= 1; dy++) {
                    const int x = bx + dx, y = by + dy, z = bz + dz;
                    if (x < 0 || y < 0 || z < 0 || x >= nb || y >= nb || z >= nb) continue;
                    hit = hit || cur[(x * nb + z) * nb + y] == (uint8_t)(k - 1);
                }
        if (hit) d = (uint8_t)k;
    }
    nxt[i] = d;
}

/* ---- device Voxelizer (Voxelizer/Private/VolumeConverter.cpp:161-252; arithmetic in voxelize_core.h) ------
 * One workgroup per triangle walks the triangle's voxel index box; every voxel keeps the minimum shell
 * density over all triangles.  The minimum is order-independent, so an integer atomicMin on order-preserving
 * keys reproduces the CPU converter's sequential "keep if smaller" bit for bit.  The keys live in the dense
 * density buffer itself and are turned back into floats (and materials) by voxelize_finish_kernel. */
__global__ void voxelize_fill_kernel(int* __restrict__ keys, size_t count, int key) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) keys[i] = key;
}

__global__ __launch_bounds__(256) void voxelize_kernel(const vrt_vox::TriangleFrame* __restrict__ tris, int* __restrict__ keys, int N,
                                                       float cell, float extent, float threshold) {
    const vrt_vox::TriangleFrame t = tris[blockIdx.x];
    const int nx = t.hi[0] - t.lo[0] + 1, ny = t.hi[1] - t.lo[1] + 1, nz = t.hi[2] - t.lo[2] + 1;
    if (nx <= 0 || ny <= 0 || nz <= 0) return;
    const long long total = (long long)nx * ny * nz;
    for (long long j = threadIdx.x; j < total; j += blockDim.x) {
        /* y fastest: neighbouring lanes hit neighbouring addresses of the x*N*N + z*N + y layout */
        const int ly = (int)(j % ny);
        const int lz = (int)((j / ny) % nz);
        const int lx = (int)(j / ((long long)ny * nz));
        const int x = t.lo[0] + lx, y = t.lo[1] + ly, z = t.lo[2] + lz;
        const float dist = vrt_vox::region_distance(t, vrt_vox::voxel_position(x, y, z, cell, extent));
        const float density = vrt_vox::shell_density(dist, threshold);
        atomicMin(&keys[((size_t)x * N + z) * N + y], vrt_vox::ordered_key(density));
    }
}

__global__ void voxelize_finish_kernel(float* __restrict__ density, uint8_t* __restrict__ material, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        const float d = vrt_vox::from_ordered_key(__float_as_int(density[i]));
        density[i] = d;
        material[i] = d <= 0.0f ? 1 : 0; /* VolumeConverter.cpp:206-207; untouched voxels keep material 0 */
    }
}

/* VVoxel records (8 B: u8 material, pad, f32 density) → dense fp32 densities + u8 materials. */
__global__ void split_voxels_kernel(const uint2* __restrict__ voxels, float* __restrict__ density,
                                    uint8_t* __restrict__ material, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        uint2 r = voxels[i];
        density[i] = __uint_as_float(r.y);
        material[i] = (uint8_t)(r.x & 0xffu);
    }
}

/* ---- the march's inner operation in isolation (vrt_debug_gather_ceiling) -------------------------------------
 * The 8 taps of a trilinear sample from a brick pool (fetch8_at: 4 x dwordx2 of fp32 bricks, or 4 x dword of int16 bricks) + the lerp tree
 * (lerp8), every lane on its own pseudo-random INDEPENDENT sequence of cells (no dependence on the loaded values), at full occupancy: the
 * rate the texture path (TA / L1 / L2 / HBM) sustains for this access pattern — the roof the march kernel's busy phase is under, measured on
 * the very box and build a bench runs on (bench.py roofline.limiter_ceiling_*).  COHERENT: the 64 lanes of a wave on the 3x3 cells an
 * 8x8-pixel tile covers (one wave-uniform base cell per iteration; lanes that share a cell share its lines) instead of a cell each.
 * (tools/microbench/gather.hip is the same measurement as a stand-alone program.) */
template <int DP, bool COHERENT>
__global__ __launch_bounds__(256) void gather_ceiling_kernel(const char* __restrict__ pool, unsigned nbricks_mask, int iters, float* __restrict__ out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    unsigned state = (COHERENT ? (gid >> 6) : gid) * 2654435761u + 12345u;
    unsigned brick = (gid >> 6) * 97u;
    const unsigned ox = ((lane & 7u) * 3u) >> 3, oz = ((lane >> 3) * 3u) >> 3;
    VolRef V = {};
    V.p = (gchar_p)pool;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
        state = state * 1664525u + 1013904223u;
        const unsigned bx = (state >> 8) & 3u, ly = (state >> 12) & 3u, bz = (state >> 16) & 3u;
        if (((state >> 20) & 7u) == 0u) brick += 1u + ((state >> 24) & 3u); /* change brick every ~8 samples */
        const unsigned cx = COHERENT ? bx + ox : bx, cz = COHERENT ? bz + oz : bz;
        const unsigned b = (brick + (cx >> 2) * 17u + (cz >> 2) * 5u) & nbricks_mask; /* a neighbouring brick past the edge */
        const Taps t = fetch8_at<DP>(V, b, (int)(cx & 3u), (int)ly, (int)(cz & 3u));
        acc += lerp8(t, 0.3f, 0.6f, 0.2f);
    }
    out[gid] = acc;
}

/* ---- launch wrappers (host) -------------------------------------------------------------- */

hipError_t launch_gather_ceiling(const void* pool, unsigned n_bricks, int format, bool coherent, int iters, float* out, int blocks, hipStream_t stream) {
    const dim3 g((unsigned)blocks), t(256);
    const char* p = static_cast<const char*>(pool);
    if (format == VRT_FORMAT_TEXEL16) {
        if (coherent) hipLaunchKernelGGL((gather_ceiling_kernel<kPathBrick16, true>), g, t, 0, stream, p, n_bricks - 1, iters, out);
        else hipLaunchKernelGGL((gather_ceiling_kernel<kPathBrick16, false>), g, t, 0, stream, p, n_bricks - 1, iters, out);
    } else {
        if (coherent) hipLaunchKernelGGL((gather_ceiling_kernel<VRT_PATH_BRICK, true>), g, t, 0, stream, p, n_bricks - 1, iters, out);
        else hipLaunchKernelGGL((gather_ceiling_kernel<VRT_PATH_BRICK, false>), g, t, 0, stream, p, n_bricks - 1, iters, out);
    }
    return hipGetLastError();
}


/* The lean kernel's REF instantiation is the one that reads DFrame::textured (constant textures only: anything else makes the launch
   a full closest hit), view_vec and zero_outside. */
static bool needs_ref_kernel(const DFrame& F) { return F.textured != 0 || F.view_vec != 0 || F.zero_outside != 0; }

template <int PATH, bool SINGLE>
static hipError_t launch_t(const DBlock& B, hipStream_t stream) {
    const DFrame& F = B.f;
    const int grid = grid_blocks(F.tiles_x, F.tiles_y, F.tile_map);
    if (grid <= 0) return hipSuccess;
    if (F.diag)
        hipLaunchKernelGGL((march_kernel<PATH, SINGLE, true>), dim3((unsigned)(grid * kMarchGridMul), (unsigned)F.n_frames), dim3(kMarchThreads), 0, stream, B);
    else
        hipLaunchKernelGGL((march_kernel<PATH, SINGLE, false>), dim3((unsigned)(grid * kMarchGridMul), (unsigned)F.n_frames), dim3(kMarchThreads), 0, stream, B);
    return hipGetLastError();
}

/* Paths without a diagnostic instantiation. */
template <int PATH, bool SINGLE>
static hipError_t launch_nodiag_t(const DBlock& B, hipStream_t stream) {
    const DFrame& F = B.f;
    const int grid = grid_blocks(F.tiles_x, F.tiles_y, F.tile_map);
    if (grid <= 0) return hipSuccess;
    const dim3 g((unsigned)(grid * kMarchGridMul), (unsigned)F.n_frames), t(kMarchThreads);
    if (needs_ref_kernel(F)) { /* constant material textures / the reference's view vector / boundary texels: the REF instantiation */
        if (F.dyn != nullptr) hipLaunchKernelGGL((march_kernel<PATH, SINGLE, false, true, true>), g, t, 0, stream, B);
        else hipLaunchKernelGGL((march_kernel<PATH, SINGLE, false, false, true>), g, t, 0, stream, B);
    } else if (F.dyn != nullptr) {
        hipLaunchKernelGGL((march_kernel<PATH, SINGLE, false, true>), g, t, 0, stream, B);
    } else {
        hipLaunchKernelGGL((march_kernel<PATH, SINGLE, false>), g, t, 0, stream, B);
    }
    return hipGetLastError();
}

static hipError_t launch_coop(const DBlock& B, hipStream_t stream) {
    const DFrame& F = B.f;
    const int grid = grid_blocks(F.tiles_x, F.tiles_y, F.tile_map);
    if (grid <= 0) return hipSuccess;
    if (F.diag)
        hipLaunchKernelGGL((march_kernel_coop<true>), dim3((unsigned)grid, (unsigned)F.n_frames), dim3(kBlockThreads), 0, stream, B);
    else
        hipLaunchKernelGGL((march_kernel_coop<false>), dim3((unsigned)grid, (unsigned)F.n_frames), dim3(kBlockThreads), 0, stream, B);
    return hipGetLastError();
}

template <int PATH, bool SINGLE>
static hipError_t launch_full_t(const DBlock& B, hipStream_t stream) {
    const DFrame& F = B.f;
    const int grid = grid_blocks(F.tiles_x, F.tiles_y, F.tile_map);
    if (grid <= 0) return hipSuccess;
    const dim3 g((unsigned)(grid * kMarchGridMul), (unsigned)F.n_frames), t(kMarchThreads);
    if (F.dyn != nullptr) { /* per-frame scene state: the DYN instantiations */
        if (F.hit_rec != nullptr) {
            hipLaunchKernelGGL((primary_pass_kernel<PATH, SINGLE, true>), g, t, 0, stream, B);
            hipLaunchKernelGGL((light_pass_kernel<PATH, SINGLE, true>), g, t, 0, stream, B);
            if (F.may_bounce) hipLaunchKernelGGL((march_kernel_full<PATH, false, true, true>), g, t, 0, stream, B);
            return hipGetLastError();
        }
        hipLaunchKernelGGL((march_kernel_full<PATH, SINGLE, false, true>), g, t, 0, stream, B);
        return hipGetLastError();
    }
    if (F.hit_rec != nullptr) { /* three passes, see primary_pass_kernel */
        hipLaunchKernelGGL((primary_pass_kernel<PATH, SINGLE>), g, t, 0, stream, B);
        hipLaunchKernelGGL((light_pass_kernel<PATH, SINGLE>), g, t, 0, stream, B);
        if (F.may_bounce) hipLaunchKernelGGL((march_kernel_full<PATH, false, true>), g, t, 0, stream, B);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((march_kernel_full<PATH, SINGLE>), g, t, 0, stream, B);
    return hipGetLastError();
}

template <int PATH>
static hipError_t launch_path(const DBlock& B, bool single, bool diag_build, hipStream_t stream) {
    const DFrame& F = B.f;
    /* the one-kernel full closest hit always walks the (wave-uniform) BVH, a one-node tree included: its single-instance
       specialisation kept every instance / volume field live across the whole kernel (128-153 VGPRs, 3 waves per SIMD, against
       108-125 and 4) and measured 7 % slower on a one-instance scene with a point light (profiles/r02_full_closest_hit_kernel.txt;
       again in round 3: 29.6 against 34.0 Grays/s).  The march passes of the pass form hold no such state: 58 / 64 VGPRs */
    if (F.full) {
        if (F.hit_rec != nullptr && single) return launch_full_t<PATH, true>(B, stream); /* passes 1 and 2 without the BVH walk */
        return launch_full_t<PATH, false>(B, stream);
    }
    if constexpr (PATH == VRT_PATH_DENSE || PATH == VRT_PATH_BRICK || PATH == kPathBrick16 || PATH == kPathCells16) {
        if (diag_build) {
            if (needs_ref_kernel(F)) return hipErrorInvalidValue; /* (the diagnostic build has no REF instantiation) */
            return single ? launch_t<PATH, true>(B, stream) : launch_t<PATH, false>(B, stream);
        }
    }
    return single ? launch_nodiag_t<PATH, true>(B, stream) : launch_nodiag_t<PATH, false>(B, stream);
}

hipError_t launch_march(const DBlock& B, int path, bool single, hipStream_t stream) {
    const DFrame& F = B.f;
    if (F.n_frames < 1 || F.n_frames > (F.cams != nullptr ? kMaxLaunchFrames : kMaxBlockFrames)) return hipErrorInvalidValue;
    if (F.dyn != nullptr && F.diag) return hipErrorInvalidValue; /* (the diagnostic build has no per-frame-scene instantiation) */
    switch (path) {
        case kPathCube: return launch_path<kPathCube>(B, single, false, stream);
        case kPathCube16: return launch_path<kPathCube16>(B, single, false, stream);
        case kPathBrick16: return launch_path<kPathBrick16>(B, single, F.diag != 0, stream);
        case kPathCells16: return launch_path<kPathCells16>(B, single, F.diag != 0, stream);
        case VRT_PATH_DENSE: return launch_path<VRT_PATH_DENSE>(B, single, F.diag != 0, stream);
        case VRT_PATH_BRICK_LDS:
            if (single && !F.full && F.dyn == nullptr && !needs_ref_kernel(F)) return launch_coop(B, stream);
            return launch_path<VRT_PATH_BRICK>(B, single, F.diag != 0, stream);
        default: return launch_path<VRT_PATH_BRICK>(B, single, F.diag != 0, stream);
    }
}

hipError_t launch_retile(const float* dense, void* bricks, int format, int N, int nb, hipStream_t stream) {
    if (format == VRT_FORMAT_TEXEL16)
        hipLaunchKernelGGL(retile_bricks16_kernel, dim3((unsigned)(nb * nb * nb)), dim3(128), 0, stream, dense, static_cast<short*>(bricks), N, nb);
    else
        hipLaunchKernelGGL(retile_bricks_kernel, dim3((unsigned)(nb * nb * nb)), dim3(128), 0, stream, dense, static_cast<float*>(bricks), N, nb);
    return hipGetLastError();
}

hipError_t launch_retile_cells16(const float* dense, void* cells, int N, int nb, hipStream_t stream) {
    hipLaunchKernelGGL(retile_cells16_kernel, dim3((unsigned)(nb * nb * nb)), dim3(64), 0, stream, dense, static_cast<short*>(cells), N, nb);
    return hipGetLastError();
}

hipError_t launch_quantize_field(float* density, size_t count, hipStream_t stream) {
    hipLaunchKernelGGL(quantize_field_kernel, dim3(2048), dim3(256), 0, stream, density, count);
    return hipGetLastError();
}

hipError_t launch_texels_to_field(const void* texels, float* density, uint8_t* material, int N, hipStream_t stream) {
    hipLaunchKernelGGL(texels_to_field_kernel, dim3(2048), dim3(256), 0, stream, reinterpret_cast<const uchar4*>(texels), density, material, N);
    return hipGetLastError();
}

static hipError_t dilate_table(uint8_t* table, uint8_t* scratch, int nb, hipStream_t stream);

hipError_t launch_skip_table(const float* dense, uint8_t* table, uint8_t* scratch, int* box6, int N, int nb, float density_scale,
                             float step_max, hipStream_t stream) {
    const int n = nb * nb * nb;
    hipLaunchKernelGGL(skip_seed_kernel, dim3((unsigned)n), dim3(128), 0, stream, dense, table, N, nb, density_scale, step_max);
    hipError_t e = dilate_table(table, scratch, nb, stream);
    if (e != hipSuccess) return e;
    const int preset[6] = {nb, nb, nb, -1, -1, -1};
    e = hipMemcpyAsync(box6, preset, sizeof preset, hipMemcpyHostToDevice, stream); /* pageable source: staged before return */
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(active_box_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, table, nb, box6);
    hipLaunchKernelGGL(skip_to_leap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, table, n);
    return hipGetLastError();
}

size_t nibble_scratch_bytes(int N) {
    const size_t C = (size_t)(N - 1);
    return C * C * C * 5 + 64; /* active flags (1 B) + two ping-pong squared-distance grids (2 B each) per cell */
}

hipError_t launch_nibble_table(const float* dense, unsigned* nib, void* scratch, int N, int nb, float density_scale, float step_max,
                               hipStream_t stream) {
    const int C = N - 1;
    const size_t cells = (size_t)C * C * C;
    uint8_t* act = static_cast<uint8_t*>(scratch);
    uint16_t* g = reinterpret_cast<uint16_t*>(act + ((cells + 63) & ~(size_t)63));
    uint16_t* h = g + cells;
    const unsigned grid = (unsigned)std::min<size_t>((cells + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(active_cells_kernel, dim3(grid), dim3(256), 0, stream, dense, act, N, density_scale, step_max);
    hipLaunchKernelGGL((edt_pass_kernel<0>), dim3(grid), dim3(256), 0, stream, act, (const uint16_t*)nullptr, g, C);
    hipLaunchKernelGGL((edt_pass_kernel<1>), dim3(grid), dim3(256), 0, stream, act, g, h, C);
    hipLaunchKernelGGL((edt_pass_kernel<2>), dim3(grid), dim3(256), 0, stream, act, h, g, C);
    hipLaunchKernelGGL(nibble_kernel, dim3((unsigned)(nb * nb * nb)), dim3(64), 0, stream, g, nib, C, nb);
    return hipGetLastError();
}

hipError_t launch_cube_table(const float* dense, uint8_t* table, uint8_t* scratch, int N, int nb, hipStream_t stream) {
    const int n = nb * nb * nb;
    hipLaunchKernelGGL(cube_seed_kernel, dim3((unsigned)n), dim3(64), 0, stream, dense, table, N, nb);
    return dilate_table(table, scratch, nb, stream);
}

static hipError_t dilate_table(uint8_t* table, uint8_t* scratch, int nb, hipStream_t stream) {
    const int n = nb * nb * nb;
    uint8_t* cur = table;
    uint8_t* nxt = scratch;
    const int rounds = nb < 254 ? nb : 254;
    for (int k = 1; k <= rounds; k++) {
        hipLaunchKernelGGL(skip_dilate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cur, nxt, nb, k);
        uint8_t* t = cur;
        cur = nxt;
        nxt = t;
    }
    if (cur != table) {
        hipError_t e = hipMemcpyAsync(table, cur, (size_t)n, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

hipError_t launch_voxelize(const void* frames, size_t n_frames, float* density, uint8_t* material, int N, float cell, float extent,
                           float threshold, hipStream_t stream) {
    const size_t count = (size_t)N * N * N;
    /* background: VVoxel{material 0, density 2*extent} (VolumeConverter.cpp:51-55) */
    hipLaunchKernelGGL(voxelize_fill_kernel, dim3(2048), dim3(256), 0, stream, reinterpret_cast<int*>(density), count,
                       vrt_vox::ordered_key(extent * 2.f));
    if (n_frames > 0)
        hipLaunchKernelGGL(voxelize_kernel, dim3((unsigned)n_frames), dim3(256), 0, stream,
                           reinterpret_cast<const vrt_vox::TriangleFrame*>(frames), reinterpret_cast<int*>(density), N, cell, extent, threshold);
    hipLaunchKernelGGL(voxelize_finish_kernel, dim3(2048), dim3(256), 0, stream, density, material, count);
    return hipGetLastError();
}

hipError_t launch_split_voxels(const void* voxels, float* density, uint8_t* material, size_t count, hipStream_t stream) {
    hipLaunchKernelGGL(split_voxels_kernel, dim3(2048), dim3(256), 0, stream,
                       reinterpret_cast<const uint2*>(voxels), density, material, count);
    return hipGetLastError();
}

}  // namespace vrt

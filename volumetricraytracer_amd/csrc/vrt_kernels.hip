/*
 * vrt_kernels.hip — gfx950 (CDNA4) kernels of the volumetric SDF ray-marcher.
 *
 * One wavefront lane per primary ray, one 64-lane wave per 8x8 pixel tile, 4 waves per
 * workgroup (16x16 pixels).  The march is a sphere-trace over the trilinear interpolant of
 * the density grid; taps come either from the dense grid or from 4^3-cell bricks
 * (5^3 samples, 512 B = four 128-B lines) so that the 8 taps of a sample share one brick.
 * Normal, shadow test, shading, cube-map lookup and tone-map stay in registers.
 * Memory-bound gather work: no MFMA.
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
#include "vrt_device.h"
#include "vrt_launch.h"

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

struct VolRef {
    gfloat_p p; /* dense grid or brick pool, by PATH */
    int N;
    int nb;
    float extent, inv_cell, dscale, step_max;
};

template <int PATH>
__device__ __forceinline__ VolRef load_vol(const DVolume* __restrict__ v) {
    VolRef r;
    r.p = (gfloat_p)((PATH == VRT_PATH_DENSE) ? v->dense : v->bricks);
    r.N = v->N;
    r.nb = v->nb;
    r.extent = v->extent;
    r.inv_cell = v->inv_cell;
    r.dscale = v->density_scale;
    r.step_max = v->step_max;
    return r;
}

/* The 8 corner taps of one cell, in the order the lerp tree consumes them. */
struct Taps {
    float y00a, y00b, y01a, y01b, y10a, y10b, y11a, y11b; /* (x,z) = 00,01,10,11; a = y, b = y+1 */
};

typedef const char __attribute__((address_space(1))) * gchar_p;

/* a*b + c with 24-bit unsigned operands (full-rate v_mad_u32_u24; v_mul_lo_u32 is quarter rate). */
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

/* Fetch the taps of cell (cx,cy,cz).  Offsets are unsigned 32-bit byte offsets from the volume
 * base (≤ 4 GiB pools), built with 24-bit multiplies (cells < 2^10, bricks < 2^24). */
template <int PATH>
__device__ __forceinline__ Taps fetch8(const VolRef& V, int cx, int cy, int cz) {
    Taps t;
    if constexpr (PATH == VRT_PATH_DENSE) {
        const unsigned N = (unsigned)V.N;
        const unsigned off = mad24(mad24((unsigned)cx, N, (unsigned)cz), N, (unsigned)cy);
        const gfloat_p b = (gfloat_p)((gchar_p)V.p + ((size_t)off << 2));
        const unsigned NN = N * N;
        t.y00a = b[0];
        t.y00b = b[1];
        t.y01a = b[N];
        t.y01b = b[N + 1];
        t.y10a = b[NN];
        t.y10b = b[NN + 1];
        t.y11a = b[NN + N];
        t.y11b = b[NN + N + 1];
    } else {
        const unsigned nb = (unsigned)V.nb;
        const unsigned brick = mad24(mad24((unsigned)cx >> 2, nb, (unsigned)cz >> 2), nb, (unsigned)cy >> 2);
        const unsigned local = ((unsigned)cx & 3u) * 25u + ((unsigned)cz & 3u) * 5u + ((unsigned)cy & 3u);
        const unsigned off = ((brick << 7) + local) << 2; /* bytes */
        const gfloat_p b = (gfloat_p)((gchar_p)V.p + off);
        t.y00a = b[0];
        t.y00b = b[1];
        t.y01a = b[5];
        t.y01b = b[6];
        t.y10a = b[25];
        t.y10b = b[26];
        t.y11a = b[30];
        t.y11b = b[31];
    }
    return t;
}

/* Trilinear interpolant from the taps: y-lerps, z-lerps, x-lerp (each lerp is one sub + one fma). */
__device__ __forceinline__ float lerp8(const Taps& t, float fx, float fy, float fz) {
    float a00 = lerp1(t.y00a, t.y00b, fy);
    float a01 = lerp1(t.y01a, t.y01b, fy);
    float a10 = lerp1(t.y10a, t.y10b, fy);
    float a11 = lerp1(t.y11a, t.y11b, fy);
    float c0 = lerp1(a00, a01, fz);
    float c1 = lerp1(a10, a11, fz);
    return lerp1(c0, c1, fx);
}

template <int PATH>
__device__ __forceinline__ float trilinear(const VolRef& V, int cx, int cy, int cz, float fx, float fy, float fz) {
    return lerp8(fetch8<PATH>(V, cx, cy, cz), fx, fy, fz);
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

/* Diagnostic-build accumulators (wave-uniform, shader-clock cycles); unused otherwise. */
struct DiagAcc {
    unsigned long long mem = 0;   /* address ready → interpolated value available (loads + lerps) */
    unsigned long long loop = 0;  /* whole march-loop iterations */
    unsigned iters = 0;           /* iterations this wave executed (any lane active) */
};
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

/*
 * Sphere-trace one instance.  o,d: world-space ray (d normalised).  Returns true on hit and
 * the ray parameter (shared by world and object space — the object-space direction is not
 * re-normalised, DXR semantics).  NORMAL: also produce the world-space normal.
 */
template <int PATH, bool NORMAL, bool DIAG = false>
__device__ __forceinline__ bool march_instance(const DFrame& F, const DInstance* __restrict__ I,
                                               const DVolume* __restrict__ Vd, F3 o, F3 d, float t_cur, float t_base,
                                               float& t_hit, F3& n_world, unsigned& steps, DiagAcc* dg = nullptr) {
    const VolRef V = load_vol<PATH>(Vd);
    F3 rel = f3(o.x - I->pos[0], o.y - I->pos[1], o.z - I->pos[2]);
    F3 oo = mul33(I->w2o, rel);
    F3 od = mul33(I->w2o, d);
    float t_enter, t_exit;
    if (!slab(oo, od, V.extent, t_cur, t_enter, t_exit)) return false;

    const float inv_len = 1.0f / sqrtf(dot3(od, od));
    const float ds = V.dscale * inv_len;
    const float smax = V.step_max * inv_len;
    const F3 uo = f3((oo.x + V.extent) * V.inv_cell, (oo.y + V.extent) * V.inv_cell, (oo.z + V.extent) * V.inv_cell);
    const F3 ud = od * V.inv_cell;
    const float cmax = (float)(V.N - 2);

    float t = (t_enter > 0.0f ? t_enter : 0.0f) + F.eps_in;
    const float t_end = minf_(t_exit, t_cur);
    /* smallest step: one pixel-footprint radius at the total path length t_base + t */
    const float base_min = __builtin_fmaf(t_base, F.cone_eps, F.step_min);
    const int max_steps = F.max_steps;
    bool hit = false;
    int i = 0;
    int cx = 0, cy = 0, cz = 0;
    float fx = 0.0f, fy = 0.0f, fz = 0.0f;
    for (; i < max_steps; i++) {
        if (t > t_end) break;
        unsigned long long st0 = 0, st1 = 0;
        if constexpr (DIAG) st0 = stamp();
        const float ux = __builtin_fmaf(ud.x, t, uo.x);
        const float uy = __builtin_fmaf(ud.y, t, uo.y);
        const float uz = __builtin_fmaf(ud.z, t, uo.z);
        /* cell = clamp(floor(u), 0, N-2): v_med3_f32 (u is finite) */
        const float cxf = __builtin_amdgcn_fmed3f(floorf(ux), 0.0f, cmax);
        const float cyf = __builtin_amdgcn_fmed3f(floorf(uy), 0.0f, cmax);
        const float czf = __builtin_amdgcn_fmed3f(floorf(uz), 0.0f, cmax);
        fx = ux - cxf;
        fy = uy - cyf;
        fz = uz - czf;
        cx = (int)cxf;
        cy = (int)cyf;
        cz = (int)czf;
        if constexpr (DIAG) {
            asm volatile("" ::"v"(cx), "v"(cy), "v"(cz), "v"(fx), "v"(fy), "v"(fz));
            st1 = stamp();
        }
        const Taps taps = fetch8<PATH>(V, cx, cy, cz);
        if constexpr (DIAG) {
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(taps.y00a), "v"(taps.y00b), "v"(taps.y01a), "v"(taps.y01b), "v"(taps.y10a),
                         "v"(taps.y10b), "v"(taps.y11a), "v"(taps.y11b));
            const unsigned long long st2 = stamp();
            dg->mem += st2 - st1; /* address arithmetic + 4 loads until the data is back */
            dg->iters++;
            dg->loop += st2 - st0;
        }
        const float s = lerp8(taps, fx, fy, fz) * ds;
        steps++;
        if (s < __builtin_fmaf(t, F.cone_eps, F.eps_hit)) {
            hit = true;
            break;
        }
        const float adv_min = __builtin_fmaf(t, F.cone_eps, base_min);
        t = t + __builtin_fmaxf(__builtin_fminf(s * F.k_relax, smax), adv_min);
    }
    if (!hit) return false;
    t_hit = t;
    if constexpr (NORMAL) {
        F3 n;
        if (i == 0 && t_enter >= 0.0f) {
            /* surface cut by the volume boundary: AABB-face normal (Raytracing.hlsl:198-226) */
            float tb = t_enter - 0.1f;
            float rx = __builtin_fmaf(od.x, tb, oo.x);
            float ry = __builtin_fmaf(od.y, tb, oo.y);
            float rz = __builtin_fmaf(od.z, tb, oo.z);
            float e = V.extent;
            n.x = rx > e ? 1.0f : (rx < -e ? -1.0f : 0.0f);
            n.y = ry > e ? 1.0f : (ry < -e ? -1.0f : 0.0f);
            n.z = rz > e ? 1.0f : (rz < -e ? -1.0f : 0.0f);
        } else {
            /* central differences of the interpolant one cell either side (Voxel.hlsli:783-804) */
            const int N2 = V.N - 2;
            int xp = cx + 1 > N2 ? N2 : cx + 1, xm = cx - 1 < 0 ? 0 : cx - 1;
            int yp = cy + 1 > N2 ? N2 : cy + 1, ym = cy - 1 < 0 ? 0 : cy - 1;
            int zp = cz + 1 > N2 ? N2 : cz + 1, zm = cz - 1 < 0 ? 0 : cz - 1;
            n.x = trilinear<PATH>(V, xp, cy, cz, fx, fy, fz) - trilinear<PATH>(V, xm, cy, cz, fx, fy, fz);
            n.y = trilinear<PATH>(V, cx, yp, cz, fx, fy, fz) - trilinear<PATH>(V, cx, ym, cz, fx, fy, fz);
            n.z = trilinear<PATH>(V, cx, cy, zp, fx, fy, fz) - trilinear<PATH>(V, cx, cy, zm, fx, fy, fz);
        }
        float l2 = dot3(n, n);
        if (!(l2 > 0.0f)) {
            n = f3(0.0f, 0.0f, 0.0f);
        } else {
            n = n * fast_rsq(l2);
        }
        n_world = mul33(I->o2w, n);
    }
    return true;
}

/* Closest hit over the scene.  SINGLE: exactly one instance, no BVH, all scene data wave-uniform. */
template <int PATH, bool SINGLE, bool DIAG = false>
__device__ __forceinline__ bool trace_closest(const DFrame& F, F3 o, F3 d, float t_max, float t_base, float& t_best,
                                              int& inst_best, F3& n_best, unsigned& steps, DiagAcc* dg = nullptr) {
    if constexpr (SINGLE) {
        float t;
        F3 n;
        if (march_instance<PATH, true, DIAG>(F, F.inst, F.vols + F.inst->slot, o, d, t_max, t_base, t, n, steps, dg)) {
            t_best = t;
            inst_best = 0;
            n_best = n;
            return true;
        }
        return false;
    } else {
        bool any = false;
        float best = t_max;
        int stack[16];
        int sp = 0;
        if (F.n_nodes > 0) stack[sp++] = 0;
        while (sp > 0) {
            const DBvhNode nd = F.nodes[stack[--sp]];
            if (!slab_box(o, d, nd, best)) continue;
            if (nd.left < 0) {
                const int ii = -nd.left - 1;
                const DInstance* I = F.inst + ii;
                float t;
                F3 n;
                if (march_instance<PATH, true, DIAG>(F, I, F.vols + I->slot, o, d, best, t_base, t, n, steps, dg)) {
                    if (!any || t < best || (t == best && ii < inst_best)) {
                        any = true;
                        best = t;
                        t_best = t;
                        inst_best = ii;
                        n_best = n;
                    }
                }
            } else {
                stack[sp++] = nd.right;
                stack[sp++] = nd.left;
            }
        }
        return any;
    }
}

template <int PATH, bool SINGLE, bool DIAG = false>
__device__ __forceinline__ bool trace_any(const DFrame& F, F3 o, F3 d, float t_max, float t_base, unsigned& steps,
                                          DiagAcc* dg = nullptr) {
    float t;
    F3 n;
    if constexpr (SINGLE) {
        return march_instance<PATH, false, DIAG>(F, F.inst, F.vols + F.inst->slot, o, d, t_max, t_base, t, n, steps, dg);
    } else {
        int stack[16];
        int sp = 0;
        if (F.n_nodes > 0) stack[sp++] = 0;
        while (sp > 0) {
            const DBvhNode nd = F.nodes[stack[--sp]];
            if (!slab_box(o, d, nd, t_max)) continue;
            if (nd.left < 0) {
                const DInstance* I = F.inst + (-nd.left - 1);
                if (march_instance<PATH, false, DIAG>(F, I, F.vols + I->slot, o, d, t_max, t_base, t, n, steps, dg)) return true;
            } else {
                stack[sp++] = nd.right;
                stack[sp++] = nd.left;
            }
        }
        return false;
    }
}

/* Cube-map point lookup with SampleLevel(dir.xzy) (Raytracing.hlsl:444-449). */
__device__ __forceinline__ F3 env_lookup(const uint8_t* __restrict__ env, int S, F3 dir) {
    if (env == nullptr) return f3(0.0f, 0.0f, 0.0f);
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
    float u = (sc / ma + 1.0f) * 0.5f;
    float v = (tc / ma + 1.0f) * 0.5f;
    int ix = (int)floorf(u * (float)S);
    int iy = (int)floorf(v * (float)S);
    ix = ix < 0 ? 0 : (ix > S - 1 ? S - 1 : ix);
    iy = iy < 0 ? 0 : (iy > S - 1 ? S - 1 : iy);
    const unsigned px = ((guint_p)env)[(face * S + iy) * S + ix]; /* little-endian R,G,B,A bytes */
    const float k = 1.0f / 255.0f;
    return f3((float)(px & 0xffu) * k, (float)((px >> 8) & 0xffu) * k, (float)((px >> 16) & 0xffu) * k);
}

/* Radiance(), Lighting.hlsli:50-101 (F enters twice, PI = 3.141592f as in Constants.hlsli). */
__device__ __forceinline__ F3 radiance(F3 Li, F3 wi, F3 wo, F3 n, F3 albedo, float rough, float metal, float k) {
    const float PI_REF = 3.141592f;
    const float INV_PI_REF = 1.0f / 3.141592f;
    F3 hv = wi + wo;
    F3 h = hv * fast_rsq(dot3(hv, hv));
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
    F3 brdf = f3((albedo.x * INV_PI_REF) * kd.x + cook.x * Fr.x, (albedo.y * INV_PI_REF) * kd.y + cook.y * Fr.y,
                 (albedo.z * INV_PI_REF) * kd.z + cook.z * Fr.z);
    float ndwi = dot3(n, wi);
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
 * Primary-ray kernel.  One workgroup = one 16x16-pixel tile.  blockIdx → tile map (F.tile_map):
 *   SUPERTILE (default): tiles are grouped into 4x4-tile supertiles (64x64 pixels); supertile s
 *     goes to XCD s % 8 (workgroups are dealt round-robin over the 8 XCDs, so the blocks
 *     b ≡ k (mod 8) share XCD k's L2).  An XCD's 16 consecutive blocks cover one supertile, so its
 *     L2 sees the bricks of a compact screen region, while heavy (object) and cheap (sky)
 *     regions are spread over all 8 XCDs.
 *   BAND: XCD k gets the k-th contiguous eighth of the tiles (best L2 locality, worst balance).
 *   LINEAR: tile = blockIdx (consecutive tiles on different XCDs).
 * Placement only affects speed, never results.
 */
template <int PATH, bool SINGLE, bool DIAG>
__global__ __launch_bounds__(kBlockThreads) void march_kernel(const DFrame F) {
    unsigned long long t_start = 0;
    if constexpr (DIAG) t_start = __builtin_amdgcn_s_memrealtime(); /* 100 MHz; diagnostic build only */
    const int nblk = (int)gridDim.x;
    const int b = (int)blockIdx.x;
    int tile_x, tile_y;
    if (F.tile_map == kMapSupertile) {
        const int xcd = b & 7, q = b >> 3;
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
    const int wave = (int)threadIdx.x >> 6;
    const int lane = (int)threadIdx.x & 63;
    const int px = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
    const int pyl = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const int py = F.row0 + pyl;
    const bool valid = tile_x < F.tiles_x && px < F.width && pyl < F.rows;

    unsigned n_primary = 0, n_shadow = 0, n_bounce = 0, s_primary = 0, s_shadow = 0, n_hits = 0;
    DiagAcc dg;

    if (valid) {
        /* camera ray (Ray.hlsli:36-48, then normalised) */
        float sx = (((float)px + 0.5f) / (float)F.width) * 2.0f - 1.0f;
        float sy = (((float)py + 0.5f) / (float)F.height) * 2.0f - 1.0f;
        float tx = sx * F.cx;
        float ty = (-sy) * F.cy;
        F3 d = normalize3(f3((tx * F.r0[0] + ty * F.r1[0]) - F.r2[0], (tx * F.r0[1] + ty * F.r1[1]) - F.r2[1],
                             (tx * F.r0[2] + ty * F.r1[2]) - F.r2[2]));
        F3 o = f3(F.cam_o[0], F.cam_o[1], F.cam_o[2]);
        n_primary = 1;

        float t_hit = 0.0f;
        int inst = 0;
        F3 n = f3(0.0f, 0.0f, 0.0f);
        F3 color;
        if (trace_closest<PATH, SINGLE, DIAG>(F, o, d, 10000.0f, 0.0f, t_hit, inst, n, s_primary, &dg)) {
            n_hits = 1;
            const DVolume* V = F.vols + F.inst[inst].slot;
            F3 albedo = f3(V->tint[0], V->tint[1], V->tint[2]);
            if (F.unlit) {
                color = albedo;
            } else {
                F3 hp = f3(__builtin_fmaf(d.x, t_hit, o.x), __builtin_fmaf(d.y, t_hit, o.y),
                           __builtin_fmaf(d.z, t_hit, o.z));
                F3 so = f3(hp.x - d.x * 0.1f, hp.y - d.y * 0.1f, hp.z - d.z * 0.1f);
                F3 wo = f3(-d.x, -d.y, -d.z);
                F3 ld = f3(F.light_dir[0], F.light_dir[1], F.light_dir[2]);
                bool shadowed = false;
                if (F.shadow) {
                    n_shadow = 1;
                    shadowed = trace_any<PATH, SINGLE, DIAG>(F, so, ld, 5000.0f, t_hit, s_shadow, &dg);
                }
                color = f3(0.0f, 0.0f, 0.0f);
                if (!shadowed) {
                    F3 Li = f3(F.light_strength, F.light_strength, F.light_strength);
                    color = color + radiance(Li, ld, wo, n, albedo, V->roughness, V->metallic, V->k);
                }
            }
        } else {
            color = env_lookup(F.env, F.env_size, d);
        }
        float4 outp = make_float4(tonemap(color.x), tonemap(color.y), tonemap(color.z), 1.0f);
        reinterpret_cast<float4*>(F.out)[(size_t)pyl * F.width + px] = outp;
    }

    /* Statistics (algorithmic-byte accounting): wave shuffle-reduce, then one 32-byte
       record per WAVE with plain stores.  No atomics (6 same-address atomics per wave serialise at
       ~12 ns each at the memory side and cost more than the march itself) and no workgroup
       barrier (it would pin the three fast waves of a tile until its slowest wave retires). */
    const unsigned s_primary_lane = s_primary, s_shadow_lane = s_shadow;
    n_primary = wave_sum(n_primary);
    n_shadow = wave_sum(n_shadow);
    n_bounce = wave_sum(n_bounce);
    s_primary = wave_sum(s_primary);
    s_shadow = wave_sum(s_shadow);
    n_hits = wave_sum(n_hits);
    unsigned max_iter = 0, d_mem = 0, d_loop = 0, d_iters = 0;
    if constexpr (DIAG) {
        /* the accumulators are per lane (each lane only counts iterations it was active in): report
           the lane with the longest chain of dependent samples, i.e. the wave's critical path */
        max_iter = s_primary_lane + s_shadow_lane;
        unsigned long long key = ((unsigned long long)dg.iters << 8) | (unsigned)lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            unsigned other = __shfl_xor(max_iter, o);
            max_iter = other > max_iter ? other : max_iter;
            unsigned long long ok = __shfl_xor(key, o);
            key = ok > key ? ok : key;
        }
        const int src = (int)(key & 0xff);
        d_mem = __shfl((unsigned)dg.mem, src);
        d_loop = __shfl((unsigned)dg.loop, src);
        d_iters = __shfl(dg.iters, src);
    }
    if (F.stats != nullptr && lane < 8) {
        unsigned v = lane == 0 ? n_primary : lane == 1 ? n_shadow : lane == 2 ? n_bounce : lane == 3 ? s_primary
                   : lane == 4 ? s_shadow : lane == 5 ? n_hits : 0u;
        F.stats[((size_t)b * 4 + wave) * kStatRecord + lane] = v;
    }
    if constexpr (DIAG) {
        /* diagnostic timeline record: where and when this wave ran and where its march cycles went
           (never in the production kernel; stamp values leave only through this buffer) */
        if (F.diag_buf != nullptr && lane < kDiagRecord) {
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
            const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4); /* HW_REG_HW_ID */
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   /* HW_REG_XCC_ID[3:0] */
            unsigned v = 0;
            switch (lane) {
                case 0: v = (unsigned)t_start; break;
                case 1: v = (unsigned)t_end; break;
                case 2: v = hw_id; break;
                case 3: v = xcc; break;
                case 4: v = max_iter; break;
                case 5: v = d_mem; break;
                case 6: v = d_loop; break;
                case 7: v = d_iters; break;
                default: break;
            }
            F.diag_buf[((size_t)b * 4 + wave) * kDiagRecord + lane] = v;
        }
    }
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

/* ---- launch wrappers (host) -------------------------------------------------------------- */

template <int PATH, bool SINGLE>
static hipError_t launch_t(const DFrame& F, hipStream_t stream) {
    const int grid = grid_blocks(F.tiles_x, F.tiles_y, F.tile_map);
    if (grid <= 0) return hipSuccess;
    if (F.diag)
        hipLaunchKernelGGL((march_kernel<PATH, SINGLE, true>), dim3((unsigned)grid), dim3(kBlockThreads), 0, stream, F);
    else
        hipLaunchKernelGGL((march_kernel<PATH, SINGLE, false>), dim3((unsigned)grid), dim3(kBlockThreads), 0, stream, F);
    return hipGetLastError();
}

hipError_t launch_march(const DFrame& F, int path, bool single, hipStream_t stream) {
    if (path == VRT_PATH_DENSE) return single ? launch_t<VRT_PATH_DENSE, true>(F, stream) : launch_t<VRT_PATH_DENSE, false>(F, stream);
    return single ? launch_t<VRT_PATH_BRICK, true>(F, stream) : launch_t<VRT_PATH_BRICK, false>(F, stream);
}

hipError_t launch_retile(const float* dense, float* bricks, int N, int nb, hipStream_t stream) {
    hipLaunchKernelGGL(retile_bricks_kernel, dim3((unsigned)(nb * nb * nb)), dim3(128), 0, stream, dense, bricks, N, nb);
    return hipGetLastError();
}

hipError_t launch_split_voxels(const void* voxels, float* density, uint8_t* material, size_t count, hipStream_t stream) {
    hipLaunchKernelGGL(split_voxels_kernel, dim3(2048), dim3(256), 0, stream,
                       reinterpret_cast<const uint2*>(voxels), density, material, count);
    return hipGetLastError();
}

}  // namespace vrt

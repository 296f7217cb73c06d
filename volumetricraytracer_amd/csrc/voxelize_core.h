/*
 * voxelize_core.h — the arithmetic of the Voxelizer's hot loop, once, for both builds of it:
 * the CPU converter (csrc/host/VolumeConverter.cpp, g++) and the HIP kernel (vrt_kernels.hip, hipcc).
 *
 * Restates Voxelizer/Private/VolumeConverter.cpp:161-252 (VoxelizeFace), :681-701 (triangle bounding box →
 * voxel index box) and :703-781 (the 7-region point/triangle classification) of the reference.  Plain
 * floats, explicit operation order, no fused multiply-add on either side (g++ targets baseline x86-64,
 * hipcc builds with -ffp-contract=off), correctly rounded sqrt and division: the two builds produce the
 * same bits.
 */
#ifndef VRT_VOXELIZE_CORE_H
#define VRT_VOXELIZE_CORE_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VRT_HD __host__ __device__ inline
#else
#define VRT_HD inline
#endif

namespace vrt_vox {

struct V3 {
    float x, y, z;
};
VRT_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
VRT_HD V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
VRT_HD V3 add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
VRT_HD V3 scale(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
VRT_HD V3 divs(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
VRT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VRT_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
VRT_HD float length(V3 a) { return sqrtf(dot(a, a)); }

/* Everything about one triangle that the per-voxel classification needs, plus its voxel index box. */
struct TriangleFrame {
    V3 v[3];        /* V1, V2, V3 */
    V3 normal;      /* A: unit face normal */
    V3 along[3];    /* unit edge directions  B: V1→V3, C: V3→V2, D: V2→V1 */
    float length[3];/* |B|, |C|, |D| */
    V3 inward[3];   /* in-plane unit normals of the edges, pointing into the triangle: E (of B), F (of C), G (of D) */
    int32_t lo[3], hi[3]; /* voxel index box, inclusive, already clipped to the volume */
};

/* false for a degenerate triangle (the reference would hit its unreachable assert, VolumeConverter.cpp:779) */
VRT_HD bool make_frame(V3 v1, V3 v2, V3 v3_, TriangleFrame& t) {
    t.v[0] = v1;
    t.v[1] = v2;
    t.v[2] = v3_;
    const V3 n = cross(sub(v2, v1), sub(v3_, v1));
    const float area2 = length(n);
    if (!(area2 > 0.f)) return false;
    t.normal = divs(n, area2);
    const V3 e[3] = {sub(v3_, v1), sub(v2, v3_), sub(v1, v2)};
    for (int k = 0; k < 3; k++) {
        t.length[k] = length(e[k]);
        if (!(t.length[k] > 0.f)) return false;
        t.along[k] = divs(e[k], t.length[k]);
        const V3 c = cross(t.along[k], t.normal);
        t.inward[k] = divs(c, length(c));
    }
    return true;
}

/* roundf(v) as an int held to [lo, hi] (a float beyond int's range — a damaged file's vertex — must not reach the conversion:
 * that is undefined on the host and saturates on the device); NaN gives if_nan. */
VRT_HD int clamped_round(float v, int lo, int hi, int if_nan) {
    const float r = roundf(v);
    return r != r ? if_nan : r <= (float)lo ? lo : r >= (float)hi ? hi : (int)r;
}

/* Index box of the voxels a triangle can influence: triangle bounds, grown by the threshold, rounded to
 * voxels, grown by one voxel, clipped (GetTriangleBoundingBox + GetVoxelizedBoundingBox,
 * VolumeConverter.cpp:681-701; VVoxelVolume::RelativePositionToVoxelIndex rounds, VoxelVolume.cpp:148-161). */
VRT_HD void index_box(TriangleFrame& t, float threshold, float extent, float cell, int n_axis) {
    const V3 lo = v3(fminf(t.v[0].x, fminf(t.v[1].x, t.v[2].x)), fminf(t.v[0].y, fminf(t.v[1].y, t.v[2].y)),
                     fminf(t.v[0].z, fminf(t.v[1].z, t.v[2].z)));
    const V3 hi = v3(fmaxf(t.v[0].x, fmaxf(t.v[1].x, t.v[2].x)), fmaxf(t.v[0].y, fmaxf(t.v[1].y, t.v[2].y)),
                     fmaxf(t.v[0].z, fmaxf(t.v[1].z, t.v[2].z)));
    const V3 half = scale(sub(hi, lo), 0.5f);
    const V3 centre = add(half, lo);
    const V3 ext = v3(fabsf(half.x), fabsf(half.y), fabsf(half.z));
    const V3 bmin = sub(sub(centre, ext), v3(threshold, threshold, threshold));
    const V3 bmax = add(add(centre, ext), v3(threshold, threshold, threshold));
    const float org = -1.0f * extent; /* volume origin on every axis: -ONE * VolumeExtends */
    const float mn[3] = {bmin.x, bmin.y, bmin.z}, mx[3] = {bmax.x, bmax.y, bmax.z};
    for (int a = 0; a < 3; a++) {
        /* held to one voxel beyond the clip below: the clipped box — empty when the triangle lies outside the volume — is unchanged */
        int i0 = clamped_round((mn[a] - org) / cell, -1, n_axis + 1, n_axis + 1) - 1;
        int i1 = clamped_round((mx[a] - org) / cell, -2, n_axis, -2) + 1;
        t.lo[a] = i0 < 0 ? 0 : i0;
        t.hi[a] = i1 > n_axis - 1 ? n_axis - 1 : i1;
    }
}

/* Position of voxel (x,y,z): index * cell - extent (VoxelVolume.cpp:139-146). */
VRT_HD V3 voxel_position(int x, int y, int z, float cell, float extent) {
    const float org = -1.0f * extent;
    return v3((float)x * cell + org, (float)y * cell + org, (float)z * cell + org);
}

/* Distance from p to the triangle, by the region p projects into (face, 3 edges, 3 vertices). */
VRT_HD float region_distance(const TriangleFrame& t, V3 p) {
    const V3 r1 = sub(p, t.v[0]), r2 = sub(p, t.v[1]), r3 = sub(p, t.v[2]);
    const float a = dot(r1, t.normal);                             /* signed plane distance */
    const float b = dot(r1, t.along[0]), e = dot(r1, t.inward[0]); /* edge V1→V3 */
    const float c = dot(r3, t.along[1]), f = dot(r3, t.inward[1]); /* edge V3→V2 */
    const float d = dot(r2, t.along[2]), g = dot(r2, t.inward[2]); /* edge V2→V1 */
    if (e >= 0.f && f >= 0.f && g >= 0.f) return fabsf(a);                     /* R1: over the face */
    if (d >= t.length[2] && b <= 0.f) return length(r1);                       /* R5: vertex V1 */
    if (b >= t.length[0] && c <= 0.f) return length(r3);                       /* R7: vertex V3 */
    if (c >= t.length[1] && d <= 0.f) return length(r2);                       /* R6: vertex V2 */
    if (g <= 0.f && d >= 0.f && d <= t.length[2]) return sqrtf(a * a + g * g); /* R2: edge V2→V1 */
    if (e <= 0.f && b >= 0.f && b <= t.length[0]) return sqrtf(a * a + e * e); /* R4: edge V1→V3 */
    if (f <= 0.f && c >= 0.f && c <= t.length[1]) return sqrtf(a * a + f * f); /* R3: edge V3→V2 */
    /* numerically between regions (the reference asserts here): nearest of the three vertices */
    return fminf(length(r1), fminf(length(r2), length(r3)));
}

/* dist → density: 1 - dist/thr, negated, + 0.5 (VolumeConverter.cpp:200-202): -0.5 on the triangle, 0 at thr/2. */
VRT_HD float shell_density(float dist, float threshold) {
    float density = 1.f - (dist / threshold);
    density = -1.f * density + 0.5f;
    return density;
}

/* Order-preserving map float → int32 (and back: it is an involution) so that the minimum over triangles
 * can be taken with an integer atomicMin. */
VRT_HD int32_t ordered_key(float f) {
    union { float f; int32_t i; } u;
    u.f = f;
    return u.i ^ ((u.i >> 31) & 0x7fffffff);
}
VRT_HD float from_ordered_key(int32_t k) {
    union { float f; int32_t i; } u;
    u.i = k ^ ((k >> 31) & 0x7fffffff);
    return u.f;
}

}  // namespace vrt_vox

#endif

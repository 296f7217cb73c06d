/*
 * vrt_oracle.h — C interface of the scalar CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (volumetricraytracer_amd/, the C-ABI
 * library) may include, link or call this.  Allowed users: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.
 *
 * Parity status: PARITY UNPINNED by the reference's own tests — the reference
 * (Elyptos/VolumetricRaytracer) ships no tests, golden images or known-answer vectors
 *  and its hot path (HLSL/DXR + D3D12) cannot be built or run here.
 * The oracle is therefore pinned by analytic ground truth (ray∩sphere, plane, box) and by
 * a restatement of the reference's own per-cell cubic iso-surface solve (vrto_ref_hit_t,
 * Voxel.hlsli:552-605,691-781) and, since round 4, by whole frames rendered with the reference's own
 * intersection (vrto_ref_render) — see tests/test_oracle_pins.py, tests/test_reference_pixels.py.
 */
#ifndef VRT_ORACLE_H
#define VRT_ORACLE_H

#include <stdint.h>
#include "../include/vrt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 2D material texture, R8G8B8A8_UNORM, row-major, mip 0 (DXTexture2D.cpp:78-81); rgba8 NULL = unbound. */
typedef struct vrto_texture {
    const uint8_t* rgba8;
    int32_t width, height;
} vrto_texture;

typedef struct vrto_volume {
    const float* density;     /* N^3, index x*N*N + z*N + y; NULL = empty slot */
    int32_t resolution;       /* N = 2^resolution + 1 */
    float extent;
    float density_scale;
    float step_max;           /* <= 0: unbounded */
    vrt_material material;
    vrto_texture albedo_tex, normal_tex, rm_tex; /* VMaterial::{Albedo,Normal,RM}TexturePath, Material.h:29-31 */
    float tex_scale[2];                          /* VMaterial::TextureScale (default 100,100), Material.h:33 */
    int32_t format;                              /* vrt_volume_format: VRT_FORMAT_TEXEL16 marches the reference's 16-bit texel of `density` */
} vrto_volume;

typedef struct vrto_stats {
    uint64_t primary_rays, shadow_rays, bounce_rays;
    uint64_t primary_steps, shadow_steps, hits;
    uint64_t exhausted_rays; /* marches that ran out of budget inside the volume (vrt_timing::exhausted_rays) */
} vrto_stats;

/* Renders rows [row0,row0+rows) of the width x height frame into out_rgba (rows*width float4).
 * volumes: array of VRT_MAX_VOLUMES slots.  env: 6*face^2 RGBA8 or NULL.  threads >= 1. */
int vrto_render(const vrt_scene* scene, const vrto_volume* volumes,
                const uint8_t* env_rgba8, int env_face_size,
                const vrt_params* params, int row0, int rows,
                float* out_rgba, vrto_stats* stats_or_null, int threads);

/* Debug: when set (non-NULL), the next vrto_render calls also write, per pixel, the number of march positions (sampled
 * or skipped) its primary ray visited (low 16 bits) and the rays after it visited (high 16 bits) into img (rows*width):
 * the length of the dependent chain a GPU lane runs for that pixel (tests/chain_lengths.py).  Not thread-safe. */
void vrto_debug_set_steps_image(uint32_t* img);
/* Debug: likewise, per pixel the number of positions its primary ray SKIPPED (empty-space leaps) before its first sample — the part of
 * a lane's chain a per-tile beam pre-pass could take over (tests/chain_lengths.py beam).  Not thread-safe. */
void vrto_debug_set_lead_image(uint32_t* img);
/* Study only: while on, the camera ray's secondary rays back off 0.1 * |un-normalised camera direction| like the reference's (Ray.hlsli:44-45,
 * Raytracing.hlsl:52) instead of 0.1: how much the documented normalisation deviation moves pixels (tests/soak_reference_pixels.py offsets). */
void vrto_debug_unnormalised_offsets(int on);
/* Debug: while set, every march position of vrto_trace (single-threaded) appends {t, sample or NaN when skipped, leap, step
 * taken (negative: the over-relaxed march went back)} to records (4 floats each, at most capacity).  Returns the number of
 * records written since the previous call. */
int vrto_debug_set_position_log(float* records, int capacity);

/* Single-ray probes used by the analytic pins (world-space ray, direction is normalised
 * internally; returns 1 on hit and writes t / world normal / instance index). */
int vrto_trace(const vrt_scene* scene, const vrto_volume* volumes, const vrt_params* params,
               const float origin[3], const float dir[3], float t_max,
               float* t_out, float normal_out[3], int* instance_out, int* steps_out);

/* The same for n rays (origins, dirs: 3 floats each) with the scene set up once; hit_out[i] 0/1, t_out[i], normal_out (3n, may
 * be NULL). */
int vrto_trace_batch(const vrt_scene* scene, const vrto_volume* volumes, const vrt_params* params, int n, const float* origins,
                     const float* dirs, float t_max, uint8_t* hit_out, float* t_out, float* normal_out_or_null, int threads);

/* Camera ray of pixel (px,py): writes origin[3], dir[3] (normalised). */
void vrto_camera_ray(const vrt_scene* scene, int width, int height, int px, int py,
                     float origin[3], float dir[3]);

/* Trilinear sample of a volume at object-space position p (clamped to the grid). */
float vrto_sample(const vrto_volume* vol, const float p[3]);

/* Restatement of the REFERENCE's hit search for one object-space ray against one volume:
 * cell-by-cell DDA + closed-form cubic of the trilinear interpolant along the ray
 * (Shaders/Include/Voxel.hlsli:552-605, 691-781; loop Raytracing.hlsl:228-323 without the
 * octree skip).  Double precision.  Returns 1 and the ray parameter of the first zero
 * crossing, or 0.  Used to check that the sphere-trace converges to the reference's surface. */
int vrto_ref_hit_t(const vrto_volume* vol, const float origin[3], const float dir[3], double* t_out);

/* The same for n rays. */
int vrto_ref_hit_batch(const vrto_volume* vol, int n, const float* origins, const float* dirs, uint8_t* hit_out, double* t_out, int threads);

/* The frame as the REFERENCE's own intersection would produce it (SH/Raytracing.hlsl:147-442): per ray the exact first root of the
 * per-cell cubic (Voxel.hlsli:552-605, 691-781), the normal from GetNormal evaluated AT that root (Voxel.hlsli:783-804), the
 * AABB-face normal for a solid start cell (Raytracing.hlsl:198-226); camera ray, closest-hit shading, miss and tone-map as in
 * vrto_render.  Reads nothing of the sphere-trace's contract (eps_hit, cone_eps, k_relax, step_*, max_steps, tables), so
 * fixtures made with it do NOT move when that contract changes (tests/golden/ref_*.npz, tests/golden/make_ref_golden.py).
 * Interp modes only.  t_out_or_null (rows*width): camera-ray hit distance, -1 = miss. */
int vrto_ref_render(const vrt_scene* scene, const vrto_volume* volumes, const uint8_t* env_rgba8, int env_face_size,
                    const vrt_params* params, int row0, int rows, float* out_rgba, float* t_out_or_null, int threads);

/* The frame the reference's shaders compute, LITERALLY (oracle/vrt_ref_literal.inl) — vrto_ref_render is the IDEALISED restatement
 * (double precision, exact root, no nudges, no octree, no budget, normalised camera direction); this one follows
 * VRIntersection / VRIntersectionShadowRay statement by statement in fp32: the un-normalised camera direction (Ray.hlsli:36-48) with
 * every offset, TMax and the closest-hit shader's wo in its units; tEnter += 0.01 (Raytracing.hlsl:178,195); the collapsed octree's
 * leaf origin and size per step (Voxel.hlsli:293-495, Voxel/Private/Octree.cpp:70-107,181-262); GoToNextVoxel's +0.1 (Voxel.hlsli:80-128);
 * the cubic on [cellEnter, cellExit] with t0 = max(0, -tIn/(tOut-tIn)), the derivative-root split, 2 regula-falsi steps + 1 secant,
 * tHit > 0 (Voxel.hlsli:691-781); GetNormal with the abs() weights at a cell-space position that may exceed 1, texels outside the
 * 3D texture = 0 (Voxel.hlsli:607-684,783-804); 255 leaves, then the red unlit hit at t = 10 (Raytracing.hlsl:229,325-334); ReportHit's
 * [0, RayTCurrent] acceptance.  Volumes are read through the reference's 16-bit texel whatever `format` says.  It exists to MEASURE
 * how far the reference's frames are from the idealisation and from the HIP frames (tests/ref_pixels.py).
 * options: VRTO_LIT_*.  t_out_or_null: camera-ray hit distance in world units, -1 = miss.  Interp modes, resolution <= 8. */
#define VRTO_LIT_NORMALISED_CAMERA 1u /* study: the same shaders fed the NORMALISED camera direction (the product's documented deviation): \
                                         isolates the intersection's numerics from what the un-normalised direction does to offsets and shading */
typedef struct vrto_literal_stats {
    uint64_t rays;             /* (ray, instance) intersection-shader invocations */
    uint64_t iterations;       /* leaves visited */
    uint64_t solid_start_hits; /* accepted hits of a solid start cell (AABB-face normal) */
    uint64_t entry_hits;       /* accepted hits reported at the START of a leaf's search interval (the cubic is <= 0 there: a root in the
                                  first 0.1 of the cell that the previous cell's extrapolated cubic did not find) */
    uint64_t root_hits;        /* accepted hits at a root of the cubic */
    uint64_t tail_hits;        /* ... of which beyond the leaf's true exit: found on the extrapolated cubic, in the next cell */
    uint64_t red_hits;         /* accepted budget-exhaustion hits (unlit red, t = 10) */
    uint64_t rejected_reports; /* ReportHit outside [0, RayTCurrent]: the shader returns without a hit */
} vrto_literal_stats;
int vrto_ref_literal_render(const vrt_scene* scene, const vrto_volume* volumes, const uint8_t* env_rgba8, int env_face_size,
                            const vrt_params* params, int row0, int rows, float* out_rgba, float* t_out_or_null, unsigned options,
                            vrto_literal_stats* stats_or_null, int threads);

typedef struct vrto_octree_info {
    uint64_t nodes;              /* nodes of the collapsed tree */
    uint64_t leaves_at_depth[9]; /* depth 0 = the whole volume ... depth `resolution` = one cell */
    int32_t texture_edge;        /* 2 * ceil(cbrt(nodes)): the traversal texture's edge in texels */
    int32_t pointer_overflow;    /* 1: a child-block coordinate exceeds 255 and wraps in its 8-bit pointer texel (RDXVoxelVolume.cpp:282-284) */
} vrto_octree_info;
int vrto_literal_octree_info(const vrto_volume* vol, vrto_octree_info* out);

/* Debug: the two-level empty-space table the march uses for `vol` under its metric (step_max > 0) — skip_out: nb^3
 * Chebyshev brick distances D, nib_out: nb^3 words of sub-block nibbles — and (field_out, N^3 floats) the field the march
 * samples (the integer field +-q for VRT_FORMAT_TEXEL16).  Any pointer may be NULL. */
int vrto_debug_tables(const vrto_volume* vol, uint8_t* skip_out, uint32_t* nib_out, float* field_out);

/* Cube-map lookup used by the miss path (dir is a world direction; returns rgb). */
void vrto_env_lookup(const uint8_t* env_rgba8, int face_size, const float dir[3], float rgb_out[3]);

#ifdef __cplusplus
}
#endif
#endif

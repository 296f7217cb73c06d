/*
 * vrt.h — C-ABI of the MI355X-native volumetric SDF ray-marcher.
 *
 * This is the drop-in boundary for the reference's renderer hot path.  Everything the
 * reference's D3D12/DXR backend did behind `VRenderer` is reachable through these entry
 * points; a C++ adaptor with the `VRenderer` shape (volumetricraytracer_amd/csrc/host/
 * HipRenderer.h) is the only thing that calls them.  POD only, plain pointers and sizes.
 *
 * Reference interfaces replaced (paths relative to
 * /root/reference/VolumetricRaytracer/VolumetricRaytracer/):
 *   vrt_create / vrt_destroy        VDXRenderer::Start/Stop + SetupRenderer/DestroyRenderer
 *                                   Renderer/DX/Private/DXRenderer.cpp:204-316, 68-100
 *   vrt_volume_upload*              VDXVoxelVolume::UpdateFromVoxelVolume / UpdateVolumeTexture
 *                                   Renderer/DX/Private/RDXVoxelVolume.cpp:33-60, 294-327
 *   vrt_set_volume_format,          VDXVoxelVolume::EncodeVoxel  RDXVoxelVolume.cpp:399-421 (the 4-byte volume texel,
 *   vrt_volume_upload_texels        DecodeDensity Shaders/Include/Voxel.hlsli:254-266)
 *   vrt_volume_set_material         VDXVoxelVolume::UpdateGeometryConstantBuffer  :368-397
 *   vrt_volume_free                 VRDXScene::RemoveVoxelVolume  Renderer/DX/Private/RDXScene.cpp:663-701
 *   vrt_env_upload                  VRDXScene::InitEnvironmentMap RDXScene.cpp:181-199
 *   vrt_scene_set                   VRDXScene::SyncWithScene + PrepareForRendering
 *                                   RDXScene.cpp:109-118, 150-174, 454-545, 703-755
 *                                   VDXLevelObject::Update  Renderer/DX/Private/RDXLevelObject.cpp:29-48
 *   vrt_render / vrt_render_rows    VDXRenderer::Render → DoRendering → DispatchRays(W,H,1)
 *                                   DXRenderer.cpp:37-66, 827-867 (+ the HLSL entry points
 *                                   Renderer/DX/Resources/Shaders/Raytracing.hlsl:26-455)
 *   vrt_last_timing                 (no reference analogue; FPS counter Engine.cpp:250-262)
 *
 * Error convention: 0 = OK, negative = error.  The reference logs and returns early
 * (DXRenderer.cpp:220-225); the adaptor maps negative codes onto that behaviour.
 * Threading: not thread-safe per context (matches the reference's single engine-loop
 * thread, Engine.cpp:201-227).  The caller owns all host buffers, the context owns all
 * device memory.
 */
#ifndef VRT_H
#define VRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRT_MAX_VOLUMES      20 /* MaxAllowedObjectData, Shaders/RaytracingHlsl.h:112 */
#define VRT_MAX_POINT_LIGHTS 5  /* RaytracingHlsl.h:114 */
#define VRT_MAX_SPOT_LIGHTS  5  /* RaytracingHlsl.h:113 */
#define VRT_MAX_INSTANCES    64
#define VRT_MAX_DEVICES      8
#define VRT_MAX_RESOLUTION   9    /* N = 513: the kernels address a volume with 32-bit byte offsets (brick pool 1.07 GB);
                                     the reference's own tools stop at 8 (VolumeConverter.cpp:44-49) */
#define VRT_FRAMES_IN_FLIGHT 3    /* frame slots of vrt_render_begin / vrt_render_end: FrameCount, DXConstants.cpp:23 */
#define VRT_MAX_TEXTURES     64   /* 2D material textures resident at once (3 per volume slot + spare) */
#define VRT_FLAG_DIAG_TIMELINE 4 /* run the diagnostic kernel build that stamps per-wave timeline records */
#define VRT_FLAG_OUTPUT_RGBA8 8  /* store R8G8B8A8_UNORM pixels (4 B, R in the low byte, A = 255) instead of float4:
                                   the reference's back-buffer precision (B8G8R8A8_UNORM, DXConstants.cpp:21);
                                   value = (uint)(min(c,1)*255 + 0.5) of the float channel the float4 path stores */

#define VRT_FLAG_BLOCK_PER_FRAME 64 /* vrt_render_block: one march launch per frame, back to back on the stream, instead of ONE launch
                                      for the block's frames (A/B measurements and tests; same pixels) */
#define VRT_FLAG_NO_CULL_RECT 128 /* the host computes no cull rectangle: every wave looks at the scene and slab-tests its rays
                                    (measurements of what the rectangle saves; same pixels) */
#define VRT_FLAG_FULL_ONE_KERNEL 256 /* full closest hit (point / spot lights, mirror bounces, material textures): one kernel even for a block
                                       of frames — vrt_render_block's launches otherwise run it in passes (camera-ray march / the hits'
                                       light shadow rays and their shading / a third pass for the lanes that mirror, in frames that can
                                       bounce; same pixels and counters, twice the waves per SIMD).  A lone frame is one kernel by default */
#define VRT_FLAG_FULL_THREE_PASS 512 /* ... and the three passes even for a lone frame (tests, measurements).  Not both */
#define VRT_FLAG_OUTPUT_BGRA8 2048 /* together with VRT_FLAG_OUTPUT_RGBA8: the bytes in the reference's back-buffer order, B8G8R8A8_UNORM
                                     (DXGI_FORMAT_B8G8R8A8_UNORM, DXConstants.cpp:21, DXRenderer.cpp:1322): B in the low byte; same values */
#define VRT_FLAG_NO_HIT_POLISH 1024 /* closest hits stay where the cone threshold stopped the ray (rounds 1-3) instead of moving on to the surface's
                                      zero crossing by VRT_HIT_POLISH_SAMPLES secant samples: A/B measurements, tests */
/* Two of the reference's artefacts, selectable so that a frame can be "what the DXR backend renders" (the C++ adaptor sets both by default,
 * like its default normal texel; measured against the literal restatement of the reference's shaders): */
#define VRT_FLAG_REFERENCE_VIEW_VECTOR 4096 /* the reference never normalises its camera direction (GenerateCameraRay, Shaders/Include/Ray.hlsli:36-48):
                                      its closest-hit shader evaluates the BRDF with wo = -WorldRayDirection(), a vector of length
                                      L = |(x aspect tan(fov/2), y tan(fov/2), -1)| = 1 (frame centre) ... 1.55 (corner of a 16:9 frame at 60 degrees),
                                      and backs the camera ray's secondary rays off by 0.1 L (Raytracing.hlsl:52,85-95).  With this flag the camera
                                      ray's hit is shaded with wo = -L d and its shadow / mirror rays start 0.1 L (Cube modes 0.2 L) back; rays
                                      behind a mirror bounce are unit vectors in the reference too.  Smooth materials' highlights move by up
                                      to 13 of 255 (9.5 % of a mirror scene's surface pixels by more than one step); rough ones do not move */
#define VRT_FLAG_REFERENCE_BOUNDARY_TEXELS 8192 /* the normal's central difference (GetNormal, Voxel.hlsli:783-804) reads the cells one step either side of
                                      the hit's; where such a cell lies outside the grid the reference's Load returns texel 0 for the samples beyond
                                      the volume texture (GetDensity, :607-617), i.e. the neighbour's interpolant is (1 - f) x the boundary plane's.
                                      Default (SURVEY App. A rule 7): the neighbour cell is clamped to the grid (a one-sided difference).  Only
                                      surfaces within one cell of the volume's box differ */
#define VRT_HIT_POLISH_SAMPLES 2   /* part of the march contract: samples a closest hit spends on its way from the stop point to the crossing */
#define VRT_FLAG_NO_TIMING 16    /* the launch records no event pair: vrt_last_timing / vrt_timing_history report 0 ms for it.
                                   An event pair costs 5-7 us of queue time per launch (profiles/r02_launch_overhead.txt);
                                   callers that keep many small launches in flight time a sample of them */

enum vrt_status {
    VRT_OK = 0,
    VRT_ERR_INVALID = -1,     /* bad argument */
    VRT_ERR_NO_DEVICE = -2,   /* no usable HIP device */
    VRT_ERR_HIP = -3,         /* a HIP runtime call failed */
    VRT_ERR_OOM = -4,
    VRT_ERR_SLOT = -5,        /* volume slot out of range or empty */
    VRT_ERR_NOT_READY = -6,   /* render without scene / volume */
    VRT_ERR_UNSUPPORTED = -7  /* feature not implemented (reserved; every EVRenderMode is implemented) */
};

/* EVRenderMode, Renderer/Public/Renderer.h:32-42 (same numeric values). */
enum vrt_render_mode {
    VRT_MODE_INTERP = 0,
    VRT_MODE_INTERP_UNLIT = 1,
    VRT_MODE_INTERP_NOTEX = 2,
    VRT_MODE_INTERP_NOTEX_UNLIT = 3,
    VRT_MODE_CUBE = 4,
    VRT_MODE_CUBE_UNLIT = 5,
    VRT_MODE_CUBE_NOTEX = 6,
    VRT_MODE_CUBE_NOTEX_UNLIT = 7
};

/* Which device data path the march uses.  All paths produce bit-identical pixels. */
enum vrt_data_path {
    VRT_PATH_AUTO = 0,
    VRT_PATH_DENSE = 1,       /* taps from the dense N^3 grid in global memory */
    VRT_PATH_BRICK = 2,       /* taps from 4^3-cell (5^3-sample) bricks in global memory */
    VRT_PATH_BRICK_LDS = 3,   /* bricks staged through a per-wave LDS brick cache */
    VRT_PATH_CELLS = 4        /* VRT_FORMAT_TEXEL16 volumes only (others: VRT_PATH_BRICK): taps from 16-byte cell records — every cell
                                 keeps its own 8 corner texels, one aligned 16-byte load per sample instead of four 4-byte ones,
                                 for 4x the bytes of the int16 bricks */
};

/* How a volume's densities are kept on the device (per upload, vrt_set_volume_format). */
enum vrt_volume_format {
    VRT_FORMAT_F32 = 0,      /* fp32 samples: 512-B bricks of 5^3 floats (round 1's layout) */
    VRT_FORMAT_TEXEL16 = 1   /* the reference's own volume texel: sign + 15-bit trunc(|d| * 100)
                                (VDXVoxelVolume::EncodeVoxel, Renderer/DX/Private/RDXVoxelVolume.cpp:399-421; DecodeDensity,
                                Shaders/Include/Voxel.hlsli:254-266), held as int16 in 256-B bricks.  The march sees exactly the
                                field the DXR backend sees, 0.01 * (+-q): the kernels interpolate the integers and fold the 0.01
                                into the volume's density scale */
};

/* Host/disk layout of one voxel: VVoxel, Voxel/Public/Voxel.h:23-30 (8 bytes). */
typedef struct vrt_voxel {
    uint8_t material;
    uint8_t pad_[3];
    float density;
} vrt_voxel;

/* VMaterial scalars that reach the GPU: VGeometryConstantBuffer, RaytracingHlsl.h:86-100.
 * Defaults Core/Public/Material.h:25-27: tint (0.8,0.8,0.8,1), roughness 0.8, metallic 0. */
typedef struct vrt_material {
    float tint[4];
    float roughness;
    float metallic;
} vrt_material;

/* One placed VVoxelObject: VDXLevelObject::Update, RDXLevelObject.cpp:29-48.
 * rotation is a quaternion in x,y,z,w order (Eigen storage order, Core/Private/Quat.cpp:98-116).
 * object→world is  p_w = scale ∘ (rotation · p_o) + position  (DirectXMath `rotation*scale*translation`). */
typedef struct vrt_instance {
    int32_t volume_slot;
    float position[3];
    float rotation[4];
    float scale[3];
} vrt_instance;

/* VPointLightBuffer / VSpotLightBuffer, RaytracingHlsl.h:64-84; DXLightFactory.cpp:20-50. */
typedef struct vrt_point_light {
    float position[3];
    float color[3];
    float intensity;
    float att_linear;
    float att_exp;
} vrt_point_light;

typedef struct vrt_spot_light {
    float position[3];
    float forward[3];
    float color[3];
    float intensity;
    float att_linear;
    float att_exp;
    float cos_angle;          /* cos(Angle/2) */
    float cos_falloff_angle;  /* cos(FalloffAngle/2) */
} vrt_spot_light;

/* VSceneConstantBuffer + TLAS instance list: RaytracingHlsl.h:53-62, RDXScene.cpp:454-545,703-724. */
typedef struct vrt_scene {
    float cam_position[3];
    float cam_rotation[4];   /* quaternion x,y,z,w; forward = q·(+X), up = q·(+Z)  (Core/Private/Vector.cpp:42-46) */
    float cam_fov_deg;       /* vertical FOV, Scene/Public/Camera.h:29 (default 60) */
    float cam_near;          /* :30 (0.01) — carried for completeness, unused by ray generation */
    float cam_far;           /* :31 (125)  — idem */
    float light_dir[3];      /* directional light: unit vector *towards* the light (RDXScene.cpp:720-723) */
    float light_strength;
    int32_t n_instances;
    int32_t n_point_lights;
    int32_t n_spot_lights;
    int32_t pad_;
    vrt_instance instances[VRT_MAX_INSTANCES];
    vrt_point_light point_lights[VRT_MAX_POINT_LIGHTS];
    vrt_spot_light spot_lights[VRT_MAX_SPOT_LIGHTS];
} vrt_scene;

/* Per-frame render parameters (DispatchRays dimensions + the march contract of DESIGN.md §3). */
typedef struct vrt_params {
    int32_t width;
    int32_t height;
    int32_t max_steps;    /* march budget per (ray, instance), 0 .. 65535 positions; reference budget: 255 (Raytracing.hlsl:229) */
    int32_t shadow;       /* 1 = cast the directional-light shadow ray (Raytracing.hlsl:52-59) */
    int32_t mode;         /* vrt_render_mode */
    int32_t path;         /* vrt_data_path */
    int32_t max_bounces;  /* mirror-reflection depth, 0..2 (MAX_RAY_RECURSION_DEPTH 3 = primary + 2) */
    int32_t flags;        /* bits 0-1: blockIdx→tile map, 0 supertile (default) / 1 XCD band / 2 linear
                             (speed only, never results); bit 2: VRT_FLAG_DIAG_TIMELINE; bit 3:
                             VRT_FLAG_OUTPUT_RGBA8; bit 4: VRT_FLAG_NO_TIMING; bit 5: accepted and ignored (it was round 1's
                             VRT_FLAG_SKIP_EMPTY: the march never samples empty cells now); bit 6: VRT_FLAG_BLOCK_PER_FRAME; bit 7: VRT_FLAG_NO_CULL_RECT;
                             bit 8: VRT_FLAG_FULL_ONE_KERNEL; bit 9: VRT_FLAG_FULL_THREE_PASS; bit 10: VRT_FLAG_NO_HIT_POLISH; bit 11: VRT_FLAG_OUTPUT_BGRA8;
                             bit 12: VRT_FLAG_REFERENCE_VIEW_VECTOR; bit 13: VRT_FLAG_REFERENCE_BOUNDARY_TEXELS.  Others 0 */
    float eps_hit;        /* hit when the scaled distance falls below this (ray-parameter units) */
    float eps_in;         /* entry offset after the AABB slab test (reference: 0.01, Raytracing.hlsl:178) */
    float step_min;       /* lower bound of one march step (ray-parameter units) */
    float k_relax;        /* sphere-trace relaxation factor.  <= 1: every distance-driven step is k_relax times the sampled distance.
                             > 1 (typically 1.7): over-relaxation — the step is stretched by k_relax, and when the empty spheres
                             around two successive samples then fail to overlap the ray returns to the first one's plain step
                             (Keinert et al., "Enhanced Sphere Tracing", 2014): same surfaces, fewer positions along grazing rays */
    float cone_eps;       /* pixel-footprint termination: the hit threshold at ray parameter t is
                             eps_hit + cone_eps * t (0 = constant threshold).  Typically the angular
                             radius of a pixel, tan(fov/2)/height */
} vrt_params;

typedef struct vrt_timing {
    float kernel_ms;         /* march kernel, hipEvent pair on the launch stream */
    float gather_ms;         /* multi-device tile gather (0 for one device) */
    float total_ms;          /* launch → image available */
    uint32_t width, height;
    uint64_t primary_rays;
    uint64_t shadow_rays;    /* shadow rays actually cast (all lights) */
    uint64_t bounce_rays;    /* mirror-reflection rays actually cast */
    uint64_t primary_steps;  /* trilinear samples taken by primary + bounce rays */
    uint64_t shadow_steps;   /* trilinear samples taken by shadow rays */
    uint64_t hits;           /* radiance hits (each costs 6 extra trilinear samples for the normal) */
    uint64_t exhausted_rays; /* marches (ray x instance) that visited max_steps positions while still inside the volume:
                                treated as misses (the reference paints them red, Raytracing.hlsl:325-334); non-zero
                                means max_steps is too small for the scene */
} vrt_timing;

typedef struct vrt_ctx vrt_ctx;

/* device_count >= 1; devices[i] are HIP ordinals.  One context may drive 1..8 devices: vrt_render deals the frame's 8-row
 * strips round-robin to them (device g renders strips g, g+n, ...; contiguous tiles would put every object row on the middle
 * devices), volumes are replicated, and every device copies its strips into device 0's frame over the peer links as soon as
 * its own march is done — one strided copy per device, joined to device 0's stream by events.  One process per GPU
 * uses vrt_render_strips / vrt_render_block + vrt_gather_tiles / vrt_exchange_tiles instead. */
int vrt_create(vrt_ctx** out, int device_count, const int* devices);
int vrt_destroy(vrt_ctx* ctx);

/* resolution <= VRT_MAX_RESOLUTION.  density: N^3 floats, N = 2^resolution + 1, index = x*N*N + z*N + y
 * (Core/Private/MathHelpers (2).cpp:43-46).  material_or_null: N^3 bytes, same indexing. */
int vrt_volume_upload(vrt_ctx* ctx, int slot, uint8_t resolution, float extent,
                      const float* density, const uint8_t* material_or_null);
/* Same, straight from VVoxelVolume's storage (std::vector<VVoxel>, 8 B records). */
int vrt_volume_upload_voxels(vrt_ctx* ctx, int slot, uint8_t resolution, float extent,
                             const vrt_voxel* voxels);
/* Device format of the volumes uploaded FROM NOW ON (vrt_volume_upload, _upload_voxels, vrt_voxelize_mesh); default
 * VRT_FORMAT_F32.  With VRT_FORMAT_TEXEL16 the upload quantises every density on the device the way
 * VDXVoxelVolume::UpdateVolumeTexture does on the host (RDXVoxelVolume.cpp:294-327).  Volumes already resident keep theirs. */
int vrt_set_volume_format(vrt_ctx* ctx, int format);
/* The reference's volume texture itself: N^3 R8G8B8A8_UINT texels, texel (x,y,z) at byte 4*(z*N*N + y*N + x)
 * (UpdateVolumeTexture, RDXVoxelVolume.cpp:294-327 with Core/Private/MathHelpers (2).cpp:26-46): R = sign<<7 | q>>8,
 * G = q & 0xff, B = A = material.  Always VRT_FORMAT_TEXEL16. */
int vrt_volume_upload_texels(vrt_ctx* ctx, int slot, uint8_t resolution, float extent, const uint8_t* rgba8_texels);
int vrt_volume_set_material(vrt_ctx* ctx, int slot, const vrt_material* material);
/* density_scale: object-space length of one density unit (1 for metric SDFs; the Voxelizer's
 * extraction threshold for its shell volumes).  step_max: largest object-space step that is
 * safe to take from any sample (<= 0: unbounded). */
int vrt_volume_set_metric(vrt_ctx* ctx, int slot, float density_scale, float step_max);
/* The Voxelizer's hot loop on the device: VVolumeConverter::ConvertMeshInfoToVoxelVolume / VoxelizeFace
 * (Voxelizer/Private/VolumeConverter.cpp:30-84, 161-252): fills slot with the unsigned shell field of a triangle
 * mesh — density = dist/thr - 0.5 (thr = cell*sqrt 3) in every triangle's (bbox +- thr +- 1 voxel) index box,
 * minimum over triangles, background 2*extent, material = (density <= 0) — and sets the slot's metric to
 * (thr, thr/2).  positions: 3 floats per vertex in volume space; resolution / extent as the converter derives
 * them from the mesh name and bounds (:32-49).  Degenerate triangles and out-of-range indices are skipped and
 * counted.  Bit-identical to the CPU converter of this build (same source, csrc/voxelize_core.h). */
int vrt_voxelize_mesh(vrt_ctx* ctx, int slot, uint8_t resolution, float extent, const float* positions, size_t n_vertices,
                      const uint32_t* indices, size_t n_indices, size_t* skipped_or_null);

/* Reads a slot back as N^3 VVoxel records (index x*N*N + z*N + y), e.g. to write the .vox file. */
int vrt_volume_download(vrt_ctx* ctx, int slot, vrt_voxel* out);

int vrt_volume_free(vrt_ctx* ctx, int slot);

/* 2D material textures — VRenderer::InitializeTexture / UploadToGPU(VTexture) (Renderer/Public/Renderer.h:54-57)
 * and the scene's geometry-texture table (VRDXScene, RDXScene.cpp:771-800, 905-925).  R8G8B8A8_UNORM, mip 0,
 * row-major (DXTexture2D.cpp:78-81); sampled with the reference's geometry sampler: point filter, wrap
 * addressing (RDXScene.cpp:262-270).  id in [0, VRT_MAX_TEXTURES); uploading to a used id replaces it.  Image
 * decoding (the reference's WIC/DDS loaders, TextureFactory.cpp:58-125) stays with the caller. */
int vrt_texture_upload(vrt_ctx* ctx, int id, int width, int height, const uint8_t* rgba8);
int vrt_texture_free(vrt_ctx* ctx, int id);

/* VMaterial::{AlbedoTexturePath, NormalTexturePath, RMTexturePath, TextureScale} (Core/Public/Material.h:29-33;
 * indices into the texture table, RDXVoxelVolume.cpp:388-391).  id -1 = unbound: an exact identity (white
 * albedo, factors (1,1), untouched normal).  Only the textured render modes (Interp, Interp_Unlit, Cube,
 * Cube_Unlit) read them.  scale must be non-zero (default 100, 100). */
int vrt_volume_set_textures(vrt_ctx* ctx, int slot, int albedo_id, int normal_id, int rm_id, float scale_u, float scale_v);

/* 6 faces (+X,-X,+Y,-Y,+Z,-Z), each face_size^2 RGBA8, row-major, D3D cube-face orientation.
 * NULL / 0 removes the environment (misses read black, like an unbound SRV). */
int vrt_env_upload(vrt_ctx* ctx, int face_size, const uint8_t* rgba8_faces);

int vrt_scene_set(vrt_ctx* ctx, const vrt_scene* scene);

/* Render the whole frame.  host_rgba_or_null: width*height float4 (RGBA, alpha 1), row-major
 * (width*height uint32 R8G8B8A8 when params->flags has VRT_FLAG_OUTPUT_RGBA8). */
int vrt_render(vrt_ctx* ctx, const vrt_params* params, float* host_rgba_or_null);
/* Render rows [row0, row0+rows) of the frame on the context's first device into a caller-owned
 * *device* buffer of rows*width float4 (uint32 R8G8B8A8 with VRT_FLAG_OUTPUT_RGBA8), asynchronously on `hip_stream` (a hipStream_t, may be
 * NULL).  No host synchronisation.  The only allocation on this path is the launch's per-wave counter buffer, which
 * belongs to the STREAM: the first launch on a stream, and any launch with more 16x16-pixel tiles than every earlier
 * one on that stream, allocates; all others do not.  So one un-captured launch of the same (or a larger) size on
 * the stream you are going to capture on makes the launch safe to capture into a hipGraph.  Counter buffers are
 * never freed before vrt_destroy, so a captured launch stays replayable after later, larger launches; it must be
 * re-captured after vrt_volume_upload* / vrt_volume_free / vrt_texture_* / a vrt_env_upload of another size /
 * vrt_scene_set (they replace device buffers or arrays the launch dereferences).  A vrt_scene_set made WHILE frames are in flight
 * (vrt_render_begin) defers its small device copies to the next launch that needs them: that launch must not be a captured one
 * (VRT_ERR_NOT_READY) — the un-captured launch ahead of a capture covers it.  Up to 16 streams may have
 * launches in flight at once (the reference keeps 3 frames in flight, DXConstants.cpp:23): each stream has its
 * own counter buffer, each launch its own event pair. */
int vrt_render_rows(vrt_ctx* ctx, const vrt_params* params, int row0, int rows,
                    void* device_rgba, void* hip_stream);

/* Interleaved-strip variant of vrt_render_rows for multi-GPU load balance (static
 * contiguous row tiles put all the object rows on the middle GPUs).  The frame is cut into strips
 * of strip_rows rows; this launch renders strips first_strip, first_strip + strip_stride, ...
 * (n_strips of them; rank g of n passes first_strip = g, strip_stride = n) into a COMPACT device
 * buffer of n_strips*strip_rows rows x width pixels, strip after strip.  Rows at or beyond
 * params->height are skipped (their pixels are left untouched).  Same pixels as vrt_render_rows for
 * the same frame rows; asynchronous on hip_stream like it.  (DispatchRays(W,H,1),
 * DXRenderer.cpp:827-867 — the reference is single-adapter, NodeMask 0.) */
int vrt_render_strips(vrt_ctx* ctx, const vrt_params* params, int strip_rows, int first_strip,
                      int strip_stride, int n_strips, void* device_rgba, void* hip_stream);

/* One camera of a block of frames (the fields of vrt_scene's camera: Scene/Public/Camera.h:27-31). */
typedef struct vrt_camera {
    float position[3];
    float rotation[4]; /* quaternion x,y,z,w */
    float fov_deg;
} vrt_camera;

/* What vrt_render_block renders: n_frames frames of the current scene, frame f from cameras[f] (NULL: the scene's own
 * camera for every frame) into device_rgba + f*frame_stride_bytes.  Rows of every frame: strip_rows > 0 -> the strips of
 * vrt_render_strips (first_strip, strip_stride, n_strips); strip_rows == 0 -> rows [row0, row0+rows) like vrt_render_rows.
 * scenes != NULL: a block of frames of a scene that CHANGES from frame to frame — frame f is rendered exactly as
 * vrt_scene_set(&scenes[f]) followed by a one-frame launch would render it (camera, directional / point / spot lights, placed
 * objects: the reference moves objects every frame and rebuilds its TLAS every frame, RendererEngineInstance.cpp:111-130,
 * DXRenderer.cpp:809-825, RDXScene.cpp:454-545) — still with ONE march launch for the block: per frame the host packs the
 * instances, builds the instance BVH and the cull rectangle, and the records travel to the device ahead of the launch on its
 * stream (<= 12 KB per frame).  cameras must then be NULL (every scene carries its camera); the scene set by vrt_scene_set is
 * neither used nor changed.  Volumes, materials, textures and the sky box are the resident ones, the same for every frame.
 * A block with scenes != NULL cannot be captured into a hipGraph (VRT_ERR_INVALID on a capturing stream): its per-frame records are
 * staged in pinned memory and copied ahead of the launch, and a replay would copy whatever they hold by then. */
typedef struct vrt_block {
    int32_t n_frames;                 /* 1 .. 256 */
    int32_t strip_rows, first_strip, strip_stride, n_strips;
    int32_t row0, rows;
    const vrt_camera* cameras;        /* n_frames cameras, or NULL */
    uint64_t frame_stride_bytes;      /* >= the bytes of one frame's rows, a multiple of the pixel size (16, or 4 with VRT_FLAG_OUTPUT_RGBA8) */
    const vrt_scene* scenes;          /* n_frames scenes (per-frame scene state), or NULL: the scene of vrt_scene_set for every frame */
} vrt_block;

/* n_frames frames with ONE call and ONE march launch (the kernel's grid has a frame axis; up to 48 frames the cameras travel in
 * the kernarg segment, for more their 64-byte records are copied to the device ahead of the launch on the same stream; a stream that
 * is being captured into a graph gets launches of 48): the same pixels as n_frames calls of vrt_render_rows / vrt_render_strips with the scene's camera set to
 * cameras[f] in between, no host synchronisation, no allocation after the stream's first launch of that size (per launch stream the
 * context keeps the per-wave counters, for blocks of more than 48 frames the camera records, and — scenes that need the full closest
 * hit: point / spot lights, mirroring materials, textures — 20 bytes per pixel of the block's tiles of hit records between the passes
 * it then runs in, at most 4 GB per launch: larger blocks are cut into several launches).  The reference
 * keeps FrameCount = 3 frames in flight on its swap chain (DXConstants.cpp:23, DXRenderer.cpp:974-989) because a frame's last
 * third is a few latency-bound waves on an otherwise idle GPU; inside one launch the dispatcher back-fills those wave slots
 * with the next frame's waves, so the tail is paid once per launch instead of once per frame, whatever the number of streams
 * and hardware queues — and a GPU's eighth of a frame that is split 8 ways (11 us of march) no longer pays a launch and an
 * event pair of its own (profiles/r02_launch_overhead.txt).  Also for rendering a camera path.  vrt_timing_history /
 * vrt_launch_history report one duration per LAUNCH (vrt_launch_history also says how many frames it covered);
 * vrt_last_timing holds the counters of the block's last frame and the duration of its last launch. */
int vrt_render_block(vrt_ctx* ctx, const vrt_params* params, const vrt_block* block, void* device_rgba, void* hip_stream);

/* vrt_render_block with the frames handed to the HOST: the block is marched into a context-owned device buffer (one launch, as
 * above) and copied into context-owned pinned host memory; *host_frames points at frame 0, frame f at + f * (bytes of one frame's
 * rows), valid until the next call of this function or vrt_destroy.  For the VRenderer-shaped side (csrc/host/HipRenderer.cpp:
 * RenderBlock): a camera path, or a stretch of a scene's animation (block->scenes), reaches the block launch without the caller
 * owning device memory.  block->frame_stride_bytes is ignored (frames are packed).  Blocks of more than 4 GB are refused
 * (VRT_ERR_INVALID: render the path in parts).  Synchronous; single-device contexts. */
int vrt_render_block_host(vrt_ctx* ctx, const vrt_params* params, const vrt_block* block, const void** host_frames);

/* ---- multi-GPU exchange (one process per GPU) --------------------------------------------------------------------------
 * The reference is single-adapter (every D3D12 object is created with NodeMask 0, DXRenderer.cpp:253); the frame of this
 * build shards by rows / interleaved strips (vrt_render_rows / vrt_render_strips above), volumes replicated, and ONE
 * collective per frame brings the tiles to rank `root`: RCCL's ncclGather over xGMI (rccl.h:745), resolved from librccl at
 * run time (the library loads without it; these entry points then return VRT_ERR_UNSUPPORTED).
 *   vrt_comm_unique_id   rank 0 makes the 128-byte communicator id (ncclGetUniqueId) and hands it to the other ranks by
 *                        any means of the application (torch.distributed broadcast, MPI, a file)
 *   vrt_comm_init        every rank: ncclCommInitRank on the context's first device; collective, blocks until all joined
 *   vrt_gather_tiles     asynchronously on hip_stream: every rank contributes tile_bytes from device_tile; on `root`,
 *                        device_frame receives world x tile_bytes, rank-major (other ranks pass NULL).  In-order with the
 *                        march launches of the same stream: no host synchronisation
 *   vrt_comm_expect_sizes  the ranks agree on the byte counts of the two calls below once, so that a wrong block size on one rank is an error
 *                        code instead of a hang inside RCCL
 *   vrt_exchange_tiles   the all-to-all form of the same exchange, for frames that are assembled on DIFFERENT ranks (frame g of a
 *                        block on rank g / m): device_tiles holds world chunks of chunk_bytes, chunk d goes to rank d;
 *                        device_recv receives world chunks, chunk s from rank s.  One group of ncclSend / ncclRecv
 *                        (rccl.h:690-725).  A gather onto ONE rank moves (world-1)/world of every frame over that rank's inbound
 *                        xGMI links (7 x ~77 GB/s): at 8.3 MB per RGBA8 1080p frame that caps an 8-GPU job near 60 000
 *                        frames/s whatever the march does; spread over all ranks the same bytes use every link of the node
 * librccl's version is checked when it is first resolved: anything but major version 2 disables these entry points. */
#define VRT_COMM_ID_BYTES 128
int vrt_comm_unique_id(void* id_out);
int vrt_comm_init(vrt_ctx* ctx, int world, int rank, const void* id);
int vrt_comm_destroy(vrt_ctx* ctx);
/* Collective, synchronous: the ranks agree on the sizes they are going to pass to vrt_gather_tiles (tile_bytes) and vrt_exchange_tiles
 * (chunk_bytes) — a collective whose ranks disagree about its size does not fail inside RCCL, it waits for ever.  Every rank's pair travels
 * to every rank in one group of fixed-size messages; a disagreement returns VRT_ERR_INVALID on every rank.  Afterwards a call with any
 * other size is refused with VRT_ERR_INVALID before it reaches RCCL (0: that collective is not going to be used).  Optional (without it the
 * sizes are unchecked, as before); call again to change the sizes. */
int vrt_comm_expect_sizes(vrt_ctx* ctx, size_t gather_tile_bytes, size_t exchange_chunk_bytes);
int vrt_gather_tiles(vrt_ctx* ctx, const void* device_tile, void* device_frame_or_null, size_t tile_bytes, int root, void* hip_stream);
int vrt_exchange_tiles(vrt_ctx* ctx, const void* device_tiles, void* device_recv, size_t chunk_bytes, void* hip_stream);

/* Pipelined rendering — the reference keeps FrameCount = 3 frames in flight and paces them with fences
 * (DXConstants.cpp:23, DXRenderer.cpp:974-989); a frame's last third is a few latency-bound waves, which the next frame's
 * march hides.  vrt_render_begin snapshots the scene as it is NOW (instances, BVH, lights, material / metric / texture
 * tables), enqueues the march of the whole frame and its copy into an internal pinned host frame on the slot's own
 * stream, and returns at once; the application may go on to vrt_scene_set the next frame.  vrt_render_end waits for that
 * slot and hands out the pinned frame (width*height float4, or uint32 with VRT_FLAG_OUTPUT_RGBA8), valid until the slot
 * is begun again.  slot in [0, VRT_FRAMES_IN_FLIGHT); single-device contexts only; volume / texture / sky uploads still
 * drain every frame in flight first. */
int vrt_render_begin(vrt_ctx* ctx, const vrt_params* params, int slot);
int vrt_render_end(vrt_ctx* ctx, int slot, const void** host_pixels);

int vrt_last_timing(vrt_ctx* ctx, vrt_timing* out);
/* Kernel durations (ms) of the last n vrt_render_rows/vrt_render launches, oldest first;
 * returns how many were written (<= n), or a negative status. */
int vrt_timing_history(vrt_ctx* ctx, int n, float* kernel_ms_out);

/* The same, with the number of frames each launch covered (vrt_render_block: up to 256 per launch; everything else: 1). */
int vrt_launch_history(vrt_ctx* ctx, int n, float* kernel_ms_out, int* frames_out);

/* Diagnostics: per-wave records of the last frame of the last launch on the first device (which = 2, 3: of ALL frames of that
 * launch, frame after frame, for which = 0, 1 respectively), 8 words each, record
 * index = blockIdx*4 + wave.  which = 0: counters {primary_rays, shadow_rays, bounce_rays,
 * primary_steps, shadow_steps, hits, exhausted_rays, 0}.  which = 1 (only after a VRT_FLAG_DIAG_TIMELINE launch):
 * {start, end (100 MHz ticks), iterations whose taps were back within 450 cycles, XCC_ID | HW_ID<<4, longest
 * per-lane sample chain, tap-fetch cycles, march-loop cycles, march-loop iterations} of the lane with the longest
 * chain.  Returns the number of words available and copies min(max_words, that). */
long long vrt_debug_wave_records(vrt_ctx* ctx, int which, uint32_t* out, long long max_words);

/* Diagnostics: which closest-hit kernel form the last march launch on the first device ran (a bit set of VRT_FORM_*; negative: error).
 * The lean kernel (camera ray + directional shadow ray, 8 waves per SIMD) is what a frame gets unless it needs more; tests and bench.py
 * use this to assert that, e.g., the reference's default material state (a 1x1 normal texel on every material) stays on it. */
#define VRT_FORM_FULL 1        /* full closest hit: point / spot lights, mirror bounces or material IMAGES */
#define VRT_FORM_PASSES 2      /* ... as three passes (a block of frames) rather than one kernel */
#define VRT_FORM_TEXTURED 4    /* a textured render mode with a bound texture in sight */
#define VRT_FORM_MAY_BOUNCE 8  /* bounces allowed and some material can mirror */
#define VRT_FORM_LEAN_REF 16   /* the lean kernel's instantiation that folds constant (1x1) textures in and honours
                                  VRT_FLAG_REFERENCE_VIEW_VECTOR / _BOUNDARY_TEXELS */
int vrt_debug_last_kernel_form(vrt_ctx* ctx);

/* Diagnostics: the rate (G trilinear samples per second) the chip sustains for the march's inner operation in isolation — the 8 taps of a
 * sample from a pool of n_bricks (a power of two) brick records of `format` plus the lerp tree, at full occupancy, every lane on an
 * independent pseudo-random sequence of cells (coherent_lanes = 0) or the 64 lanes of a wave on the 3x3 cells an 8x8-pixel tile covers
 * (1).  32 bricks stay in L1, 4096 in L2, 262144 are the pool of a 256^3 volume.  The roof bench.py's roofline.limiter_frac is measured
 * against, taken on the box and build of the run itself.  About 10 ms per call; synchronous; not for use while frames are in flight. */
int vrt_debug_gather_ceiling(vrt_ctx* ctx, int format, int coherent_lanes, unsigned n_bricks, float* gsamples_per_s_out);

const char* vrt_strerror(int status);
/* "x.y.z gfx950" */
const char* vrt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VRT_H */

/*
 * vrt_device.h — flat device-side structs shared by the host runtime (vrt_api.hip) and the
 * gfx950 kernels (vrt_kernels.hip).  These are what the reference kept in D3D12 constant
 * buffers / the TLAS (Shaders/RaytracingHlsl.h:53-100, RDXScene.cpp:454-545), re-laid for
 * scalar (SGPR) loads from the kernarg segment and small read-only global arrays.
 */
#pragma once
#include <stdint.h>

namespace vrt {

constexpr int kBrickCells = 4;                 /* cells per brick edge */
constexpr int kBrickSamples = 5;               /* samples per brick edge (cells + 1 apron) */
constexpr int kBrickFloats = 128;              /* 125 samples padded to 128 per brick record: 512 B (fp32) = four 128-B lines,
                                                  256 B (int16, VRT_FORMAT_TEXEL16) = two */
/* Internal data paths (template parameter of the march kernels).  1-3 are vrt_data_path's values. */
constexpr int kPathCube = 8;                   /* Cube render modes on fp32 bricks (bricks + cube_skip, exact grid traversal) */
constexpr int kPathBrick16 = 9;                /* VRT_PATH_BRICK on a VRT_FORMAT_TEXEL16 volume: int16 bricks */
constexpr int kPathCube16 = 10;                /* Cube render modes on int16 bricks */
constexpr int kPathCells16 = 11;               /* VRT_PATH_CELLS: int16 cell records (VRT_FORMAT_TEXEL16 volumes) */
constexpr int kNibWindow = 16;                 /* reach, in cells, of the sub-block distance transform (nibbles cap at 15) */
constexpr int kTile = 8;                       /* one wave = 8x8 pixels */
constexpr int kBlockThreads = 256;             /* 4 waves = 16x16 pixels */
constexpr int kMaxBvhNodes = 2 * 64 - 1;
constexpr int kStatWords = 7;
constexpr int kStatRecord = 8;                 /* words per wave record (32 B) */
constexpr int kDiagRecord = 8;                 /* words per wave in the diagnostic timeline buffer */
constexpr int kMaxBlocks = 1 << 20;            /* 16x16-pixel workgroups per frame of a launch (e.g. 16384 x 16384) */
#ifndef VRT_MAX_BLOCK_FRAMES
#define VRT_MAX_BLOCK_FRAMES 48
#endif
constexpr int kMaxBlockFrames = VRT_MAX_BLOCK_FRAMES;            /* frames ONE march launch covers (grid.y) with their cameras in the kernarg segment */
constexpr int kMaxLaunchFrames = 256;                            /* ... with their cameras in device memory (DFrame::cams): one small copy ahead of the launch */

/* blockIdx → tile maps of the march kernel (vrt_params.flags bits 0-1; speed only). */
constexpr int kMapSupertile = 0;
constexpr int kMapBand = 1;
constexpr int kMapLinear = 2;

/* Workgroups launched for a tiles_x x tiles_y frame under a tile map. */
inline int grid_blocks(int tiles_x, int tiles_y, int tile_map) {
    if (tile_map != kMapSupertile) return tiles_x * tiles_y;
    const int st = ((tiles_x + 3) / 4) * ((tiles_y + 3) / 4);
    return ((st + 7) / 8) * 8 * 16; /* whole supertiles, a multiple of 8 of them */
}

/* Per-volume record (VGeometryConstantBuffer analogue). */
struct DVolume {
    const float* dense;    /* N^3 fp32, index x*N*N + z*N + y (VRT_FORMAT_TEXEL16: the integer field +-q as floats) */
    const void* bricks;    /* nb^3 brick records x 128 samples (fp32, or int16 for VRT_FORMAT_TEXEL16), brick (bx,bz,by)
                              major like the dense grid, in-brick index lx*25 + lz*5 + ly */
    const void* cells;     /* VRT_FORMAT_TEXEL16: nb^3 x 64 cell records of 8 int16 (the cell's corners in tap order: (x,z) = 00, 01, 10, 11,
                              y then y+1), cell (lx,lz,ly) of a brick at record lx*16 + lz*4 + ly; else null */
    int32_t N;
    int32_t nb;
    float extent;
    float inv_cell;
    float density_scale;
    float step_max;        /* +inf when unbounded */
    float tint[3];
    float roughness;       /* clamped to [0,1] */
    float metallic;        /* clamped to [0,1] */
    float k;               /* (roughness+1)^2 / 8 from the unclamped roughness */
    float cell;            /* 2*extent / (N-1): the leap unit must use this very value (1/inv_cell can differ in the last bit) */
    int32_t format;        /* vrt_volume_format of `bricks` */
    const uint8_t* skip;   /* nb^3 bytes or null: leap count max(D-1, 0), D = Chebyshev distance (bricks) to the nearest
                              brick holding an active cell (a cell with a corner closer than step_max to the surface) */
    const uint32_t* nib;   /* nb^3 words (with skip): per brick eight 4-bit fields, one per 2^3-cell sub-block at bit
                              4*((lx>>1)*4 + (lz>>1)*2 + (ly>>1)): floor of the Euclidean distance, in cells, from the
                              sub-block's cells to the nearest active cell, capped at 15.  Read where skip[] is 0 */
    /* material textures (textured render modes): R8G8B8A8, point/wrap; px null = unbound */
    const uint8_t* tex_px[3];  /* albedo, normal, rm */
    int32_t tex_w[3], tex_h[3];
    float tex_scale[2];
    float roughness_raw, metallic_raw; /* unclamped: the textured modes clamp after the RM factor */
    const uint8_t* cube_skip; /* nb^3 bytes: Chebyshev distance (bricks) to the nearest brick holding a solid voxel
                                 (density <= 0 at a cell-origin voxel); the Cube modes' octree stand-in */
    float abox_lo[3], abox_hi[3]; /* with skip: object-space bounding box of the near bricks, (float)(cell index) * cell - extent
                                     per axis; the sphere-trace is clipped to it.  lo > hi: no near brick, every ray misses */
    /* 1x1 material textures — the reference binds a 1x1 default to every unbound slot (VRDXScene::AllocateDefaultTextures,
       RDXScene.cpp:241-260) — are constants: bit i of tex_const_mask says slot i's image is one texel, whose channels (byte / 255, the
       value a fetch would decode) are tex_const[3*i ..]; the tri-planar code then runs on the constant without a fetch (same arithmetic,
       same result), and a scene whose bound textures are all constants keeps the lean kernel */
    int32_t tex_const_mask;
    float tex_const[9];
};

struct DInstance {
    float w2o[9];          /* R^T * S^-1, row-major */
    float o2w[9];          /* S * R, row-major */
    float pos[3];
    int32_t slot;
    /* The directional light's shadow ray has the same direction for every pixel: its object-space direction
       w2o * light_dir, the inf-safe reciprocals of the slab test and 1/|direction| are computed once on the host
       (same operations, same rounding as setup_ray) instead of once per lane: 4 IEEE divisions + 1 sqrt less per hit */
    float sh_od[3];
    float sh_inv[3];
    float sh_inv_len;
    float pad_;
};

/* Flat AABB BVH over instance world boxes, nodes in preorder, threaded.  Leaf: left = -(instance+1); inner: left = the node's
 * own index + 1.  right = the node that follows this node's whole subtree in preorder (n_nodes behind the last): where the
 * walk continues after a leaf, or when the box is missed. */
struct DBvhNode {
    float lo[3];
    int32_t left;
    float hi[3];
    int32_t right;
};

struct DPointLight {
    float pos[3];
    float intensity;
    float color[3];
    float att_linear;
    float att_exp;
    float pad_[3];
};

struct DSpotLight {
    float pos[3];
    float intensity;
    float fwd[3];
    float att_linear;
    float color[3];
    float att_exp;
    float cos_angle;
    float cos_falloff;
    float pad_[2];
};

/* What differs between the frames of one launch (blockIdx.y = frame): the camera and the screen rectangle its rays can
 * reach anything in.  64 bytes: one s_load_dwordx16 at kernarg + frame * 64. */
struct DCam {
    float cam_o[3];
    float r0[3], r1[3], r2[3];
    float cx, cy;
    /* Primary rays outside this pixel rectangle (inclusive; x | y << 16) cannot reach any instance: it bounds the projection of
       every instance's active box (its whole box for volumes without an empty-space table), two pixels of margin included; the
       whole frame when a box reaches behind the camera; lo = (1,1), hi = (0,0) when nothing can be hit.  A wave whose pixels
       all lie outside goes straight to the sky */
    uint32_t cull_lo, cull_hi;
};
inline uint32_t pack_cull(int x, int y) { return (uint32_t)x | ((uint32_t)y << 16); }

/* Everything the frames of a launch share; passed by value (kernarg segment → scalar loads). */
struct HitRecord {
    float nx, ny, nz, t;
};

struct DFrame {
    float inv_w, inv_h; /* 1 / width, 1 / height */
    /* directional light */
    float light_dir[3];
    float light_strength;
    /* march contract */
    float eps_hit, eps_in, step_min, k_relax, cone_eps;
    int32_t max_steps;
    int32_t shadow;
    int32_t unlit;
    int32_t max_bounces;
    /* frame geometry */
    int32_t width, height;
    int32_t row0, rows;        /* this launch renders rows [row0,row0+rows) */
    int32_t tiles_x, tiles_y;  /* 16x16-pixel blocks covering width x rows */
    int32_t tile_map;          /* kMapSupertile / kMapBand / kMapLinear */
    int32_t diag;              /* 1: diagnostic kernel build that stamps per-wave timeline records */
    int32_t full;              /* 1: full closest hit needed (point/spot lights, or bounces allowed and a smooth material in the scene) */
    int32_t rgba8;             /* 1: store R8G8B8A8_UNORM (4 B/pixel) instead of float4; 2: B8G8R8A8_UNORM (the reference's back buffer) */
    /* interleaved strips (multi-GPU load balance): local row l of the compact tile is frame row
       ((l / strip_rows) * strip_stride + strip_first) * strip_rows + l % strip_rows; strip_rows == 0:
       contiguous rows row0 + l */
    int32_t strip_rows, strip_first, strip_stride;
    int32_t textured;          /* 1: textured render mode and some instanced volume has a texture bound */
    float back;                /* secondary rays start this far back along the ray: 0.1, Cube modes 0.2 */
    /* scene arrays */
    int32_t n_inst, n_nodes;
    int32_t n_point, n_spot;
    const DVolume* vols;
    const DInstance* inst;
    const DBvhNode* nodes;
    const DPointLight* point;
    const DSpotLight* spot;
    const uint8_t* env;        /* 6 x S x S RGBA8 or null */
    int32_t env_size;
    int32_t polish;            /* samples a closest hit spends moving from the cone threshold's stop point on to the zero crossing
                                  (VRT_HIT_POLISH_SAMPLES; 0 with VRT_FLAG_NO_HIT_POLISH) */
    float* out;                /* frame 0: rows x width float4 (or uint32 R8G8B8A8 when rgba8); frame f at out + f * frame_stride bytes */
    unsigned* stats;           /* one 8-word record per wave (4 per workgroup): primary_rays, shadow_rays,
                                  bounce_rays, primary_steps, shadow_steps, hits, exhausted_rays, 0; frame f at stats + f * stats_stride words */
    unsigned* diag_buf;        /* diagnostic build only: 8 words per wave {start, end (100 MHz), fast fetches, xcc|hw_id,
                                  longest sample chain, load+lerp cycles, loop cycles, loop iterations}, frames like stats */
    uint64_t frame_stride;     /* bytes between the frames of a launch in `out` */
    uint32_t stats_stride;     /* words between the frames of a launch in `stats` (and diag_buf) */
    int32_t n_frames;          /* = gridDim.y, 1 .. kMaxBlockFrames (kMaxLaunchFrames with cams) */
    const DVolume* vol0;       /* single-instance scenes: vols + inst[0].slot, resolved on the host so that a wave loads its instance
                                  and its volume record side by side instead of one after the other (four out of five waves of
                                  a frame only need them to find out that their rays miss) */
    /* full closest hit in passes (camera-ray march / light shadow rays + shading / mirror bounces), null: the one-kernel form.
       One record per (frame, wave, lane) in the launch's own order: record ((frame * blocks + b) * 4 + wave) * 64 + lane. */
    HitRecord* hit_rec;        /* {world normal, t} of the camera ray's closest hit (hit lanes only) */
    unsigned* hit_aux;         /* instance | shadowed-by-light bits << 16 (bit 0 directional, 1.. point, 6.. spot lights) */
    unsigned long long* hit_mask; /* per (frame, wave): the lanes whose camera ray hit */
    const DCam* cams;          /* launches of more than kMaxBlockFrames frames: the frames' camera records in device memory (null: DBlock::cam) */
    uint32_t rec_stride;       /* records between the frames of a launch (= workgroups per frame * 256: four waves of 64 lanes) */
    int32_t may_bounce;        /* 1: bounces allowed and some material can mirror (smooth, or roughness from a texture) */
    int32_t view_vec;          /* VRT_FLAG_REFERENCE_VIEW_VECTOR: the camera ray's hit is shaded with wo = -L d and its secondary rays start back * L
                                  back, L = the camera direction's length before normalisation (the reference's WorldRayDirection()) */
    int32_t zero_outside;      /* VRT_FLAG_REFERENCE_BOUNDARY_TEXELS: normal taps beyond the grid read 0 (the reference's out-of-bounds Load) */
    const void* dyn;           /* per-frame scene state (vrt_block::scenes): n_frames sections of kDynStride bytes in device memory, each a
                                  DDyn record followed by the frame's instances, BVH nodes, point and spot lights; null: every frame of the
                                  launch renders the scene in this struct.  Read by the DYN instantiations of the march kernels only */
};

/* Per-frame scene state of a launch over a scene that changes from frame to frame — what the reference re-sends every frame: the
 * scene constant buffer's light (RDXScene.cpp:703-724), the light buffers (:726-755) and the TLAS's instance list (:454-545,
 * rebuilt every frame, DXRenderer.cpp:809-825).  One section per frame at dyn + frame * kDynStride: */
struct DDyn {                  /* 64 bytes: one s_load_dwordx16 */
    float light_dir[3];
    float light_strength;
    int32_t n_inst, n_nodes, n_point, n_spot;
    int32_t vol0_slot;         /* single-instance launches: the frame's one instance's volume slot */
    int32_t pad_[7];
};
/* capacities of a section (vrt.h is not visible here; vrt_api.hip static_asserts them against VRT_MAX_INSTANCES / VRT_MAX_*_LIGHTS) */
constexpr int kDynMaxInstances = 64;
constexpr int kDynMaxNodes = 128;          /* >= kMaxBvhNodes */
constexpr int kDynMaxPointLights = 5;
constexpr int kDynMaxSpotLights = 5;
static_assert(kDynMaxNodes >= kMaxBvhNodes, "a frame's BVH fits its section");
constexpr uint32_t kDynInstOff = sizeof(DDyn);
constexpr uint32_t kDynNodesOff = kDynInstOff + kDynMaxInstances * sizeof(DInstance);
constexpr uint32_t kDynPointOff = kDynNodesOff + kDynMaxNodes * sizeof(DBvhNode);
constexpr uint32_t kDynSpotOff = kDynPointOff + kDynMaxPointLights * sizeof(DPointLight);
constexpr uint32_t kDynStride = (kDynSpotOff + kDynMaxSpotLights * sizeof(DSpotLight) + 63u) & ~63u;
static_assert(sizeof(DDyn) == 64 && kDynStride % 64 == 0, "frame sections stay 64-byte aligned");

/* The kernarg of a march launch: the shared part and one DCam per frame of the block.  The dispatcher walks blockIdx.x
 * first, so frame f + 1's waves back-fill the wave slots frame f's latency-bound tail leaves empty — what the reference gets from
 * three back buffers in flight (DXConstants.cpp:23, DXRenderer.cpp:974-989), inside ONE launch. */
struct DBlock {
    DFrame f;
    DCam cam[kMaxBlockFrames];
};
static_assert(sizeof(DCam) == 64, "DCam is one s_load_dwordx16");
static_assert(sizeof(DBlock) <= 4096, "HIP kernarg segment limit");

}  // namespace vrt

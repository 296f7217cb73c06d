/*
 * vrt_api.hip — host runtime behind the C-ABI of include/vrt.h.
 *
 * Owns every device allocation, mirrors the caller's scene into flat device structs
 * (the job VRDXScene / VDXVoxelVolume / VDXLevelObject did with D3D12 resources:
 * Renderer/DX/Private/RDXScene.cpp:109-174,454-545,703-755, RDXVoxelVolume.cpp:33-60,294-397,
 * RDXLevelObject.cpp:29-48), builds the small AABB BVH the DXR driver used to build
 * (DXRenderer.cpp:809-825), and launches the march kernel (DXRenderer.cpp:827-867).
 *
 * There is no CPU fallback: every entry point fails with a negative status when HIP does.
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <limits>
#include <new>
#include <vector>

#include "../../include/vrt.h"
#include "vrt_device.h"
#include "vrt_launch.h"
#include "voxelize_core.h"

using namespace vrt;

namespace {

/* RCCL, resolved at run time (librccl is not a link-time dependency: the library loads on hosts without it).  Only what
   the tile exchange needs; signatures from /opt/rocm/include/rccl/rccl.h:164,187,220,260,339,700,722,745,910-920.  The
   function pointers below are declared by hand, so the library's version is checked once: only major version 2 (the API these
   signatures and the value of ncclUint8 = 1, rccl.h:460, belong to) is accepted. */
constexpr int kNcclUint8 = 1;
struct RcclApi {
    typedef struct { char internal[VRT_COMM_ID_BYTES]; } UniqueId;
    int (*GetVersion)(int*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int version = 0;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*Gather)(const void*, void*, size_t, int /*ncclDataType_t*/, int, void*, hipStream_t) = nullptr;
    bool ok = false;
};
RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.ok ? &api : nullptr;
    tried = true;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) /* the copy a host application (e.g. PyTorch) already mapped wins */
        if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!h) return nullptr;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    api.Gather = reinterpret_cast<decltype(api.Gather)>(dlsym(h, "ncclGather"));
    api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(h, "ncclGetVersion"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(h, "ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(h, "ncclRecv"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GetErrorString && api.Gather && api.GetVersion && api.GroupStart &&
             api.GroupEnd && api.Send && api.Recv;
    if (api.ok) {
        /* NCCL_VERSION_CODE: major * 10000 + minor * 100 + patch from 2.9 on, major * 1000 + minor * 100 + patch before (rccl.h:20) */
        int v = 0;
        const int major = api.GetVersion(&v) == 0 ? (v >= 10000 ? v / 10000 : v / 1000) : -1;
        api.version = v;
        if (major != 2) {
            fprintf(stderr, "[vrt] librccl reports version code %d (major %d): only major version 2 is known to this build; the multi-GPU exchange is disabled\n", v, major);
            api.ok = false;
        }
    }
    return api.ok ? &api : nullptr;
}

/* the per-frame scene sections (vrt_device.h, DDyn) hold what the C-ABI's limits allow */
static_assert(kDynMaxInstances == VRT_MAX_INSTANCES && kMaxBvhNodes == 2 * VRT_MAX_INSTANCES - 1, "section capacity = VRT_MAX_INSTANCES");
static_assert(kDynMaxPointLights == VRT_MAX_POINT_LIGHTS && kDynMaxSpotLights == VRT_MAX_SPOT_LIGHTS, "section capacity = VRT_MAX_*_LIGHTS");

constexpr int kStatSlots = 16; /* streams that may have launches in flight at once without sharing a counter buffer */
constexpr int kRing = 256; /* per-launch event pairs + stat slots kept for vrt_timing_history */

struct HostVolume {
    bool used = false;
    int resolution = 0, N = 0, nb = 0;
    float extent = 0.f;
    int format = VRT_FORMAT_F32;
    int abox[6] = {0, 0, 0, -1, -1, -1}; /* bounding box of the near bricks {min x, z, y, max x, z, y} (with the empty-space table) */
    float density_scale = 1.f;
    float step_max = 0.f; /* <= 0: unbounded */
    vrt_material mat = {{0.8f, 0.8f, 0.8f, 1.0f}, 0.8f, 0.0f};
    int tex[3] = {-1, -1, -1};            /* albedo, normal, rm texture ids; -1 unbound */
    float tex_scale[2] = {100.f, 100.f};  /* VMaterial::TextureScale default, Material.h:33 */
};

struct HostTexture {
    bool used = false;
    int width = 0, height = 0;
    uint8_t first[4] = {0, 0, 0, 0}; /* texel (0,0): a 1x1 image is a constant (DVolume::tex_const) */
};

struct DeviceVolume {
    float* dense = nullptr;
    void* bricks = nullptr;       /* fp32 or int16 records, by the slot's format */
    void* cells = nullptr;        /* VRT_FORMAT_TEXEL16: 16-byte cell records (VRT_PATH_CELLS) */
    uint8_t* material = nullptr;
    uint8_t* skip = nullptr;      /* 2 x nb^3 bytes: the empty-space table (level 1) and its build scratch */
    unsigned* nib = nullptr;      /* nb^3 words: the empty-space table, level 2 (sub-block nibbles) */
    bool skip_valid = false;      /* tables built for the current metric (only when step_max > 0) */
    uint8_t* cube_skip = nullptr; /* 2 x nb^3 bytes: the Cube modes' distance-to-solid table and its build scratch */
};

/* The small read-only arrays a launch dereferences, in ONE device allocation so that a frame in flight can keep its
   own snapshot while the application already edits the scene for the next frame. */
struct SceneArrays {
    DVolume vols[VRT_MAX_VOLUMES];
    DInstance inst[VRT_MAX_INSTANCES];
    DBvhNode nodes[kMaxBvhNodes];
    DPointLight point[VRT_MAX_POINT_LIGHTS];
    DSpotLight spot[VRT_MAX_SPOT_LIGHTS];
};

/* One frame in flight (vrt_render_begin / vrt_render_end): stream, completion event, scene snapshot, device frame and
   a pinned host frame. */
struct FrameSlot {
    hipStream_t stream = nullptr;  /* the device's flight stream (shared by the slots) */
    hipEvent_t marched = nullptr;  /* the slot's march is done: its read-back may start (copy stream) */
    hipEvent_t done = nullptr;
    SceneArrays* d_scene = nullptr;
    SceneArrays* h_scene = nullptr; /* pinned */
    void* d_fb = nullptr;
    void* h_fb = nullptr; /* pinned */
    size_t fb_bytes = 0;
    bool busy = false;
    int ring = 0;       /* event-ring slot of the launch (timing) */
};

constexpr size_t kStatsBytesMax = (size_t)512 << 20; /* per-wave counters of one launch (1080p: 1.1 MB per frame, 3840x2160: 4.2 MB) */
constexpr size_t kPassBytesMax = (size_t)4 << 30; /* hit records of one launch of the full closest hit in passes (1080p: 42 MB per frame) */

struct DeviceState {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    DeviceVolume vol[VRT_MAX_VOLUMES];
    DVolume* d_vols = nullptr;
    DInstance* d_inst = nullptr;
    DBvhNode* d_nodes = nullptr;
    DPointLight* d_point = nullptr;
    DSpotLight* d_spot = nullptr;
    uint8_t* d_env = nullptr;
    uint8_t* tex[VRT_MAX_TEXTURES] = {};
    float* fb = nullptr;
    size_t fb_bytes = 0;
    /* per-wave records: one counter buffer per launch STREAM (kStatSlots streams at a time, least recently used
       slot recycled), so frames in flight on different streams — the reference keeps 3 (DXConstants.cpp:23) — never
       share one, and launches on one stream, which run in order, reuse theirs.  Buffers only ever grow, and an
       outgrown buffer is retired, not freed, until vrt_destroy: a launch captured into a hipGraph has its buffer's
       address baked into the kernarg and may be replayed at any later time. */
    unsigned* d_stats[kStatSlots] = {};
    size_t stats_cap[kStatSlots] = {}; /* blocks */
    hipStream_t stats_stream[kStatSlots] = {};
    bool stats_bound[kStatSlots] = {};
    uint64_t stats_used[kStatSlots] = {}; /* launch number of the last use (LRU) */
    std::vector<unsigned*> stats_retired;
    /* three-pass full closest hit: hit records of a launch (16 + 4 bytes per pixel of the block's tiles + 8 per wave), per launch
       stream like the counters, allocated by the first such launch of that size */
    /* launches of more than kMaxBlockFrames frames: the frames' camera records, packed into pinned host memory and copied ahead of
       the launch on its stream (per launch stream, like the counters); `cams_copied` says when the pinned copy may be packed again */
    DCam* d_cams[kStatSlots] = {};
    DCam* h_cams[kStatSlots] = {};
    hipEvent_t cams_copied[kStatSlots] = {};
    /* launches over per-frame scene state (vrt_block::scenes): the frames' sections (DDyn + instances + BVH + lights, kDynStride bytes
       each), packed into pinned host memory and copied ahead of the launch together with the camera records */
    char* d_dyn[kStatSlots] = {};
    char* h_dyn[kStatSlots] = {};
    void* d_pass[kStatSlots] = {};
    size_t pass_cap[kStatSlots] = {}; /* records */
    std::vector<void*> pass_retired;
    /* vrt_render_block_host: the block's frames on the device and in pinned host memory (grown, never shrunk) */
    void* d_blockfb = nullptr;
    void* h_blockfb = nullptr;
    size_t blockfb_bytes = 0;
    hipStream_t copy_stream = nullptr;
    hipStream_t flight_stream = nullptr; /* vrt_render_begin: the marches of the frames in flight */
    hipEvent_t part_done = nullptr;
    bool peer_to_first = false;  /* this device maps device 0's memory (hipDeviceEnablePeerAccess succeeded): the strided 2D gather copy may be used */
    int last_slot = 0;
    unsigned* d_diag = nullptr;  /* allocated on first use of VRT_FLAG_DIAG_TIMELINE (never in a capture) */
    int last_blocks = 0;         /* workgroups per frame of the last launch */
    int last_frames = 1;         /* frames of the last launch: vrt_last_timing / vrt_debug_wave_records read its LAST frame's records */
    bool last_diag = false;
    int last_form = 0;           /* vrt_debug_last_kernel_form */
    hipEvent_t joined = nullptr; /* multi-device vrt_render: this device's strips have arrived in device 0's frame */
    hipEvent_t ev0[kRing];
    hipEvent_t ev1[kRing];
    FrameSlot slot[VRT_FRAMES_IN_FLIGHT];
    bool events_ok = false;
    int ring_frames[kRing] = {}; /* frames the launch in this ring slot covered (vrt_timing_history_frames) */
    bool timed[kRing] = {}; /* launch in this ring slot recorded its event pair (false: captured into a graph, or VRT_FLAG_NO_TIMING) */
};

}  // namespace

struct vrt_ctx {
    std::vector<DeviceState> dev;
    HostVolume vol[VRT_MAX_VOLUMES];
    HostTexture tex[VRT_MAX_TEXTURES];
    int env_size = 0;
    int upload_format = VRT_FORMAT_F32; /* vrt_set_volume_format: device format of the following uploads */
    bool have_scene = false;
    bool arrays_uploaded = false; /* the device copies of inst / nodes / point / spot hold the packed scene (pack_scene defers them while frames are in flight) */
    bool scene_stale = true; /* a volume was uploaded / freed / re-measured since the scene arrays were packed (boxes, BVH) */
    vrt_scene scene;
    DInstance inst[VRT_MAX_INSTANCES];
    DBvhNode nodes[kMaxBvhNodes];
    int n_nodes = 0;
    DPointLight point[VRT_MAX_POINT_LIGHTS];
    DSpotLight spot[VRT_MAX_SPOT_LIGHTS];
    /* launch history */
    uint64_t launches = 0;
    int last_devices = 0; /* how many devices took part in the last launch */
    uint32_t last_w = 0, last_h = 0;
    float last_gather_ms = 0.f, last_total_ms = 0.f;
    float* gather = nullptr; /* full frame on device 0 (multi-device only) */
    size_t gather_bytes = 0;
    void* comm = nullptr;    /* ncclComm_t of vrt_comm_init (one process per GPU) */
    int comm_world = 0, comm_rank = 0;
    bool sizes_agreed = false;
    size_t expect_tile_bytes = 0, expect_chunk_bytes = 0; /* vrt_comm_expect_sizes: what every rank agreed to pass (0: not agreed, unchecked) */
};

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "[vrt] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, \
                    __LINE__);                                                                     \
            return e_ == hipErrorOutOfMemory ? VRT_ERR_OOM : VRT_ERR_HIP;                           \
        }                                                                                          \
    } while (0)

struct H3 {
    float x, y, z;
};
inline H3 h3(float x, float y, float z) { return H3{x, y, z}; }
inline float hdot(H3 a, H3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline H3 hcross(H3 a, H3 b) { return h3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline H3 hnormalize(H3 a) {
    float inv = 1.0f / sqrtf(hdot(a, a));
    return h3(a.x * inv, a.y * inv, a.z * inv);
}

/* Unit quaternion (x,y,z,w) → rotation matrix, v' = R v (Eigen's Quaternionf::toRotationMatrix). */
void quat_to_mat(const float q[4], float R[3][3]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float xx = x * x, yy = y * y, zz = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0][0] = 1.0f - 2.0f * (yy + zz);
    R[0][1] = 2.0f * (xy - wz);
    R[0][2] = 2.0f * (xz + wy);
    R[1][0] = 2.0f * (xy + wz);
    R[1][1] = 1.0f - 2.0f * (xx + zz);
    R[1][2] = 2.0f * (yz - wx);
    R[2][0] = 2.0f * (xz - wy);
    R[2][1] = 2.0f * (yz + wx);
    R[2][2] = 1.0f - 2.0f * (xx + yy);
}

/* View basis of XMMatrixLookToRH(eye, forward, up) and the two projection scalars of
 * XMMatrixPerspectiveFovRH that survive GenerateCameraRay (RDXScene.cpp:703-724, Ray.hlsli:36-48). */
void pack_camera(const vrt_camera& s, int width, int height, DCam& F) {
    float R[3][3];
    quat_to_mat(s.rotation, R);
    H3 fwd = h3(R[0][0], R[1][0], R[2][0]);
    H3 up = h3(R[0][2], R[1][2], R[2][2]);
    H3 r2 = hnormalize(h3(-fwd.x, -fwd.y, -fwd.z));
    H3 r0 = hnormalize(hcross(up, r2));
    H3 r1 = hcross(r2, r0);
    F.cam_o[0] = s.position[0];
    F.cam_o[1] = s.position[1];
    F.cam_o[2] = s.position[2];
    F.r0[0] = r0.x; F.r0[1] = r0.y; F.r0[2] = r0.z;
    F.r1[0] = r1.x; F.r1[1] = r1.y; F.r1[2] = r1.z;
    F.r2[0] = r2.x; F.r2[1] = r2.y; F.r2[2] = r2.z;
    float aspect = (float)width / (float)height; /* DXRenderer.cpp:47 */
    float half = tanf(s.fov_deg * (3.14159265358979323846f / 180.0f) * 0.5f);
    F.cx = aspect * half;
    F.cy = half;
}

/* The scene's own camera as a vrt_camera. */
vrt_camera scene_camera(const vrt_scene& s) {
    vrt_camera c;
    for (int a = 0; a < 3; a++) c.position[a] = s.cam_position[a];
    for (int a = 0; a < 4; a++) c.rotation[a] = s.cam_rotation[a];
    c.fov_deg = s.cam_fov_deg;
    return c;
}

/* object→world = S·R (+T), world→object = R^T·S^-1 (RDXLevelObject.cpp:38-47). */
void pack_instance(const vrt_instance& in, DInstance& out) {
    float R[3][3];
    quat_to_mat(in.rotation, R);
    float inv_s[3] = {1.0f / in.scale[0], 1.0f / in.scale[1], 1.0f / in.scale[2]};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            out.o2w[i * 3 + j] = in.scale[i] * R[i][j];
            out.w2o[i * 3 + j] = R[j][i] * inv_s[j];
        }
    out.pos[0] = in.position[0];
    out.pos[1] = in.position[1];
    out.pos[2] = in.position[2];
    out.slot = in.volume_slot;
    out.pad_ = 0.f;
}

/* Constants of the directional light's shadow ray in an instance's object space — the very expressions of the
   kernel's setup_ray / slab (mul33, dot3, 1/sqrt, inf-safe reciprocals), evaluated once here. */
void pack_shadow_ray(const float light_dir[3], DInstance& I) {
    const float* m = I.w2o;
    const float lx = light_dir[0], ly = light_dir[1], lz = light_dir[2];
    I.sh_od[0] = (m[0] * lx + m[1] * ly) + m[2] * lz;
    I.sh_od[1] = (m[3] * lx + m[4] * ly) + m[5] * lz;
    I.sh_od[2] = (m[6] * lx + m[7] * ly) + m[8] * lz;
    const float inf = std::numeric_limits<float>::infinity();
    for (int a = 0; a < 3; a++) I.sh_inv[a] = I.sh_od[a] != 0.0f ? 1.0f / I.sh_od[a] : (I.sh_od[a] > 0.0f ? inf : -inf);
    I.sh_inv_len = 1.0f / sqrtf((I.sh_od[0] * I.sh_od[0] + I.sh_od[1] * I.sh_od[1]) + I.sh_od[2] * I.sh_od[2]);
}

struct Box {
    float lo[3], hi[3];
};

Box instance_box(const DInstance& I, float extent) {
    Box b;
    for (int a = 0; a < 3; a++) {
        b.lo[a] = std::numeric_limits<float>::infinity();
        b.hi[a] = -std::numeric_limits<float>::infinity();
    }
    for (int k = 0; k < 8; k++) {
        float c[3] = {(k & 1) ? extent : -extent, (k & 2) ? extent : -extent, (k & 4) ? extent : -extent};
        for (int a = 0; a < 3; a++) {
            float w = I.o2w[a * 3 + 0] * c[0] + I.o2w[a * 3 + 1] * c[1] + I.o2w[a * 3 + 2] * c[2] + I.pos[a];
            b.lo[a] = std::min(b.lo[a], w);
            b.hi[a] = std::max(b.hi[a], w);
        }
    }
    /* pad by a relative epsilon so the (exact) object-space slab test never sees a ray the
       world-space box culled by rounding */
    for (int a = 0; a < 3; a++) {
        float pad = 1e-4f * (1.0f + std::max(fabsf(b.lo[a]), fabsf(b.hi[a])));
        b.lo[a] -= pad;
        b.hi[a] += pad;
    }
    return b;
}

/* Median-split BVH over instance boxes; ≤ 20 leaves in the reference's scenes, ≤ 64 here. */
int build_bvh(std::vector<int>& ids, int begin, int end, const std::vector<Box>& boxes, DBvhNode* nodes, int& n_nodes) {
    int me = n_nodes++;
    DBvhNode& nd = nodes[me];
    Box b;
    for (int a = 0; a < 3; a++) {
        b.lo[a] = std::numeric_limits<float>::infinity();
        b.hi[a] = -std::numeric_limits<float>::infinity();
    }
    for (int i = begin; i < end; i++)
        for (int a = 0; a < 3; a++) {
            b.lo[a] = std::min(b.lo[a], boxes[ids[i]].lo[a]);
            b.hi[a] = std::max(b.hi[a], boxes[ids[i]].hi[a]);
        }
    for (int a = 0; a < 3; a++) {
        nd.lo[a] = b.lo[a];
        nd.hi[a] = b.hi[a];
    }
    if (end - begin == 1) {
        nd.left = -(ids[begin] + 1);
        nd.right = n_nodes; /* the next node in preorder: where the walk goes on after a leaf, or when a box is missed */
        return me;
    }
    int axis = 0;
    float best = -1.f;
    for (int a = 0; a < 3; a++) {
        float lo = std::numeric_limits<float>::infinity(), hi = -lo;
        for (int i = begin; i < end; i++) {
            float c = 0.5f * (boxes[ids[i]].lo[a] + boxes[ids[i]].hi[a]);
            lo = std::min(lo, c);
            hi = std::max(hi, c);
        }
        if (hi - lo > best) {
            best = hi - lo;
            axis = a;
        }
    }
    int mid = (begin + end) / 2;
    std::nth_element(ids.begin() + begin, ids.begin() + mid, ids.begin() + end, [&](int l, int r) {
        float cl = boxes[l].lo[axis] + boxes[l].hi[axis], cr = boxes[r].lo[axis] + boxes[r].hi[axis];
        return cl < cr || (cl == cr && l < r);
    });
    int l = build_bvh(ids, begin, mid, boxes, nodes, n_nodes); /* = me + 1: the kernels rely on it */
    (void)build_bvh(ids, mid, end, boxes, nodes, n_nodes);
    nodes[me].left = l;
    nodes[me].right = n_nodes; /* threaded tree: the node that follows this whole subtree in preorder (n_nodes at the very end) */
    return me;
}

void fill_dvolume(const vrt_ctx* ctx, const DeviceState& D, const HostVolume& h, const DeviceVolume& d, DVolume& out) {
    memset(&out, 0, sizeof out);
    if (!h.used) return;
    for (int i = 0; i < 3; i++) {
        const int id = h.tex[i];
        if (id >= 0 && ctx->tex[id].used) {
            out.tex_px[i] = D.tex[id];
            out.tex_w[i] = ctx->tex[id].width;
            out.tex_h[i] = ctx->tex[id].height;
            if (ctx->tex[id].width == 1 && ctx->tex[id].height == 1) {
                out.tex_const_mask |= 1 << i;
                for (int c = 0; c < 3; c++) out.tex_const[3 * i + c] = (float)ctx->tex[id].first[c] / 255.0f; /* what tex_point_wrap decodes */
            }
        }
    }
    out.tex_scale[0] = h.tex_scale[0];
    out.tex_scale[1] = h.tex_scale[1];
    out.roughness_raw = h.mat.roughness;
    out.metallic_raw = h.mat.metallic;
    out.dense = d.dense;
    out.bricks = d.bricks;
    out.cells = d.cells;
    out.N = h.N;
    out.nb = h.nb;
    out.extent = h.extent;
    float cell = (h.extent * 2.0f) / (float)(h.N - 1); /* RDXVoxelVolume.cpp:386 */
    out.inv_cell = 1.0f / cell;
    out.cell = cell;
    /* VRT_FORMAT_TEXEL16: the field is the integer +-q; its unit 0.01 (DecodeDensity, Voxel.hlsli:254-266) goes into the scale */
    out.density_scale = h.format == VRT_FORMAT_TEXEL16 ? h.density_scale * 0.01f : h.density_scale;
    out.format = h.format;
    out.step_max = h.step_max > 0.0f ? h.step_max : std::numeric_limits<float>::infinity();
    out.tint[0] = h.mat.tint[0];
    out.tint[1] = h.mat.tint[1];
    out.tint[2] = h.mat.tint[2];
    out.roughness = std::min(std::max(h.mat.roughness, 0.0f), 1.0f);
    out.metallic = std::min(std::max(h.mat.metallic, 0.0f), 1.0f);
    float r1 = h.mat.roughness + 1.0f;
    out.k = (r1 * r1) / 8.0f; /* RDXVoxelVolume.cpp:383 */
    out.skip = (h.step_max > 0.0f && d.skip_valid) ? d.skip : nullptr;
    out.nib = out.skip ? d.nib : nullptr;
    for (int a = 0; a < 3; a++) { /* brick box {x, z, y} -> object-space box per axis x, y, z */
        const int ax = a == 0 ? 0 : (a == 1 ? 2 : 1);
        const int lo_cell = h.abox[ax] * kBrickCells;
        const int hi_cell = std::min((h.abox[3 + ax] + 1) * kBrickCells, h.N - 1);
        out.abox_lo[a] = (float)lo_cell * cell - h.extent;
        out.abox_hi[a] = (float)hi_cell * cell - h.extent;
    }
    out.cube_skip = d.cube_skip;
}

/* (Re)builds the two-level empty-space table of a slot on every device for its current metric. */
int rebuild_skip(vrt_ctx* ctx, int slot) {
    HostVolume& h = ctx->vol[slot];
    const float scale = h.format == VRT_FORMAT_TEXEL16 ? h.density_scale * 0.01f : h.density_scale;
    for (auto& D : ctx->dev) {
        DeviceVolume& v = D.vol[slot];
        v.skip_valid = false;
        if (!h.used || !(h.step_max > 0.0f) || !v.dense) continue;
        HIP_TRY(hipSetDevice(D.ordinal));
        const size_t n = (size_t)h.nb * h.nb * h.nb;
        if (!v.skip) HIP_TRY(hipMalloc(&v.skip, 2 * n));
        if (!v.nib) HIP_TRY(hipMalloc(&v.nib, n * sizeof(unsigned)));
        void* scratch = nullptr;
        HIP_TRY(hipMalloc(&scratch, nibble_scratch_bytes(h.N)));
        int* d_box = static_cast<int*>(scratch); /* the first 24 bytes of the scratch: read back before the nibble pass reuses them */
        int box[6] = {h.nb, h.nb, h.nb, -1, -1, -1};
        hipError_t e = launch_skip_table(v.dense, v.skip, v.skip + n, d_box, h.N, h.nb, scale, h.step_max, D.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(box, d_box, sizeof box, hipMemcpyDeviceToHost, D.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(D.stream);
        if (e == hipSuccess) e = launch_nibble_table(v.dense, v.nib, scratch, h.N, h.nb, scale, h.step_max, D.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(D.stream);
        (void)hipFree(scratch);
        HIP_TRY(e);
        memcpy(ctx->vol[slot].abox, box, sizeof box);
        v.skip_valid = true;
    }
    return VRT_OK;
}

int sync_volume_table(vrt_ctx* ctx) {
    DVolume table[VRT_MAX_VOLUMES];
    for (auto& D : ctx->dev) {
        for (int i = 0; i < VRT_MAX_VOLUMES; i++) fill_dvolume(ctx, D, ctx->vol[i], D.vol[i], table[i]);
        HIP_TRY(hipSetDevice(D.ordinal));
        HIP_TRY(hipMemcpy(D.d_vols, table, sizeof table, hipMemcpyHostToDevice));
    }
    return VRT_OK;
}

int free_device_volume(DeviceState& D, int slot) {
    HIP_TRY(hipSetDevice(D.ordinal));
    DeviceVolume& v = D.vol[slot];
    if (v.dense) HIP_TRY(hipFree(v.dense));
    if (v.bricks) HIP_TRY(hipFree(v.bricks));
    if (v.cells) HIP_TRY(hipFree(v.cells));
    if (v.material) HIP_TRY(hipFree(v.material));
    if (v.skip) HIP_TRY(hipFree(v.skip));
    if (v.nib) HIP_TRY(hipFree(v.nib));
    if (v.cube_skip) HIP_TRY(hipFree(v.cube_skip));
    v = DeviceVolume();
    return VRT_OK;
}

int init_device(DeviceState& D, int ordinal) {
    D.ordinal = ordinal;
    HIP_TRY(hipSetDevice(ordinal));
    HIP_TRY(hipStreamCreateWithFlags(&D.stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&D.d_vols, sizeof(DVolume) * VRT_MAX_VOLUMES));
    HIP_TRY(hipMemset(D.d_vols, 0, sizeof(DVolume) * VRT_MAX_VOLUMES));
    HIP_TRY(hipMalloc(&D.d_inst, sizeof(DInstance) * VRT_MAX_INSTANCES));
    HIP_TRY(hipMalloc(&D.d_nodes, sizeof(DBvhNode) * kMaxBvhNodes));
    HIP_TRY(hipMalloc(&D.d_point, sizeof(DPointLight) * VRT_MAX_POINT_LIGHTS));
    HIP_TRY(hipMalloc(&D.d_spot, sizeof(DSpotLight) * VRT_MAX_SPOT_LIGHTS));
    for (int i = 0; i < kRing; i++) {
        HIP_TRY(hipEventCreate(&D.ev0[i]));
        HIP_TRY(hipEventCreate(&D.ev1[i]));
    }
    D.events_ok = true;
    return VRT_OK;
}

void destroy_device(DeviceState& D) {
    if (hipSetDevice(D.ordinal) != hipSuccess) return;
    for (int i = 0; i < VRT_MAX_VOLUMES; i++) {
        if (D.vol[i].dense) (void)hipFree(D.vol[i].dense);
        if (D.vol[i].bricks) (void)hipFree(D.vol[i].bricks);
        if (D.vol[i].cells) (void)hipFree(D.vol[i].cells);
        if (D.vol[i].material) (void)hipFree(D.vol[i].material);
        if (D.vol[i].skip) (void)hipFree(D.vol[i].skip);
        if (D.vol[i].nib) (void)hipFree(D.vol[i].nib);
        if (D.vol[i].cube_skip) (void)hipFree(D.vol[i].cube_skip);
    }
    if (D.d_vols) (void)hipFree(D.d_vols);
    if (D.d_inst) (void)hipFree(D.d_inst);
    if (D.d_nodes) (void)hipFree(D.d_nodes);
    if (D.d_point) (void)hipFree(D.d_point);
    if (D.d_spot) (void)hipFree(D.d_spot);
    if (D.d_env) (void)hipFree(D.d_env);
    for (int i = 0; i < VRT_MAX_TEXTURES; i++)
        if (D.tex[i]) (void)hipFree(D.tex[i]);
    if (D.flight_stream) (void)hipStreamDestroy(D.flight_stream);
    for (auto& S : D.slot) {
        if (S.marched) (void)hipEventDestroy(S.marched);
        if (S.done) (void)hipEventDestroy(S.done);
        if (S.d_scene) (void)hipFree(S.d_scene);
        if (S.h_scene) (void)hipHostFree(S.h_scene);
        if (S.d_fb) (void)hipFree(S.d_fb);
        if (S.h_fb) (void)hipHostFree(S.h_fb);
    }
    if (D.fb) (void)hipFree(D.fb);
    for (int i = 0; i < kStatSlots; i++)
        if (D.d_stats[i]) (void)hipFree(D.d_stats[i]);
    for (unsigned* r : D.stats_retired) (void)hipFree(r);
    for (int i = 0; i < kStatSlots; i++) {
        if (D.d_cams[i]) (void)hipFree(D.d_cams[i]);
        if (D.h_cams[i]) (void)hipHostFree(D.h_cams[i]);
        if (D.d_dyn[i]) (void)hipFree(D.d_dyn[i]);
        if (i == 0 && D.d_blockfb) (void)hipFree(D.d_blockfb);
        if (i == 0 && D.h_blockfb) (void)hipHostFree(D.h_blockfb);
        if (i == 0 && D.copy_stream) (void)hipStreamDestroy(D.copy_stream);
        if (i == 0 && D.part_done) (void)hipEventDestroy(D.part_done);
        if (D.h_dyn[i]) (void)hipHostFree(D.h_dyn[i]);
        if (D.cams_copied[i]) (void)hipEventDestroy(D.cams_copied[i]);
    }
    for (int i = 0; i < kStatSlots; i++)
        if (D.d_pass[i]) (void)hipFree(D.d_pass[i]);
    for (void* r : D.pass_retired) (void)hipFree(r);
    if (D.d_diag) (void)hipFree(D.d_diag);
    if (D.joined) (void)hipEventDestroy(D.joined);
    if (D.events_ok)
        for (int i = 0; i < kRing; i++) {
            (void)hipEventDestroy(D.ev0[i]);
            (void)hipEventDestroy(D.ev1[i]);
        }
    if (D.stream) (void)hipStreamDestroy(D.stream);
}

bool valid_slot(int slot) { return slot >= 0 && slot < VRT_MAX_VOLUMES; }

/* Uploads N^3 densities (already on the host as fp32, or as VVoxel records) to every device,
 * then re-tiles them into bricks on the device. */
/* A triangle mesh prepared for the device Voxelizer: one frame (+ voxel index box) per usable triangle. */
struct MeshSource {
    std::vector<vrt_vox::TriangleFrame> frames;
    float threshold = 0.f;
};

int upload_volume(vrt_ctx* ctx, int slot, uint8_t resolution, float extent, const float* density,
                  const uint8_t* material, const vrt_voxel* voxels, const MeshSource* mesh = nullptr, const uint8_t* texels = nullptr) {
    if (!ctx) return VRT_ERR_INVALID;
    if (!valid_slot(slot)) return VRT_ERR_SLOT;
    if (resolution > VRT_MAX_RESOLUTION || !(extent > 0.0f) || (!density && !voxels && !mesh && !texels)) return VRT_ERR_INVALID;
    const int format = texels ? VRT_FORMAT_TEXEL16 : ctx->upload_format;
    const size_t sample_bytes = format == VRT_FORMAT_TEXEL16 ? sizeof(short) : sizeof(float);
    const int N = (1 << resolution) + 1; /* VoxelVolume.cpp:23 */
    const int nb = (N - 1 + kBrickCells - 1) / kBrickCells;
    const size_t count = (size_t)N * N * N;
    for (auto& D : ctx->dev) {
        int rc = free_device_volume(D, slot);
        if (rc != VRT_OK) return rc;
        HIP_TRY(hipSetDevice(D.ordinal));
        DeviceVolume& v = D.vol[slot];
        HIP_TRY(hipMalloc(&v.dense, count * sizeof(float)));
        HIP_TRY(hipMalloc(&v.material, count));
        HIP_TRY(hipMalloc(&v.bricks, (size_t)nb * nb * nb * kBrickFloats * sample_bytes));
        if (texels) {
            /* the reference's own volume texture: decode on the device (UpdateVolumeTexture's loop, inverted) */
            void* staging = nullptr;
            HIP_TRY(hipMalloc(&staging, count * 4));
            hipError_t e = hipMemcpyAsync(staging, texels, count * 4, hipMemcpyHostToDevice, D.stream);
            if (e == hipSuccess) e = launch_texels_to_field(staging, v.dense, v.material, N, D.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(D.stream);
            (void)hipFree(staging);
            HIP_TRY(e);
        } else if (mesh) {
            /* device Voxelizer: background 2*extent everywhere, then the minimum over the triangles */
            void* d_frames = nullptr;
            const size_t fbytes = mesh->frames.size() * sizeof(vrt_vox::TriangleFrame);
            hipError_t e = hipSuccess;
            if (fbytes > 0) {
                e = hipMalloc(&d_frames, fbytes);
                if (e == hipSuccess) e = hipMemcpyAsync(d_frames, mesh->frames.data(), fbytes, hipMemcpyHostToDevice, D.stream);
            }
            const float cell = (extent * 2.0f) / (float)(N - 1);
            if (e == hipSuccess)
                e = launch_voxelize(d_frames, mesh->frames.size(), v.dense, v.material, N, cell, extent, mesh->threshold, D.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(D.stream);
            if (d_frames) (void)hipFree(d_frames);
            HIP_TRY(e);
        } else if (voxels) {
            void* staging = nullptr;
            HIP_TRY(hipMalloc(&staging, count * sizeof(vrt_voxel)));
            hipError_t e = hipMemcpyAsync(staging, voxels, count * sizeof(vrt_voxel), hipMemcpyHostToDevice, D.stream);
            if (e == hipSuccess) e = launch_split_voxels(staging, v.dense, v.material, count, D.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(D.stream);
            (void)hipFree(staging);
            HIP_TRY(e);
        } else {
            HIP_TRY(hipMemcpyAsync(v.dense, density, count * sizeof(float), hipMemcpyHostToDevice, D.stream));
            if (material)
                HIP_TRY(hipMemcpyAsync(v.material, material, count, hipMemcpyHostToDevice, D.stream));
            else
                HIP_TRY(hipMemsetAsync(v.material, 0, count, D.stream));
        }
        /* VRT_FORMAT_TEXEL16: quantise like VDXVoxelVolume::EncodeVoxel; the dense grid then holds the integer field too,
           so that every data path and every table sees the same values */
        if (format == VRT_FORMAT_TEXEL16 && !texels) HIP_TRY(launch_quantize_field(v.dense, count, D.stream));
        HIP_TRY(launch_retile(v.dense, v.bricks, format, N, nb, D.stream));
        if (format == VRT_FORMAT_TEXEL16) {
            HIP_TRY(hipMalloc(&v.cells, (size_t)nb * nb * nb * 64 * 16));
            HIP_TRY(launch_retile_cells16(v.dense, v.cells, N, nb, D.stream));
        }
        const size_t nbricks = (size_t)nb * nb * nb;
        HIP_TRY(hipMalloc(&v.cube_skip, 2 * nbricks));
        HIP_TRY(launch_cube_table(v.dense, v.cube_skip, v.cube_skip + nbricks, N, nb, D.stream));
        HIP_TRY(hipStreamSynchronize(D.stream));
    }
    HostVolume& h = ctx->vol[slot];
    const bool was_used = h.used;
    h.used = true;
    h.resolution = resolution;
    h.N = N;
    h.nb = nb;
    h.extent = extent;
    h.format = format;
    if (!was_used) {
        h.density_scale = 1.0f;
        h.step_max = 0.0f;
        h.mat = vrt_material{{0.8f, 0.8f, 0.8f, 1.0f}, 0.8f, 0.0f};
        h.tex[0] = h.tex[1] = h.tex[2] = -1;
        h.tex_scale[0] = h.tex_scale[1] = 100.f;
    }
    ctx->scene_stale = true;
    int rc = rebuild_skip(ctx, slot);
    if (rc != VRT_OK) return rc;
    return sync_volume_table(ctx);
}

/* A scene's small arrays as the kernels read them: instances (with the directional light's shadow-ray constants), the threaded
   instance BVH, point and spot lights.  What VRDXScene::PrepareForRendering re-sends every frame (RDXScene.cpp:150-174, 454-545,
   726-755). */
void pack_scene_arrays(const vrt_ctx* ctx, const vrt_scene& s, DInstance* inst, DBvhNode* nodes, int& n_nodes, DPointLight* point,
                       DSpotLight* spot) {
    std::vector<Box> boxes((size_t)s.n_instances);
    for (int i = 0; i < s.n_instances; i++) {
        pack_instance(s.instances[i], inst[i]);
        pack_shadow_ray(s.light_dir, inst[i]);
        boxes[(size_t)i] = instance_box(inst[i], ctx->vol[s.instances[i].volume_slot].extent);
    }
    n_nodes = 0;
    if (s.n_instances > 0) {
        std::vector<int> ids((size_t)s.n_instances);
        for (int i = 0; i < s.n_instances; i++) ids[(size_t)i] = i;
        build_bvh(ids, 0, s.n_instances, boxes, nodes, n_nodes);
    }
    const int npl = std::min(s.n_point_lights, VRT_MAX_POINT_LIGHTS);
    for (int i = 0; i < npl; i++) {
        const vrt_point_light& L = s.point_lights[i];
        DPointLight& o = point[i];
        memset(&o, 0, sizeof o);
        memcpy(o.pos, L.position, sizeof o.pos);
        memcpy(o.color, L.color, sizeof o.color);
        o.intensity = L.intensity;
        o.att_linear = L.att_linear;
        o.att_exp = L.att_exp;
    }
    const int nsl = std::min(s.n_spot_lights, VRT_MAX_SPOT_LIGHTS);
    for (int i = 0; i < nsl; i++) {
        const vrt_spot_light& L = s.spot_lights[i];
        DSpotLight& o = spot[i];
        memset(&o, 0, sizeof o);
        memcpy(o.pos, L.position, sizeof o.pos);
        memcpy(o.fwd, L.forward, sizeof o.fwd);
        memcpy(o.color, L.color, sizeof o.color);
        o.intensity = L.intensity;
        o.att_linear = L.att_linear;
        o.att_exp = L.att_exp;
        o.cos_angle = L.cos_angle;
        o.cos_falloff = L.cos_falloff_angle;
    }
}

/* The packed scene's arrays -> every device (synchronous copies on the null stream). */
int upload_scene_arrays(vrt_ctx* ctx) {
    const vrt_scene& s = ctx->scene;
    const int npl = std::min(s.n_point_lights, VRT_MAX_POINT_LIGHTS);
    const int nsl = std::min(s.n_spot_lights, VRT_MAX_SPOT_LIGHTS);
    for (auto& D : ctx->dev) {
        HIP_TRY(hipSetDevice(D.ordinal));
        if (s.n_instances > 0) {
            HIP_TRY(hipMemcpy(D.d_inst, ctx->inst, sizeof(DInstance) * (size_t)s.n_instances, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(D.d_nodes, ctx->nodes, sizeof(DBvhNode) * (size_t)ctx->n_nodes, hipMemcpyHostToDevice));
        }
        if (npl > 0) HIP_TRY(hipMemcpy(D.d_point, ctx->point, sizeof(DPointLight) * (size_t)npl, hipMemcpyHostToDevice));
        if (nsl > 0) HIP_TRY(hipMemcpy(D.d_spot, ctx->spot, sizeof(DSpotLight) * (size_t)nsl, hipMemcpyHostToDevice));
    }
    ctx->arrays_uploaded = true;
    return VRT_OK;
}

int pack_scene(vrt_ctx* ctx) {
    pack_scene_arrays(ctx, ctx->scene, ctx->inst, ctx->nodes, ctx->n_nodes, ctx->point, ctx->spot);
    ctx->arrays_uploaded = false;
    /* While frames are in flight (vrt_render_begin: each carries its own snapshot of these arrays, copied on its own stream) the
       device copies of the context are not what the next frame reads: they are brought up to date by the first launch that needs
       them.  A synchronous copy here would queue behind a frame's march and stall the application for its whole duration — the
       reference's per-frame scene update (RendererEngineInstance.cpp:111-130) would cost the overlap of march and read-back. */
    for (auto& D : ctx->dev)
        for (const FrameSlot& S : D.slot)
            if (S.busy) return VRT_OK;
    return upload_scene_arrays(ctx);
}

/* What vrt_scene_set refuses (and vrt_render_block, per frame of vrt_block::scenes). */
int validate_scene(const vrt_ctx* ctx, const vrt_scene* scene) {
    if (!scene) return VRT_ERR_INVALID;
    if (scene->n_instances < 0 || scene->n_instances > VRT_MAX_INSTANCES) return VRT_ERR_INVALID;
    if (scene->n_point_lights < 0 || scene->n_spot_lights < 0) return VRT_ERR_INVALID;
    for (int i = 0; i < scene->n_instances; i++) {
        const vrt_instance& I = scene->instances[i];
        if (I.volume_slot < 0 || I.volume_slot >= VRT_MAX_VOLUMES || !ctx->vol[I.volume_slot].used) return VRT_ERR_SLOT;
        if (I.scale[0] == 0.0f || I.scale[1] == 0.0f || I.scale[2] == 0.0f) return VRT_ERR_INVALID;
    }
    return VRT_OK;
}

int check_params(const vrt_ctx* ctx, const vrt_params* p, bool own_scenes = false) {
    if (!ctx || !p) return VRT_ERR_INVALID;
    if (p->width <= 0 || p->height <= 0 || p->width > 16384 || p->height > 16384) return VRT_ERR_INVALID;
    if (p->max_steps < 0 || p->max_steps > 65535 || p->max_bounces < 0 || p->max_bounces > 2) return VRT_ERR_INVALID;
    if (!(p->eps_hit == p->eps_hit) || !(p->step_min > 0.0f) || !(p->k_relax > 0.0f) || !(p->eps_in >= 0.0f) ||
        !(p->cone_eps >= 0.0f))
        return VRT_ERR_INVALID;
    if (p->mode < VRT_MODE_INTERP || p->mode > VRT_MODE_CUBE_NOTEX_UNLIT) return VRT_ERR_INVALID;
    if (p->path < VRT_PATH_AUTO || p->path > VRT_PATH_CELLS) return VRT_ERR_INVALID;
    /* (bit 5 was round 1's VRT_FLAG_SKIP_EMPTY: empty-space skipping is always on now; the bit is accepted and ignored) */
    if ((p->flags & ~(3 | VRT_FLAG_DIAG_TIMELINE | VRT_FLAG_OUTPUT_RGBA8 | VRT_FLAG_NO_TIMING | 32 | VRT_FLAG_BLOCK_PER_FRAME | VRT_FLAG_NO_CULL_RECT |
                      VRT_FLAG_FULL_ONE_KERNEL | VRT_FLAG_FULL_THREE_PASS | VRT_FLAG_NO_HIT_POLISH | VRT_FLAG_OUTPUT_BGRA8 |
                      VRT_FLAG_REFERENCE_VIEW_VECTOR | VRT_FLAG_REFERENCE_BOUNDARY_TEXELS)) != 0 || (p->flags & 3) == 3 ||
        ((p->flags & VRT_FLAG_OUTPUT_BGRA8) && !(p->flags & VRT_FLAG_OUTPUT_RGBA8)) ||
        ((p->flags & VRT_FLAG_FULL_ONE_KERNEL) && (p->flags & VRT_FLAG_FULL_THREE_PASS)))
        return VRT_ERR_INVALID;
    if (!ctx->have_scene && !own_scenes) return VRT_ERR_NOT_READY;
    return VRT_OK;
}

/* The internal data path of a launch.  AUTO = bricks.  Bricks come in the format of the scene's volumes: fp32, or int16
 * when every instanced volume is VRT_FORMAT_TEXEL16; a scene that mixes the two marches the dense grids (which hold the
 * same values in either format).  The LDS brick cache is wave-cooperative and covers single-instance fp32 scenes.  Returns
 * a negative status for the one combination without a kernel: a Cube mode over mixed formats. */
int resolve_path(const vrt_ctx* ctx, int path, bool single, int mode, const vrt_scene* scenes = nullptr, int n_scenes = 0) {
    int n16 = 0, n32 = 0;
    if (!scenes) {
        scenes = &ctx->scene;
        n_scenes = 1;
    }
    for (int f = 0; f < n_scenes; f++)
        for (int i = 0; i < scenes[f].n_instances; i++)
            (ctx->vol[scenes[f].instances[i].volume_slot].format == VRT_FORMAT_TEXEL16 ? n16 : n32)++;
    const bool mixed = n16 > 0 && n32 > 0, all16 = n16 > 0 && n32 == 0;
    if (mode >= VRT_MODE_CUBE) { /* exact grid traversal over the bricks, whatever path was asked for */
        if (mixed) return VRT_ERR_UNSUPPORTED;
        return all16 ? kPathCube16 : kPathCube;
    }
    if (path == VRT_PATH_DENSE || mixed) return VRT_PATH_DENSE;
    if (all16) return path == VRT_PATH_CELLS ? kPathCells16 : kPathBrick16;
    if (path == VRT_PATH_BRICK_LDS && single) return VRT_PATH_BRICK_LDS;
    return VRT_PATH_BRICK;
}

/* Rows of one launch: contiguous [row0, row0+rows), or (strip_rows > 0) n_strips interleaved strips. */
struct RowSet {
    int row0 = 0, rows = 0;
    int strip_rows = 0, strip_first = 0, strip_stride = 0;
};

/* Screen rectangle of everything a primary ray can reach (DFrame::cull_*): the corners of every instance's active box
   (object space -> world -> camera -> pixel), double precision, two pixels of margin.  Any corner at or behind the camera
   plane: the whole frame. */
void cull_rect(const vrt_ctx* ctx, const vrt_params* p, DCam& F, const vrt_scene* scene = nullptr, const DInstance* inst = nullptr) {
    if (!scene) {
        scene = &ctx->scene;
        inst = ctx->inst;
    }
    /* (frames are at most 16384 pixels wide and high: 16 bits per coordinate) */
    F.cull_lo = pack_cull(0, 0);
    F.cull_hi = pack_cull(p->width - 1, p->height - 1);
    if (p->flags & VRT_FLAG_NO_CULL_RECT) return;
    double x0 = 1e30, y0 = 1e30, x1 = -1e30, y1 = -1e30;
    for (int i = 0; i < scene->n_instances; i++) {
        const HostVolume& h = ctx->vol[scene->instances[i].volume_slot];
        const DInstance& I = inst[i];
        double lo[3], hi[3];
        const double cell = ((double)h.extent * 2.0) / (double)(h.N - 1);
        /* The kernel clips a ray to the active box only when its whole interval ends before t_skip_end = (smax/2 - eps_hit) /
           cone_eps, smax = step_max / |w2o d| >= step_max * min|scale| (clip_to_active_box).  The rectangle may rely on the clip
           only if that holds for EVERY primary ray: the farthest corner of the volume box is nearer than the smallest t_skip_end */
        bool clip_everywhere = true;
        if (h.step_max > 0.0f) {
            const vrt_instance& in = scene->instances[i];
            const double min_scale = std::min(std::min(fabs((double)in.scale[0]), fabs((double)in.scale[1])), fabs((double)in.scale[2]));
            const double smax_min = (double)h.step_max * min_scale;
            double far2 = 0.0;
            for (int k = 0; k < 8; k++) {
                const double c[3] = {(k & 1) ? (double)h.extent : -(double)h.extent, (k & 2) ? (double)h.extent : -(double)h.extent,
                                     (k & 4) ? (double)h.extent : -(double)h.extent};
                double d2 = 0.0;
                for (int a = 0; a < 3; a++) {
                    const double w = (double)I.o2w[a * 3 + 0] * c[0] + (double)I.o2w[a * 3 + 1] * c[1] + (double)I.o2w[a * 3 + 2] * c[2] + (double)I.pos[a] - (double)F.cam_o[a];
                    d2 += w * w;
                }
                far2 = std::max(far2, d2);
            }
            const double t_far = sqrt(far2) * 1.001;
            clip_everywhere = p->cone_eps > 0.0f ? t_far <= (0.5 * smax_min * 0.999 - (double)p->eps_hit) / (double)p->cone_eps
                                                 : 2.0 * (double)p->eps_hit <= smax_min * 0.999;
        }
        for (int a = 0; a < 3; a++) {
            lo[a] = -(double)h.extent;
            hi[a] = (double)h.extent;
            if (h.step_max > 0.0f && p->mode < VRT_MODE_CUBE && clip_everywhere) { /* the sphere-trace is clipped to the active box (a little
                                                                   slack for its float rounding); the Cube modes visit the whole volume box */
                const int ax = a == 0 ? 0 : (a == 1 ? 2 : 1);
                lo[a] = std::max(lo[a], (double)(h.abox[ax] * kBrickCells) * cell - (double)h.extent - 0.01 * cell);
                hi[a] = std::min(hi[a], (double)std::min((h.abox[3 + ax] + 1) * kBrickCells, h.N - 1) * cell - (double)h.extent + 0.01 * cell);
            }
        }
        if (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) continue; /* nothing to hit in this volume */
        for (int k = 0; k < 8; k++) {
            const double c[3] = {(k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2]};
            double w[3];
            for (int a = 0; a < 3; a++)
                w[a] = (double)I.o2w[a * 3 + 0] * c[0] + (double)I.o2w[a * 3 + 1] * c[1] + (double)I.o2w[a * 3 + 2] * c[2] + (double)I.pos[a] - (double)F.cam_o[a];
            const double ca = w[0] * F.r0[0] + w[1] * F.r0[1] + w[2] * F.r0[2];
            const double cb = w[0] * F.r1[0] + w[1] * F.r1[1] + w[2] * F.r1[2];
            const double cc = -(w[0] * F.r2[0] + w[1] * F.r2[1] + w[2] * F.r2[2]); /* depth along the view direction */
            if (!(cc > 1e-6 * (fabs(ca) + fabs(cb) + 1.0))) return; /* at or behind the camera plane: no culling */
            const double sx = (ca / cc) / (double)F.cx, sy = -(cb / cc) / (double)F.cy;
            const double px = (sx + 1.0) * 0.5 * (double)p->width - 0.5, py = (sy + 1.0) * 0.5 * (double)p->height - 0.5;
            x0 = std::min(x0, px);
            x1 = std::max(x1, px);
            y0 = std::min(y0, py);
            y1 = std::max(y1, py);
        }
    }
    const double m = 2.0;
    const int cx0 = (int)std::max(0.0, std::min((double)p->width, floor(x0 - m)));
    const int cy0 = (int)std::max(0.0, std::min((double)p->height, floor(y0 - m)));
    const int cx1 = (int)std::max(-1.0, std::min((double)p->width - 1.0, ceil(x1 + m)));
    const int cy1 = (int)std::max(-1.0, std::min((double)p->height - 1.0, ceil(y1 + m)));
    if (x1 < x0 || cx1 < cx0 || cy1 < cy0) { /* no instance can be hit, or everything projects off the screen: an empty rectangle */
        F.cull_lo = pack_cull(1, 1);
        F.cull_hi = pack_cull(0, 0);
        return;
    }
    F.cull_lo = pack_cull(cx0, cy0);
    F.cull_hi = pack_cull(cx1, cy1);
}

/* Which closest-hit kernel a frame of this scene needs. */
struct ClosestHitForm {
    bool textured, full, may_bounce;
};
ClosestHitForm form_of_scene(const vrt_ctx* ctx, const vrt_params* p, const vrt_scene& sc) {
    /* the lean kernel covers directional light + shadow; the full closest hit is only launched when the
       frame can need it: extra lights, or bounces allowed and some instanced material mirrors (roughness < 0.3) */
    /* textured modes read the material textures; a frame needs that code only when a bound texture is in sight */
    const bool tex_mode = p->mode == VRT_MODE_INTERP || p->mode == VRT_MODE_INTERP_UNLIT || p->mode == VRT_MODE_CUBE ||
                          p->mode == VRT_MODE_CUBE_UNLIT;
    bool smooth = false;
    bool textured = false, images = false; /* a bound texture in sight; one that is more than a single texel */
    for (int i = 0; i < sc.n_instances; i++) {
        const HostVolume& hv = ctx->vol[sc.instances[i].volume_slot];
        float rough = hv.mat.roughness;
        for (int k = 0; tex_mode && k < 3; k++) {
            if (hv.tex[k] < 0 || !ctx->tex[hv.tex[k]].used) continue;
            const HostTexture& t = ctx->tex[hv.tex[k]];
            textured = true;
            if (t.width != 1 || t.height != 1) images = true;
            /* a constant RM texel scales the roughness by its red channel — times a tri-planar blend sum that is 1 to a few ulp: the
               lower bound decides whether the material can mirror */
            else if (k == 2) rough = hv.mat.roughness * ((float)t.first[0] / 255.0f) * (hv.mat.roughness >= 0.0f ? 0.99999f : 1.00001f);
        }
        smooth = smooth || std::min(std::max(rough, 0.0f), 1.0f) < 0.3f;
    }
    /* A 1x1 texture is a constant (the reference's default normal texel on every material without a normal map, RDXScene.cpp:241-260):
       folded into the lean kernel's REF instantiation — no fetch, no full closest hit.  Only real images need the texture code. */
    ClosestHitForm f;
    f.textured = textured;
    f.full = sc.n_point_lights > 0 || sc.n_spot_lights > 0 || (p->max_bounces > 0 && smooth) || images;
    f.may_bounce = p->max_bounces > 0 && (smooth || images); /* (a roughness IMAGE can make any material mirror) */
    return f;
}
/* ... of a launch: the scene of vrt_scene_set, or — a block over per-frame scenes (vrt_block::scenes) — the union of what its frames
   need (the full closest hit renders a frame without extra lights like the lean kernel does: same pixels, same counters). */
ClosestHitForm closest_hit_form(const vrt_ctx* ctx, const vrt_params* p, const vrt_scene* scenes = nullptr, int n_scenes = 0) {
    if (!scenes) return form_of_scene(ctx, p, ctx->scene);
    ClosestHitForm u = {false, false, false};
    for (int f = 0; f < n_scenes; f++) {
        const ClosestHitForm x = form_of_scene(ctx, p, scenes[f]);
        u.textured = u.textured || x.textured;
        u.full = u.full || x.full;
        u.may_bounce = u.may_bounce || x.may_bounce;
    }
    return u;
}

void build_frame(const vrt_ctx* ctx, const DeviceState& D, const vrt_params* p, const RowSet& rs, float* out,
                 unsigned* stats, DFrame& F, const SceneArrays* snapshot = nullptr, const vrt_scene* scenes = nullptr, int n_scenes = 0) {
    const int row0 = rs.row0, rows = rs.rows;
    memset(&F, 0, sizeof F);
    F.inv_w = 1.0f / (float)p->width;
    F.inv_h = 1.0f / (float)p->height;
    F.n_frames = 1;
    F.light_dir[0] = ctx->scene.light_dir[0];
    F.light_dir[1] = ctx->scene.light_dir[1];
    F.light_dir[2] = ctx->scene.light_dir[2];
    F.light_strength = ctx->scene.light_strength;
    F.eps_hit = p->eps_hit;
    F.eps_in = p->eps_in;
    F.step_min = p->step_min;
    F.k_relax = p->k_relax;
    F.cone_eps = p->cone_eps;
    F.max_steps = p->max_steps;
    F.shadow = p->shadow ? 1 : 0;
    F.unlit = (p->mode == VRT_MODE_INTERP_UNLIT || p->mode == VRT_MODE_INTERP_NOTEX_UNLIT || p->mode == VRT_MODE_CUBE_UNLIT ||
               p->mode == VRT_MODE_CUBE_NOTEX_UNLIT)
                  ? 1
                  : 0;
    F.back = p->mode >= VRT_MODE_CUBE ? 0.2f : 0.1f; /* Raytracing.hlsl:52 / Raytracing_Cube.hlsl:52 */
    F.max_bounces = p->max_bounces;
    F.width = p->width;
    F.height = p->height;
    F.row0 = row0;
    F.rows = rows;
    F.tiles_x = (p->width + 15) / 16;
    F.tiles_y = (rows + 15) / 16;
    F.tile_map = p->flags & 3;
    F.diag = (p->flags & VRT_FLAG_DIAG_TIMELINE) ? 1 : 0;
    F.rgba8 = (p->flags & VRT_FLAG_OUTPUT_RGBA8) ? ((p->flags & VRT_FLAG_OUTPUT_BGRA8) ? 2 : 1) : 0;
    F.polish = (p->flags & VRT_FLAG_NO_HIT_POLISH) ? 0 : VRT_HIT_POLISH_SAMPLES;
    F.view_vec = (p->flags & VRT_FLAG_REFERENCE_VIEW_VECTOR) ? 1 : 0;
    F.zero_outside = (p->flags & VRT_FLAG_REFERENCE_BOUNDARY_TEXELS) ? 1 : 0;
    F.strip_rows = rs.strip_rows;
    F.strip_first = rs.strip_first;
    F.strip_stride = rs.strip_stride;
    const ClosestHitForm form = closest_hit_form(ctx, p, scenes, n_scenes);
    F.textured = form.textured ? 1 : 0;
    F.full = form.full ? 1 : 0;
    F.may_bounce = form.may_bounce ? 1 : 0;
    F.n_inst = ctx->scene.n_instances;
    F.n_nodes = ctx->n_nodes;
    F.n_point = std::min(ctx->scene.n_point_lights, VRT_MAX_POINT_LIGHTS);
    F.n_spot = std::min(ctx->scene.n_spot_lights, VRT_MAX_SPOT_LIGHTS);
    F.vols = snapshot ? snapshot->vols : D.d_vols;
    F.inst = snapshot ? snapshot->inst : D.d_inst;
    F.nodes = snapshot ? snapshot->nodes : D.d_nodes;
    F.point = snapshot ? snapshot->point : D.d_point;
    F.spot = snapshot ? snapshot->spot : D.d_spot;
    F.env = ctx->env_size > 0 ? D.d_env : nullptr;
    F.env_size = ctx->env_size;
    F.out = out;
    F.stats = stats;
    F.vol0 = ctx->scene.n_instances == 1 ? F.vols + ctx->scene.instances[0].volume_slot : nullptr;
}

/* Enqueue n_frames frames of one tile on one device as ONE launch (grid.y = frame): frame f from cams[f] (null: the scene's own
   camera) into out + f * frame_stride bytes.  No allocation after the stream's first launch of that size, no host sync. */
int enqueue_rows(vrt_ctx* ctx, DeviceState& D, const vrt_params* p, const RowSet& rs, float* out, hipStream_t stream,
                 int ring, const SceneArrays* snapshot = nullptr, int n_frames = 1, const vrt_camera* cams = nullptr, size_t frame_stride = 0,
                 const vrt_scene* scenes = nullptr) {
    if (n_frames < 1 || n_frames > kMaxLaunchFrames) return VRT_ERR_INVALID;
    /* more cameras than the kernarg segment holds, or per-frame scene state (its records live in device memory, the cameras with them) */
    const bool device_cams = n_frames > kMaxBlockFrames || scenes != nullptr;
    DBlock B;
    memset(B.cam, 0, sizeof B.cam); /* (the whole struct travels as the kernarg: no stale stack bytes behind the block's frames) */
    DFrame& F = B.f;
    build_frame(ctx, D, p, rs, out, nullptr, F, snapshot, scenes, n_frames);
    if ((long long)F.tiles_x * F.tiles_y > kMaxBlocks / 2) return VRT_ERR_INVALID;
    bool single = ctx->scene.n_instances == 1;
    if (scenes) { /* one kernel form for the launch: the single-instance one only when every frame has exactly one instance */
        if (snapshot || (p->flags & VRT_FLAG_DIAG_TIMELINE)) return VRT_ERR_INVALID;
        single = true;
        for (int f = 0; f < n_frames; f++) single = single && scenes[f].n_instances == 1;
        F.vol0 = nullptr;
    }
    const int path = resolve_path(ctx, p->path, single, p->mode, scenes, n_frames);
    if (path < 0) return path;
    if (!snapshot && !scenes && !ctx->arrays_uploaded) { /* deferred by pack_scene (frames were in flight): now */
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream != nullptr && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return VRT_ERR_NOT_READY;
        const int rc = upload_scene_arrays(ctx);
        if (rc != VRT_OK) return rc;
    }
    D.last_blocks = grid_blocks(F.tiles_x, F.tiles_y, F.tile_map);
    if ((long long)D.last_blocks * n_frames > 8LL * kMaxBlocks) return VRT_ERR_INVALID;
    D.last_frames = n_frames;
    F.n_frames = n_frames;
    F.frame_stride = frame_stride;
    F.stats_stride = (uint32_t)((size_t)D.last_blocks * 4 * kStatRecord);
    const vrt_camera own = scene_camera(ctx->scene);
    for (int f = 0; f < n_frames && !device_cams; f++) {
        pack_camera(cams ? cams[f] : own, p->width, p->height, B.cam[f]);
        cull_rect(ctx, p, B.cam[f]);
    }
    const size_t stat_blocks = (size_t)D.last_blocks * (size_t)n_frames; /* one record set per (frame, wave) */
    int slot = -1;
    for (int i = 0; i < kStatSlots; i++)
        if (D.stats_bound[i] && D.stats_stream[i] == stream) slot = i;
    if (slot < 0) { /* a stream not seen lately: take the least recently used slot */
        slot = 0;
        for (int i = 1; i < kStatSlots; i++) {
            const bool freer = !D.stats_bound[i] && D.stats_bound[slot];
            const bool older = D.stats_bound[i] == D.stats_bound[slot] && D.stats_used[i] < D.stats_used[slot];
            if (freer || older) slot = i;
        }
        /* the slot's previous stream may still be running a full closest hit in passes, or a launch with its cameras in device memory:
           hit records and camera records (unlike the counters) decide pixels: wait for the device once before another stream takes
           them over (a 17th stream: rare) */
        if (D.stats_bound[slot] && (D.d_pass[slot] != nullptr || D.d_cams[slot] != nullptr)) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (stream == nullptr || hipStreamIsCapturing(stream, &cs) != hipSuccess || cs == hipStreamCaptureStatusNone)
                HIP_TRY(hipDeviceSynchronize());
        }
        D.stats_bound[slot] = true;
        D.stats_stream[slot] = stream;
    }
    D.stats_used[slot] = ctx->launches;
    if (D.stats_cap[slot] < stat_blocks) { /* first launch of this size on this stream: the only allocation on this path */
        unsigned* grown = nullptr;
        HIP_TRY(hipMalloc(&grown, sizeof(unsigned) * kStatRecord * 4 * stat_blocks));
        if (D.d_stats[slot]) D.stats_retired.push_back(D.d_stats[slot]); /* a captured graph may still write to it */
        D.d_stats[slot] = grown;
        D.stats_cap[slot] = stat_blocks;
    }
    D.last_slot = slot;
    F.stats = D.d_stats[slot];
    if (device_cams) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream != nullptr && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return VRT_ERR_INVALID; /* (a replay would copy whatever the pinned records hold by then; vrt_render_block captures in kernarg-sized launches) */
        if (!D.d_cams[slot]) {
            HIP_TRY(hipMalloc(&D.d_cams[slot], sizeof(DCam) * kMaxLaunchFrames));
            HIP_TRY(hipHostMalloc(&D.h_cams[slot], sizeof(DCam) * kMaxLaunchFrames));
            HIP_TRY(hipEventCreateWithFlags(&D.cams_copied[slot], hipEventDisableTiming));
        } else {
            HIP_TRY(hipEventSynchronize(D.cams_copied[slot])); /* the previous launch's copy has read the pinned records (long ago, normally) */
        }
        if (scenes && !D.d_dyn[slot]) {
            HIP_TRY(hipMalloc(&D.d_dyn[slot], (size_t)kDynStride * kMaxLaunchFrames));
            HIP_TRY(hipHostMalloc(&D.h_dyn[slot], (size_t)kDynStride * kMaxLaunchFrames));
        }
        for (int f = 0; f < n_frames; f++) {
            if (!scenes) {
                pack_camera(cams ? cams[f] : own, p->width, p->height, D.h_cams[slot][f]);
                cull_rect(ctx, p, D.h_cams[slot][f]);
                continue;
            }
            /* frame f's scene as vrt_scene_set would pack it (instances, BVH rebuilt per frame like the reference's TLAS,
               DXRenderer.cpp:809-825; lights), its camera and its own cull rectangle */
            const vrt_scene& sc = scenes[f];
            char* sec = D.h_dyn[slot] + (size_t)f * kDynStride;
            DDyn* rec = reinterpret_cast<DDyn*>(sec);
            DInstance* inst = reinterpret_cast<DInstance*>(sec + kDynInstOff);
            memset(rec, 0, sizeof *rec);
            pack_scene_arrays(ctx, sc, inst, reinterpret_cast<DBvhNode*>(sec + kDynNodesOff), rec->n_nodes,
                              reinterpret_cast<DPointLight*>(sec + kDynPointOff), reinterpret_cast<DSpotLight*>(sec + kDynSpotOff));
            memcpy(rec->light_dir, sc.light_dir, sizeof rec->light_dir);
            rec->light_strength = sc.light_strength;
            rec->n_inst = sc.n_instances;
            rec->n_point = std::min(sc.n_point_lights, VRT_MAX_POINT_LIGHTS);
            rec->n_spot = std::min(sc.n_spot_lights, VRT_MAX_SPOT_LIGHTS);
            rec->vol0_slot = sc.n_instances > 0 ? sc.instances[0].volume_slot : 0;
            pack_camera(scene_camera(sc), p->width, p->height, D.h_cams[slot][f]);
            cull_rect(ctx, p, D.h_cams[slot][f], &sc, inst);
        }
        HIP_TRY(hipMemcpyAsync(D.d_cams[slot], D.h_cams[slot], sizeof(DCam) * (size_t)n_frames, hipMemcpyHostToDevice, stream));
        if (scenes) {
            HIP_TRY(hipMemcpyAsync(D.d_dyn[slot], D.h_dyn[slot], (size_t)kDynStride * (size_t)n_frames, hipMemcpyHostToDevice, stream));
            F.dyn = D.d_dyn[slot];
        }
        HIP_TRY(hipEventRecord(D.cams_copied[slot], stream));
        F.cams = D.d_cams[slot];
    }
    /* the full closest hit of a block of frames runs as three passes (vrt_kernels.hip, primary_pass_kernel); a lone frame as one
       kernel: three launches would pay a launch's latency-bound tail three times */
    const bool passes = F.full && !F.diag && ((n_frames > 1 && !(p->flags & VRT_FLAG_FULL_ONE_KERNEL)) || (p->flags & VRT_FLAG_FULL_THREE_PASS));
    if (passes) {
        const size_t per_frame = (size_t)D.last_blocks * 256, records = per_frame * (size_t)n_frames;
        if (per_frame > 0xffffffffull) return VRT_ERR_INVALID;
        if (D.pass_cap[slot] < records) {
            void* grown = nullptr;
            HIP_TRY(hipMalloc(&grown, records * (sizeof(HitRecord) + sizeof(unsigned)) + records / 64 * sizeof(unsigned long long)));
            if (D.d_pass[slot]) D.pass_retired.push_back(D.d_pass[slot]);
            D.d_pass[slot] = grown;
            D.pass_cap[slot] = records;
        }
        char* base = static_cast<char*>(D.d_pass[slot]);
        F.hit_rec = reinterpret_cast<HitRecord*>(base);
        F.hit_mask = reinterpret_cast<unsigned long long*>(base + D.pass_cap[slot] * sizeof(HitRecord));
        F.hit_aux = reinterpret_cast<unsigned*>(base + D.pass_cap[slot] * sizeof(HitRecord) + D.pass_cap[slot] / 64 * sizeof(unsigned long long));
        F.rec_stride = (uint32_t)per_frame;
    }
    D.last_diag = F.diag != 0;
    D.last_form = (F.full ? VRT_FORM_FULL : 0) | (passes ? VRT_FORM_PASSES : 0) | (F.textured ? VRT_FORM_TEXTURED : 0) |
                  (F.may_bounce ? VRT_FORM_MAY_BOUNCE : 0) | ((!F.full && (F.textured || F.view_vec || F.zero_outside)) ? VRT_FORM_LEAN_REF : 0);
    if (F.diag) {
        if (stat_blocks > (size_t)kMaxBlocks) return VRT_ERR_INVALID; /* the timeline buffer holds kMaxBlocks workgroups */
        if (!D.d_diag) HIP_TRY(hipMalloc(&D.d_diag, sizeof(unsigned) * kDiagRecord * 4 * (size_t)kMaxBlocks));
        F.diag_buf = D.d_diag;
    }
    /* a launch that is being captured into a hipGraph is not timed: the event pair would become graph nodes and
       never hold a time stamp of its own (vrt_last_timing / vrt_timing_history report 0 ms for it) */
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream != nullptr && hipStreamIsCapturing(stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    D.timed[ring] = cap == hipStreamCaptureStatusNone && !(p->flags & VRT_FLAG_NO_TIMING);
    D.ring_frames[ring] = n_frames;
    if (D.timed[ring]) HIP_TRY(hipEventRecord(D.ev0[ring], stream));
    HIP_TRY(launch_march(B, path, single, stream));
    if (D.timed[ring]) HIP_TRY(hipEventRecord(D.ev1[ring], stream));
    return VRT_OK;
}

}  // namespace

extern "C" {

int vrt_create(vrt_ctx** out, int device_count, const int* devices) {
    if (!out || device_count < 1 || device_count > VRT_MAX_DEVICES) return VRT_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return VRT_ERR_NO_DEVICE;
    for (int i = 0; i < device_count; i++) {
        int ord = devices ? devices[i] : i;
        if (ord < 0 || ord >= n) return VRT_ERR_NO_DEVICE;
    }
    vrt_ctx* ctx = new (std::nothrow) vrt_ctx;
    if (!ctx) return VRT_ERR_OOM;
    memset(&ctx->scene, 0, sizeof ctx->scene);
    ctx->dev.resize((size_t)device_count);
    for (int i = 0; i < device_count; i++) {
        int rc = init_device(ctx->dev[(size_t)i], devices ? devices[i] : i);
        if (rc != VRT_OK) {
            for (auto& D : ctx->dev) destroy_device(D);
            delete ctx;
            return rc;
        }
    }
    /* peer access for the tile gather (xGMI); failure is not fatal, hipMemcpyPeer still works */
    for (int i = 1; i < device_count; i++) {
        if (ctx->dev[(size_t)i].ordinal == ctx->dev[0].ordinal) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, ctx->dev[(size_t)i].ordinal, ctx->dev[0].ordinal) == hipSuccess && can) {
            (void)hipSetDevice(ctx->dev[(size_t)i].ordinal);
            const hipError_t e = hipDeviceEnablePeerAccess(ctx->dev[0].ordinal, 0);
            ctx->dev[(size_t)i].peer_to_first = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            (void)hipGetLastError();
        }
    }
    *out = ctx;
    return VRT_OK;
}

int vrt_destroy(vrt_ctx* ctx) {
    if (!ctx) return VRT_ERR_INVALID;
    (void)vrt_comm_destroy(ctx);
    for (auto& D : ctx->dev) {
        if (hipSetDevice(D.ordinal) == hipSuccess) (void)hipDeviceSynchronize();
    }
    if (ctx->gather && !ctx->dev.empty() && hipSetDevice(ctx->dev[0].ordinal) == hipSuccess) (void)hipFree(ctx->gather);
    for (auto& D : ctx->dev) destroy_device(D);
    delete ctx;
    return VRT_OK;
}

int vrt_volume_upload(vrt_ctx* ctx, int slot, uint8_t resolution, float extent, const float* density,
                      const uint8_t* material_or_null) {
    if (!ctx || !density) return VRT_ERR_INVALID;
    return upload_volume(ctx, slot, resolution, extent, density, material_or_null, nullptr);
}

int vrt_volume_upload_voxels(vrt_ctx* ctx, int slot, uint8_t resolution, float extent, const vrt_voxel* voxels) {
    if (!ctx || !voxels) return VRT_ERR_INVALID;
    return upload_volume(ctx, slot, resolution, extent, nullptr, nullptr, voxels);
}

int vrt_set_volume_format(vrt_ctx* ctx, int format) {
    if (!ctx || (format != VRT_FORMAT_F32 && format != VRT_FORMAT_TEXEL16)) return VRT_ERR_INVALID;
    ctx->upload_format = format;
    return VRT_OK;
}

int vrt_volume_upload_texels(vrt_ctx* ctx, int slot, uint8_t resolution, float extent, const uint8_t* rgba8_texels) {
    if (!ctx || !rgba8_texels) return VRT_ERR_INVALID;
    return upload_volume(ctx, slot, resolution, extent, nullptr, nullptr, nullptr, nullptr, rgba8_texels);
}

int vrt_voxelize_mesh(vrt_ctx* ctx, int slot, uint8_t resolution, float extent, const float* positions, size_t n_vertices,
                      const uint32_t* indices, size_t n_indices, size_t* skipped_or_null) {
    if (!ctx || (!positions && n_vertices > 0) || (!indices && n_indices > 0)) return VRT_ERR_INVALID;
    if (!valid_slot(slot)) return VRT_ERR_SLOT;
    if (resolution > VRT_MAX_RESOLUTION || !(extent > 0.0f)) return VRT_ERR_INVALID;
    const int N = (1 << resolution) + 1;
    const float cell = (extent * 2.0f) / (float)(N - 1);
    MeshSource mesh;
    mesh.threshold = cell * sqrtf(3.f); /* extraction threshold, VolumeConverter.cpp:57 */
    size_t skipped = 0;
    mesh.frames.reserve(n_indices / 3);
    for (size_t i = 0; i + 3 <= n_indices; i += 3) {
        const size_t a = indices[i], b = indices[i + 1], c = indices[i + 2];
        vrt_vox::TriangleFrame t;
        if (a >= n_vertices || b >= n_vertices || c >= n_vertices ||
            !vrt_vox::make_frame(vrt_vox::v3(positions[3 * a], positions[3 * a + 1], positions[3 * a + 2]),
                                 vrt_vox::v3(positions[3 * b], positions[3 * b + 1], positions[3 * b + 2]),
                                 vrt_vox::v3(positions[3 * c], positions[3 * c + 1], positions[3 * c + 2]), t)) {
            skipped++; /* out-of-range index or degenerate triangle, like the CPU converter */
            continue;
        }
        vrt_vox::index_box(t, mesh.threshold, extent, cell, N);
        mesh.frames.push_back(t);
    }
    if (skipped_or_null) *skipped_or_null = skipped;
    int rc = upload_volume(ctx, slot, resolution, extent, nullptr, nullptr, nullptr, &mesh);
    if (rc != VRT_OK) return rc;
    /* the shell's metric (VolumeConverter sets it on the CPU volume): density*thr is a distance below thr */
    return vrt_volume_set_metric(ctx, slot, mesh.threshold, 0.5f * mesh.threshold);
}

int vrt_volume_download(vrt_ctx* ctx, int slot, vrt_voxel* out) {
    if (!ctx || !out) return VRT_ERR_INVALID;
    if (!valid_slot(slot) || !ctx->vol[slot].used) return VRT_ERR_SLOT;
    DeviceState& D = ctx->dev[0];
    const DeviceVolume& v = D.vol[slot];
    const size_t count = (size_t)ctx->vol[slot].N * ctx->vol[slot].N * ctx->vol[slot].N;
    HIP_TRY(hipSetDevice(D.ordinal));
    std::vector<float> den(count);
    std::vector<uint8_t> mat(count);
    HIP_TRY(hipMemcpy(den.data(), v.dense, count * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mat.data(), v.material, count, hipMemcpyDeviceToHost));
    memset(out, 0, count * sizeof(vrt_voxel));
    const bool t16 = ctx->vol[slot].format == VRT_FORMAT_TEXEL16; /* integer field +-q: decode like DecodeDensity */
    for (size_t i = 0; i < count; i++) {
        out[i].material = mat[i];
        out[i].density = t16 ? den[i] * 0.01f : den[i];
    }
    return VRT_OK;
}

int vrt_volume_set_material(vrt_ctx* ctx, int slot, const vrt_material* material) {
    if (!ctx || !material) return VRT_ERR_INVALID;
    if (!valid_slot(slot) || !ctx->vol[slot].used) return VRT_ERR_SLOT;
    if (memcmp(&ctx->vol[slot].mat, material, sizeof *material) == 0) return VRT_OK; /* adaptors re-send it every frame */
    ctx->vol[slot].mat = *material;
    return sync_volume_table(ctx);
}

int vrt_volume_set_metric(vrt_ctx* ctx, int slot, float density_scale, float step_max) {
    if (!ctx || !(density_scale > 0.0f)) return VRT_ERR_INVALID;
    if (!valid_slot(slot) || !ctx->vol[slot].used) return VRT_ERR_SLOT;
    if (ctx->vol[slot].density_scale == density_scale && ctx->vol[slot].step_max == step_max) return VRT_OK;
    for (auto& D : ctx->dev) { /* frames in flight still read the old table */
        HIP_TRY(hipSetDevice(D.ordinal));
        HIP_TRY(hipDeviceSynchronize());
    }
    ctx->vol[slot].density_scale = density_scale;
    ctx->vol[slot].step_max = step_max;
    ctx->scene_stale = true;
    int rc = rebuild_skip(ctx, slot);
    if (rc != VRT_OK) return rc;
    return sync_volume_table(ctx);
}

int vrt_texture_upload(vrt_ctx* ctx, int id, int width, int height, const uint8_t* rgba8) {
    if (!ctx || id < 0 || id >= VRT_MAX_TEXTURES || width < 1 || height < 1 || width > 16384 || height > 16384 || !rgba8)
        return VRT_ERR_INVALID;
    const size_t bytes = (size_t)width * height * 4;
    for (auto& D : ctx->dev) {
        HIP_TRY(hipSetDevice(D.ordinal));
        HIP_TRY(hipDeviceSynchronize()); /* frames in flight may still sample the old image */
        if (D.tex[id]) {
            HIP_TRY(hipFree(D.tex[id]));
            D.tex[id] = nullptr;
        }
        HIP_TRY(hipMalloc(&D.tex[id], bytes));
        HIP_TRY(hipMemcpy(D.tex[id], rgba8, bytes, hipMemcpyHostToDevice));
    }
    ctx->tex[id].used = true;
    ctx->tex[id].width = width;
    ctx->tex[id].height = height;
    memcpy(ctx->tex[id].first, rgba8, 4);
    return sync_volume_table(ctx);
}

int vrt_texture_free(vrt_ctx* ctx, int id) {
    if (!ctx || id < 0 || id >= VRT_MAX_TEXTURES) return VRT_ERR_INVALID;
    if (!ctx->tex[id].used) return VRT_ERR_SLOT;
    for (auto& D : ctx->dev) {
        HIP_TRY(hipSetDevice(D.ordinal));
        HIP_TRY(hipDeviceSynchronize());
        if (D.tex[id]) HIP_TRY(hipFree(D.tex[id]));
        D.tex[id] = nullptr;
    }
    ctx->tex[id] = HostTexture();
    return sync_volume_table(ctx); /* volumes that referenced it read as unbound */
}

int vrt_volume_set_textures(vrt_ctx* ctx, int slot, int albedo_id, int normal_id, int rm_id, float scale_u, float scale_v) {
    if (!ctx || scale_u == 0.0f || scale_v == 0.0f || !(scale_u == scale_u) || !(scale_v == scale_v)) return VRT_ERR_INVALID;
    if (!valid_slot(slot) || !ctx->vol[slot].used) return VRT_ERR_SLOT;
    const int ids[3] = {albedo_id, normal_id, rm_id};
    for (int i = 0; i < 3; i++) {
        if (ids[i] < -1 || ids[i] >= VRT_MAX_TEXTURES) return VRT_ERR_INVALID;
        if (ids[i] >= 0 && !ctx->tex[ids[i]].used) return VRT_ERR_SLOT;
    }
    {
        const HostVolume& cur = ctx->vol[slot];
        if (cur.tex[0] == ids[0] && cur.tex[1] == ids[1] && cur.tex[2] == ids[2] && cur.tex_scale[0] == scale_u && cur.tex_scale[1] == scale_v)
            return VRT_OK; /* unchanged: nothing to drain, nothing to send */
    }
    for (auto& D : ctx->dev) {
        HIP_TRY(hipSetDevice(D.ordinal));
        HIP_TRY(hipDeviceSynchronize());
    }
    HostVolume& h = ctx->vol[slot];
    for (int i = 0; i < 3; i++) h.tex[i] = ids[i];
    h.tex_scale[0] = scale_u;
    h.tex_scale[1] = scale_v;
    return sync_volume_table(ctx);
}

int vrt_volume_free(vrt_ctx* ctx, int slot) {
    if (!ctx) return VRT_ERR_INVALID;
    if (!valid_slot(slot) || !ctx->vol[slot].used) return VRT_ERR_SLOT;
    if (ctx->have_scene)
        for (int i = 0; i < ctx->scene.n_instances; i++)
            if (ctx->scene.instances[i].volume_slot == slot) ctx->have_scene = false; /* scene must be re-set */
    for (auto& D : ctx->dev) {
        HIP_TRY(hipSetDevice(D.ordinal));
        HIP_TRY(hipDeviceSynchronize());
        int rc = free_device_volume(D, slot);
        if (rc != VRT_OK) return rc;
    }
    ctx->vol[slot] = HostVolume();
    ctx->scene_stale = true;
    return sync_volume_table(ctx);
}

int vrt_env_upload(vrt_ctx* ctx, int face_size, const uint8_t* rgba8_faces) {
    if (!ctx || face_size < 0 || face_size > 8192) return VRT_ERR_INVALID;
    if (face_size > 0 && !rgba8_faces) return VRT_ERR_INVALID;
    const size_t bytes = (size_t)6 * face_size * face_size * 4;
    for (auto& D : ctx->dev) {
        HIP_TRY(hipSetDevice(D.ordinal));
        HIP_TRY(hipDeviceSynchronize());
        if (D.d_env && face_size != ctx->env_size) { /* same size: updated in place (captured launches keep a valid pointer) */
            HIP_TRY(hipFree(D.d_env));
            D.d_env = nullptr;
        }
        if (face_size > 0) {
            if (!D.d_env) HIP_TRY(hipMalloc(&D.d_env, bytes));
            HIP_TRY(hipMemcpy(D.d_env, rgba8_faces, bytes, hipMemcpyHostToDevice));
        }
    }
    ctx->env_size = face_size;
    return VRT_OK;
}

int vrt_scene_set(vrt_ctx* ctx, const vrt_scene* scene) {
    if (!ctx || !scene) return VRT_ERR_INVALID;
    const int bad = validate_scene(ctx, scene);
    if (bad != VRT_OK) return bad;
    /* an adaptor re-sends the scene every frame (VRDXScene::SyncWithScene runs per frame): the same scene over the same volumes
       needs no re-packing and no copies to the device */
    if (ctx->have_scene && !ctx->scene_stale && memcmp(&ctx->scene, scene, sizeof *scene) == 0) return VRT_OK;
    ctx->scene = *scene;
    ctx->have_scene = false;
    int rc = pack_scene(ctx);
    if (rc != VRT_OK) return rc;
    ctx->have_scene = true;
    ctx->scene_stale = false;
    return VRT_OK;
}

int vrt_render_rows(vrt_ctx* ctx, const vrt_params* params, int row0, int rows, void* device_rgba, void* hip_stream) {
    int rc = check_params(ctx, params);
    if (rc != VRT_OK) return rc;
    if (row0 < 0 || rows < 0 || row0 + rows > params->height || (!device_rgba && rows > 0)) return VRT_ERR_INVALID;
    DeviceState& D = ctx->dev[0];
    HIP_TRY(hipSetDevice(D.ordinal));
    const int ring = (int)(ctx->launches % kRing);
    RowSet rs;
    rs.row0 = row0;
    rs.rows = rows;
    rc = enqueue_rows(ctx, D, params, rs, static_cast<float*>(device_rgba), static_cast<hipStream_t>(hip_stream), ring);
    if (rc != VRT_OK) return rc;
    ctx->launches++;
    ctx->last_devices = 1;
    ctx->last_w = (uint32_t)params->width;
    ctx->last_h = (uint32_t)rows;
    ctx->last_gather_ms = 0.f;
    ctx->last_total_ms = 0.f;
    return VRT_OK;
}

int vrt_render_strips(vrt_ctx* ctx, const vrt_params* params, int strip_rows, int first_strip, int strip_stride, int n_strips,
                      void* device_rgba, void* hip_stream) {
    int rc = check_params(ctx, params);
    if (rc != VRT_OK) return rc;
    if (strip_rows < 1 || strip_stride < 1 || first_strip < 0 || first_strip >= strip_stride || n_strips < 0 ||
        (long long)n_strips * strip_rows > 16384 || (!device_rgba && n_strips > 0))
        return VRT_ERR_INVALID;
    DeviceState& D = ctx->dev[0];
    HIP_TRY(hipSetDevice(D.ordinal));
    const int ring = (int)(ctx->launches % kRing);
    RowSet rs;
    rs.rows = n_strips * strip_rows;
    rs.strip_rows = strip_rows;
    rs.strip_first = first_strip;
    rs.strip_stride = strip_stride;
    rc = enqueue_rows(ctx, D, params, rs, static_cast<float*>(device_rgba), static_cast<hipStream_t>(hip_stream), ring);
    if (rc != VRT_OK) return rc;
    ctx->launches++;
    ctx->last_devices = 1;
    ctx->last_w = (uint32_t)params->width;
    ctx->last_h = (uint32_t)rs.rows;
    ctx->last_gather_ms = 0.f;
    ctx->last_total_ms = 0.f;
    return VRT_OK;
}

int vrt_render_block(vrt_ctx* ctx, const vrt_params* params, const vrt_block* block, void* device_rgba, void* hip_stream) {
    int rc = check_params(ctx, params, block && block->scenes);
    if (rc != VRT_OK) return rc;
    if (!block || block->n_frames < 1 || block->n_frames > kMaxLaunchFrames || !device_rgba) return VRT_ERR_INVALID;
    if (block->scenes) { /* per-frame scene state: every frame's scene must be one vrt_scene_set would take */
        if (block->cameras || (params->flags & VRT_FLAG_DIAG_TIMELINE)) return VRT_ERR_INVALID;
        for (int f = 0; f < block->n_frames; f++) {
            rc = validate_scene(ctx, block->scenes + f);
            if (rc != VRT_OK) return rc;
        }
    }
    RowSet rs;
    if (block->strip_rows > 0) {
        if (block->strip_stride < 1 || block->first_strip < 0 || block->first_strip >= block->strip_stride || block->n_strips < 0 ||
            (long long)block->n_strips * block->strip_rows > 16384)
            return VRT_ERR_INVALID;
        rs.rows = block->n_strips * block->strip_rows;
        rs.strip_rows = block->strip_rows;
        rs.strip_first = block->first_strip;
        rs.strip_stride = block->strip_stride;
    } else {
        if (block->strip_rows < 0 || block->row0 < 0 || block->rows < 0 || block->row0 + block->rows > params->height) return VRT_ERR_INVALID;
        rs.row0 = block->row0;
        rs.rows = block->rows;
    }
    const size_t frame_bytes = (size_t)rs.rows * (size_t)params->width * ((params->flags & VRT_FLAG_OUTPUT_RGBA8) ? 4 : 16);
    const size_t pixel_bytes = (params->flags & VRT_FLAG_OUTPUT_RGBA8) ? 4 : 16;
    if (block->frame_stride_bytes < frame_bytes || block->frame_stride_bytes % pixel_bytes != 0) return VRT_ERR_INVALID;
    DeviceState& D = ctx->dev[0];
    HIP_TRY(hipSetDevice(D.ordinal));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    /* ONE launch per kMaxBlockFrames frames (grid.y = frame; only the camera differs between the frames, it travels in the
       kernarg): the dispatcher back-fills the wave slots a frame's latency-bound tail leaves empty with the next frame's waves.
       VRT_FLAG_BLOCK_PER_FRAME: one launch per frame, back to back (what this entry point did before; A/B and tests) */
    /* up to kMaxBlockFrames frames the cameras travel in the kernarg segment; a larger block is still ONE launch, its camera records
       copied to the device ahead of it on the stream (a launch's latency-bound tail is paid once per launch) — unless the stream is
       being captured into a graph: then launches of kernarg size */
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (stream != nullptr && hipStreamIsCapturing(stream, &capture) != hipSuccess) capture = hipStreamCaptureStatusNone;
    int chunk = (params->flags & VRT_FLAG_BLOCK_PER_FRAME) ? 1 : (capture != hipStreamCaptureStatusNone || block->n_frames <= kMaxBlockFrames) ? kMaxBlockFrames : kMaxLaunchFrames;
    {   /* the full closest hit in passes keeps 20 bytes per pixel of the launch's tiles between its passes: at most kPassBytesMax per launch */
        const size_t per_frame = (size_t)((params->width + 15) / 16) * (size_t)((rs.rows + 15) / 16) * 256 * (sizeof(HitRecord) + sizeof(unsigned));
        if (per_frame > 0 && closest_hit_form(ctx, params, block->scenes, block->n_frames).full)
            chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)chunk, kPassBytesMax / per_frame));
    }
    {   /* a launch's grid holds at most 8 * kMaxBlocks workgroups (enqueue_rows), and its per-wave counters — 128 bytes per workgroup and
           frame, kept per launch stream until vrt_destroy — at most kStatsBytesMax: larger blocks are cut into several launches instead of
           being refused (3840x2160 x 256 frames fitted by 0.4 %, 4096x2304 x 256 did not) */
        const long long blocks = grid_blocks((params->width + 15) / 16, (rs.rows + 15) / 16, params->flags & 3);
        if (blocks > 0) {
            chunk = (int)std::max<long long>(1, std::min<long long>(chunk, 8LL * kMaxBlocks / blocks));
            chunk = (int)std::max<long long>(1, std::min<long long>(chunk, (long long)(kStatsBytesMax / (sizeof(unsigned) * kStatRecord * 4)) / blocks));
        }
    }
    vrt_params q = *params;
    for (int f = 0; f < block->n_frames; f += chunk) {
        const int n = std::min(chunk, block->n_frames - f);
        /* per-frame launches: the block's first frame is its timing sample (an event pair costs 5-7 us of queue time) */
        q.flags = (f == 0 || chunk > 1) ? params->flags : (params->flags | VRT_FLAG_NO_TIMING);
        const int ring = (int)(ctx->launches % kRing);
        rc = enqueue_rows(ctx, D, &q, rs, reinterpret_cast<float*>(static_cast<char*>(device_rgba) + (size_t)f * block->frame_stride_bytes), stream,
                          ring, nullptr, n, block->cameras ? block->cameras + f : nullptr, (size_t)block->frame_stride_bytes,
                          block->scenes ? block->scenes + f : nullptr);
        if (rc != VRT_OK) return rc;
        ctx->launches++;
    }
    ctx->last_devices = 1;
    ctx->last_w = (uint32_t)params->width;
    ctx->last_h = (uint32_t)rs.rows;
    ctx->last_gather_ms = 0.f;
    ctx->last_total_ms = 0.f;
    return VRT_OK;
}

int vrt_render_block_host(vrt_ctx* ctx, const vrt_params* params, const vrt_block* block, const void** host_frames) {
    if (!ctx || !params || !block || !host_frames || ctx->dev.size() != 1) return VRT_ERR_INVALID;
    if (params->width <= 0 || params->height <= 0 || block->n_frames < 1 || block->n_frames > kMaxLaunchFrames) return VRT_ERR_INVALID;
    const long long rows = block->strip_rows > 0 ? (long long)block->n_strips * block->strip_rows : (long long)block->rows;
    if (block->n_strips < 0 || rows < 0 || rows > 16384) return VRT_ERR_INVALID;
    const size_t frame_bytes = (size_t)rows * (size_t)params->width * ((params->flags & VRT_FLAG_OUTPUT_RGBA8) ? 4 : 16);
    const size_t need = std::max<size_t>(frame_bytes * (size_t)block->n_frames, 16);
    if (need > ((size_t)4 << 30)) return VRT_ERR_INVALID;
    { /* refuse bad parameters BEFORE the block buffers are touched */
        const int rc = check_params(ctx, params, block->scenes != nullptr);
        if (rc != VRT_OK) return rc;
    }
    DeviceState& D = ctx->dev[0];
    HIP_TRY(hipSetDevice(D.ordinal));
    if (D.blockfb_bytes < need) {
        HIP_TRY(hipStreamSynchronize(D.stream));
        if (D.d_blockfb) HIP_TRY(hipFree(D.d_blockfb));
        if (D.h_blockfb) HIP_TRY(hipHostFree(D.h_blockfb));
        D.d_blockfb = D.h_blockfb = nullptr;
        D.blockfb_bytes = 0;
        HIP_TRY(hipMalloc(&D.d_blockfb, need));
        HIP_TRY(hipHostMalloc(&D.h_blockfb, need, hipHostMallocDefault));
        D.blockfb_bytes = need;
    }
    if (!D.copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&D.copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&D.part_done, hipEventDisableTiming));
    }
    /* In parts: the copy of one part (its own stream) runs under the march of the next.  The march is ~4x faster than the copy over
       PCIe (1080p RGBA8: 0.04 against 0.16 ms per frame), so what the parts buy is that the first frames do not wait for the last. */
    const int part = block->n_frames <= 32 ? block->n_frames : 32;
    for (int f = 0; f < block->n_frames; f += part) {
        const int n = std::min(part, block->n_frames - f);
        vrt_block b = *block;
        b.n_frames = n;
        b.frame_stride_bytes = frame_bytes;
        if (b.cameras) b.cameras += f;
        if (b.scenes) b.scenes += f;
        char* dst = static_cast<char*>(D.d_blockfb) + (size_t)f * frame_bytes;
        const int rc = vrt_render_block(ctx, params, &b, dst, D.stream);
        if (rc != VRT_OK) return rc;
        HIP_TRY(hipEventRecord(D.part_done, D.stream));
        HIP_TRY(hipStreamWaitEvent(D.copy_stream, D.part_done, 0));
        HIP_TRY(hipMemcpyAsync(static_cast<char*>(D.h_blockfb) + (size_t)f * frame_bytes, dst, frame_bytes * (size_t)n, hipMemcpyDeviceToHost, D.copy_stream));
    }
    HIP_TRY(hipStreamSynchronize(D.copy_stream));
    HIP_TRY(hipStreamSynchronize(D.stream));
    *host_frames = D.h_blockfb;
    return VRT_OK;
}

int vrt_render(vrt_ctx* ctx, const vrt_params* params, float* host_rgba_or_null) {
    int rc = check_params(ctx, params);
    if (rc != VRT_OK) return rc;
    const int n = (int)ctx->dev.size();
    const int W = params->width, H = params->height;
    const size_t row_bytes = (size_t)W * ((params->flags & VRT_FLAG_OUTPUT_RGBA8) ? 4 : 4 * sizeof(float));
    auto t0 = std::chrono::steady_clock::now();

    /* One device: the whole frame.  Several: 8-row strips (one row of waves, what bench.py's ranks use: samples max / mean over 8
       devices 1.02 against 1.17 with 32-row strips, profiles/r02_strip_balance_c3.txt) dealt round-robin (device g renders
       strips g, g+n, ... into a compact tile) — contiguous tiles would put every object row on the middle devices */
    constexpr int kStripRows = 8;
    const int total_strips = (H + kStripRows - 1) / kStripRows;
    const int strips_per = n > 1 ? (total_strips + n - 1) / n : 0;
    const int tile_rows = n > 1 ? strips_per * kStripRows : H;

    for (int g = 0; g < n; g++) {
        DeviceState& D = ctx->dev[(size_t)g];
        const size_t need = std::max<size_t>((size_t)tile_rows * row_bytes, 16);
        HIP_TRY(hipSetDevice(D.ordinal));
        if (D.fb_bytes < need) {
            if (D.fb) HIP_TRY(hipFree(D.fb));
            D.fb = nullptr;
            D.fb_bytes = 0;
            HIP_TRY(hipMalloc(&D.fb, need));
            D.fb_bytes = need;
        }
    }
    if (n > 1) {
        const size_t need = (size_t)H * row_bytes;
        HIP_TRY(hipSetDevice(ctx->dev[0].ordinal));
        if (ctx->gather_bytes < need) {
            if (ctx->gather) HIP_TRY(hipFree(ctx->gather));
            ctx->gather = nullptr;
            ctx->gather_bytes = 0;
            HIP_TRY(hipMalloc(&ctx->gather, need));
            ctx->gather_bytes = need;
        }
    }

    const int ring = (int)(ctx->launches % kRing);
    for (int g = 0; g < n; g++) {
        DeviceState& D = ctx->dev[(size_t)g];
        HIP_TRY(hipSetDevice(D.ordinal));
        RowSet rs;
        if (n > 1) {
            rs.rows = tile_rows;
            rs.strip_rows = kStripRows;
            rs.strip_first = g;
            rs.strip_stride = n;
        } else {
            rs.rows = H;
        }
        rc = enqueue_rows(ctx, D, params, rs, D.fb, D.stream, ring);
        if (rc != VRT_OK) return rc;
    }
    ctx->launches++;
    ctx->last_devices = n;
    ctx->last_w = (uint32_t)W;
    ctx->last_h = (uint32_t)H;

    float gather_ms = 0.f;
    if (n > 1) {
        /* every device copies its strips into device 0's frame over the peer links right behind its own march, in its own
           stream: ONE strided 2D copy per device (a "row" of the copy is one strip: contiguous in the compact tile, n strips
           apart in the frame) plus one plain copy when the frame ends inside the device's last strip.  No host synchronisation
           between march and gather: a device that is done early transfers while the others still march; device 0's stream then
           waits for every device's event, and the host waits for device 0 alone.  gather_ms = what the frame waits for after the
           slowest march has finished. */
        const size_t strip_bytes = (size_t)kStripRows * row_bytes;
        for (int g = 0; g < n; g++) {
            DeviceState& D = ctx->dev[(size_t)g];
            HIP_TRY(hipSetDevice(D.ordinal));
            int whole = 0; /* strips of this device that lie wholly inside the frame */
            while (whole < strips_per && ((whole * n + g) + 1) * kStripRows <= H) whole++;
            char* dst = reinterpret_cast<char*>(ctx->gather) + (size_t)g * strip_bytes;
            const char* src = reinterpret_cast<const char*>(D.fb);
            /* the strided 2D copy needs device 0's memory mapped into this device (checked at vrt_create: an asynchronous failure of an
               unmapped peer copy would only surface at the final synchronise); VRT_GATHER_PER_STRIP=1 forces the per-strip path.
               (Unverified on more than one physical GPU in any round: no multi-GPU box was available.) */
            static const bool per_strip = getenv("VRT_GATHER_PER_STRIP") != nullptr;
            const bool direct = !per_strip && (g == 0 || D.ordinal == ctx->dev[0].ordinal || D.peer_to_first);
            if (whole > 0 && (!direct || hipMemcpy2DAsync(dst, (size_t)n * strip_bytes, src, strip_bytes, strip_bytes, (size_t)whole, hipMemcpyDeviceToDevice,
                                                          D.stream) != hipSuccess)) {
                (void)hipGetLastError(); /* no direct peer mapping between the two devices: strip by strip through hipMemcpyPeerAsync */
                for (int k = 0; k < whole; k++)
                    HIP_TRY(hipMemcpyPeerAsync(dst + (size_t)k * n * strip_bytes, ctx->dev[0].ordinal, src + (size_t)k * strip_bytes, D.ordinal,
                                               strip_bytes, D.stream));
            }
            const int frame_row = (whole * n + g) * kStripRows;
            if (whole < strips_per && frame_row < H) /* the frame's last, partial strip */
                HIP_TRY(hipMemcpyPeerAsync(reinterpret_cast<char*>(ctx->gather) + (size_t)frame_row * row_bytes, ctx->dev[0].ordinal,
                                           src + (size_t)whole * strip_bytes, D.ordinal, (size_t)(H - frame_row) * row_bytes, D.stream));
            if (g > 0) {
                if (!D.joined) HIP_TRY(hipEventCreateWithFlags(&D.joined, hipEventDisableTiming));
                HIP_TRY(hipEventRecord(D.joined, D.stream));
            }
        }
        HIP_TRY(hipSetDevice(ctx->dev[0].ordinal));
        for (int g = 1; g < n; g++) HIP_TRY(hipStreamWaitEvent(ctx->dev[0].stream, ctx->dev[(size_t)g].joined, 0));
        HIP_TRY(hipStreamSynchronize(ctx->dev[0].stream));
        float march_ms = 0.f;
        for (int g = 0; g < n; g++) {
            DeviceState& D = ctx->dev[(size_t)g];
            HIP_TRY(hipSetDevice(D.ordinal));
            float ms = 0.f;
            if (D.timed[ring] && hipEventElapsedTime(&ms, D.ev0[ring], D.ev1[ring]) == hipSuccess) march_ms = std::max(march_ms, ms);
        }
        const float all_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        gather_ms = std::max(0.f, all_ms - march_ms);
        if (host_rgba_or_null) {
            HIP_TRY(hipSetDevice(ctx->dev[0].ordinal));
            HIP_TRY(hipMemcpy(host_rgba_or_null, ctx->gather, (size_t)H * row_bytes, hipMemcpyDeviceToHost));
        }
    } else {
        DeviceState& D = ctx->dev[0];
        HIP_TRY(hipSetDevice(D.ordinal));
        if (host_rgba_or_null)
            HIP_TRY(hipMemcpyAsync(host_rgba_or_null, D.fb, (size_t)H * row_bytes, hipMemcpyDeviceToHost, D.stream));
        HIP_TRY(hipStreamSynchronize(D.stream));
    }
    ctx->last_gather_ms = gather_ms;
    ctx->last_total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return VRT_OK;
}

int vrt_comm_unique_id(void* id_out) {
    if (!id_out) return VRT_ERR_INVALID;
    RcclApi* R = rccl();
    if (!R) return VRT_ERR_UNSUPPORTED;
    RcclApi::UniqueId id;
    const int rc = R->GetUniqueId(&id);
    if (rc != 0) {
        fprintf(stderr, "[vrt] ncclGetUniqueId failed: %s\n", R->GetErrorString(rc));
        return VRT_ERR_HIP;
    }
    memcpy(id_out, id.internal, VRT_COMM_ID_BYTES);
    return VRT_OK;
}

int vrt_comm_init(vrt_ctx* ctx, int world, int rank, const void* id) {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world || ctx->comm) return VRT_ERR_INVALID;
    RcclApi* R = rccl();
    if (!R) return VRT_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->dev[0].ordinal));
    RcclApi::UniqueId uid;
    memcpy(uid.internal, id, VRT_COMM_ID_BYTES);
    void* comm = nullptr;
    const int rc = R->CommInitRank(&comm, world, uid, rank);
    if (rc != 0) {
        fprintf(stderr, "[vrt] ncclCommInitRank failed: %s\n", R->GetErrorString(rc));
        return VRT_ERR_HIP;
    }
    ctx->comm = comm;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    ctx->expect_tile_bytes = ctx->expect_chunk_bytes = 0;
    ctx->sizes_agreed = false;
    return VRT_OK;
}

/* A collective whose ranks disagree about its size does not fail inside RCCL: it waits for ever.  This call (collective, synchronous, a
   setup call) makes the sizes part of the communicator: every rank sends its pair to every rank in one group of 16-byte messages — whose
   size cannot disagree — and compares; afterwards vrt_gather_tiles / vrt_exchange_tiles refuse any other size with VRT_ERR_INVALID before
   they touch RCCL.  0 = that collective is not going to be used (then a call to it is refused).  Call again to change the sizes. */
int vrt_comm_expect_sizes(vrt_ctx* ctx, size_t gather_tile_bytes, size_t exchange_chunk_bytes) {
    if (!ctx || !ctx->comm) return VRT_ERR_NOT_READY;
    RcclApi* R = rccl();
    if (!R) return VRT_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->dev[0].ordinal));
    const int W = ctx->comm_world;
    struct Pair { unsigned long long tile, chunk; };
    std::vector<Pair> mine((size_t)W, Pair{(unsigned long long)gather_tile_bytes, (unsigned long long)exchange_chunk_bytes}), theirs((size_t)W);
    Pair* d_send = nullptr;
    Pair* d_recv = nullptr;
    hipStream_t st = nullptr;
    int status = VRT_OK;
    do {
        if (hipMalloc(&d_send, sizeof(Pair) * (size_t)W) != hipSuccess || hipMalloc(&d_recv, sizeof(Pair) * (size_t)W) != hipSuccess) { status = VRT_ERR_OOM; break; }
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
            hipMemcpyAsync(d_send, mine.data(), sizeof(Pair) * (size_t)W, hipMemcpyHostToDevice, st) != hipSuccess) { status = VRT_ERR_HIP; break; }
        int rc = R->GroupStart();
        for (int peer = 0; rc == 0 && peer < W; peer++) {
            rc = R->Send(d_send + peer, sizeof(Pair), kNcclUint8, peer, ctx->comm, st);
            if (rc == 0) rc = R->Recv(d_recv + peer, sizeof(Pair), kNcclUint8, peer, ctx->comm, st);
        }
        const int rc_end = R->GroupEnd();
        if (rc != 0 || rc_end != 0) { status = VRT_ERR_HIP; break; }
        if (hipMemcpyAsync(theirs.data(), d_recv, sizeof(Pair) * (size_t)W, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { status = VRT_ERR_HIP; break; }
        for (int peer = 0; peer < W; peer++)
            if (theirs[(size_t)peer].tile != mine[0].tile || theirs[(size_t)peer].chunk != mine[0].chunk) {
                fprintf(stderr, "[vrt] vrt_comm_expect_sizes: rank %d expects tile %llu / chunk %llu bytes, rank %d expects %llu / %llu\n", ctx->comm_rank,
                        mine[0].tile, mine[0].chunk, peer, theirs[(size_t)peer].tile, theirs[(size_t)peer].chunk);
                status = VRT_ERR_INVALID; /* every rank sees the same disagreement and returns the same status */
            }
    } while (false);
    if (st) (void)hipStreamDestroy(st);
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    ctx->expect_tile_bytes = status == VRT_OK ? gather_tile_bytes : 0;
    ctx->expect_chunk_bytes = status == VRT_OK ? exchange_chunk_bytes : 0;
    ctx->sizes_agreed = status == VRT_OK;
    return status;
}

int vrt_comm_destroy(vrt_ctx* ctx) {
    if (!ctx) return VRT_ERR_INVALID;
    if (!ctx->comm) return VRT_OK;
    RcclApi* R = rccl();
    if (R) {
        (void)hipSetDevice(ctx->dev[0].ordinal);
        (void)hipDeviceSynchronize();
        (void)R->CommDestroy(ctx->comm);
    }
    ctx->comm = nullptr;
    return VRT_OK;
}

int vrt_gather_tiles(vrt_ctx* ctx, const void* device_tile, void* device_frame_or_null, size_t tile_bytes, int root, void* hip_stream) {
    if (!ctx || !ctx->comm) return VRT_ERR_NOT_READY;
    if (root < 0 || root >= ctx->comm_world || (!device_tile && tile_bytes > 0) ||
        (ctx->comm_rank == root && !device_frame_or_null && tile_bytes > 0))
        return VRT_ERR_INVALID;
    if (ctx->sizes_agreed && tile_bytes != ctx->expect_tile_bytes) return VRT_ERR_INVALID; /* not the size the ranks agreed on: it would wait for ever */
    RcclApi* R = rccl();
    if (!R) return VRT_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->dev[0].ordinal));
    const int rc = R->Gather(device_tile, device_frame_or_null, tile_bytes, kNcclUint8, root, ctx->comm, static_cast<hipStream_t>(hip_stream));
    if (rc != 0) {
        fprintf(stderr, "[vrt] ncclGather failed: %s\n", R->GetErrorString(rc));
        return VRT_ERR_HIP;
    }
    return VRT_OK;
}

int vrt_exchange_tiles(vrt_ctx* ctx, const void* device_tiles, void* device_recv, size_t chunk_bytes, void* hip_stream) {
    if (!ctx || !ctx->comm) return VRT_ERR_NOT_READY;
    if ((!device_tiles || !device_recv) && chunk_bytes > 0) return VRT_ERR_INVALID;
    if (ctx->sizes_agreed && chunk_bytes != ctx->expect_chunk_bytes) return VRT_ERR_INVALID; /* not the size the ranks agreed on */
    RcclApi* R = rccl();
    if (!R) return VRT_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->dev[0].ordinal));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    /* one group: every rank's sends and receives progress together (rccl.h:690-691); chunk d of my buffer goes to rank d, chunk
       s of the receive buffer comes from rank s (the own chunk is a device copy inside RCCL) */
    int rc = R->GroupStart();
    for (int peer = 0; rc == 0 && peer < ctx->comm_world; peer++) {
        rc = R->Send(static_cast<const char*>(device_tiles) + (size_t)peer * chunk_bytes, chunk_bytes, kNcclUint8, peer, ctx->comm, stream);
        if (rc == 0) rc = R->Recv(static_cast<char*>(device_recv) + (size_t)peer * chunk_bytes, chunk_bytes, kNcclUint8, peer, ctx->comm, stream);
    }
    const int rc_end = R->GroupEnd();
    if (rc == 0) rc = rc_end;
    if (rc != 0) {
        fprintf(stderr, "[vrt] ncclSend/ncclRecv group failed: %s\n", R->GetErrorString(rc));
        return VRT_ERR_HIP;
    }
    return VRT_OK;
}

int vrt_render_begin(vrt_ctx* ctx, const vrt_params* params, int slot) {
    int rc = check_params(ctx, params);
    if (rc != VRT_OK) return rc;
    if (slot < 0 || slot >= VRT_FRAMES_IN_FLIGHT || ctx->dev.size() != 1) return VRT_ERR_INVALID;
    DeviceState& D = ctx->dev[0];
    FrameSlot& S = D.slot[slot];
    if (S.busy) return VRT_ERR_NOT_READY; /* vrt_render_end(slot) first */
    HIP_TRY(hipSetDevice(D.ordinal));
    const size_t need = (size_t)params->width * params->height * ((params->flags & VRT_FLAG_OUTPUT_RGBA8) ? 4 : 16);
    if (!D.copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&D.copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&D.part_done, hipEventDisableTiming));
    }
    if (!D.flight_stream) HIP_TRY(hipStreamCreateWithFlags(&D.flight_stream, hipStreamNonBlocking));
    if (!S.done) {
        /* ONE march stream and ONE copy stream for all the frame slots (a slot owns its buffers and two events): the marches run back
           to back, frame k's read-back runs under frame k+1's march whatever the number of slots and however HIP maps streams to
           hardware queues.  (A stream per slot — round 3 — gave 4 300 frames/s with three slots and 5 500 with two on the demo
           scene: with three, a slot's stream shares a hardware queue with another's.) */
        S.stream = D.flight_stream;
        HIP_TRY(hipEventCreateWithFlags(&S.marched, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
        HIP_TRY(hipMalloc(&S.d_scene, sizeof(SceneArrays)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S.h_scene), sizeof(SceneArrays), hipHostMallocDefault));
    }
    if (S.fb_bytes < need) {
        if (S.d_fb) HIP_TRY(hipFree(S.d_fb));
        if (S.h_fb) HIP_TRY(hipHostFree(S.h_fb));
        S.d_fb = S.h_fb = nullptr;
        S.fb_bytes = 0;
        HIP_TRY(hipMalloc(&S.d_fb, need));
        HIP_TRY(hipHostMalloc(&S.h_fb, need, hipHostMallocDefault));
        S.fb_bytes = need;
    }
    /* snapshot of everything the kernel dereferences besides the volumes themselves: later vrt_scene_set /
       vrt_volume_set_* calls do not reach a frame that is already in flight */
    for (int i = 0; i < VRT_MAX_VOLUMES; i++) fill_dvolume(ctx, D, ctx->vol[i], D.vol[i], S.h_scene->vols[i]);
    memcpy(S.h_scene->inst, ctx->inst, sizeof S.h_scene->inst);
    memcpy(S.h_scene->nodes, ctx->nodes, sizeof S.h_scene->nodes);
    memcpy(S.h_scene->point, ctx->point, sizeof S.h_scene->point);
    memcpy(S.h_scene->spot, ctx->spot, sizeof S.h_scene->spot);
    HIP_TRY(hipMemcpyAsync(S.d_scene, S.h_scene, sizeof(SceneArrays), hipMemcpyHostToDevice, S.stream));
    const int ring = (int)(ctx->launches % kRing);
    RowSet rs;
    rs.rows = params->height;
    rc = enqueue_rows(ctx, D, params, rs, static_cast<float*>(S.d_fb), S.stream, ring, S.d_scene);
    if (rc != VRT_OK) return rc;
    HIP_TRY(hipEventRecord(S.marched, S.stream));
    HIP_TRY(hipStreamWaitEvent(D.copy_stream, S.marched, 0));
    HIP_TRY(hipMemcpyAsync(S.h_fb, S.d_fb, need, hipMemcpyDeviceToHost, D.copy_stream));
    HIP_TRY(hipEventRecord(S.done, D.copy_stream));
    S.busy = true;
    S.ring = ring;
    ctx->launches++;
    ctx->last_devices = 1;
    ctx->last_w = (uint32_t)params->width;
    ctx->last_h = (uint32_t)params->height;
    ctx->last_gather_ms = 0.f;
    ctx->last_total_ms = 0.f;
    return VRT_OK;
}

int vrt_render_end(vrt_ctx* ctx, int slot, const void** host_pixels) {
    if (!ctx || slot < 0 || slot >= VRT_FRAMES_IN_FLIGHT || ctx->dev.size() != 1) return VRT_ERR_INVALID;
    DeviceState& D = ctx->dev[0];
    FrameSlot& S = D.slot[slot];
    if (!S.busy) return VRT_ERR_NOT_READY;
    HIP_TRY(hipSetDevice(D.ordinal));
    HIP_TRY(hipEventSynchronize(S.done));
    S.busy = false;
    if (host_pixels) *host_pixels = S.h_fb;
    return VRT_OK;
}

int vrt_last_timing(vrt_ctx* ctx, vrt_timing* out) {
    if (!ctx || !out) return VRT_ERR_INVALID;
    if (ctx->launches == 0) return VRT_ERR_NOT_READY;
    memset(out, 0, sizeof *out);
    const int ring = (int)((ctx->launches - 1) % kRing);
    unsigned long long tot[kStatWords] = {0, 0, 0, 0, 0, 0, 0};
    float kernel_ms = 0.f;
    for (int g = 0; g < ctx->last_devices; g++) {
        DeviceState& D = ctx->dev[(size_t)g];
        HIP_TRY(hipSetDevice(D.ordinal));
        float ms = 0.f;
        if (D.timed[ring]) {
            HIP_TRY(hipEventSynchronize(D.ev1[ring]));
            HIP_TRY(hipEventElapsedTime(&ms, D.ev0[ring], D.ev1[ring]));
        } else {
            HIP_TRY(hipDeviceSynchronize()); /* captured launch: wait for whatever replay is in flight before reading the counters */
        }
        kernel_ms = std::max(kernel_ms, ms);
        std::vector<unsigned> rec((size_t)D.last_blocks * 4 * kStatRecord);
        if (D.last_blocks > 0)
            HIP_TRY(hipMemcpy(rec.data(), D.d_stats[D.last_slot] + (size_t)(D.last_frames - 1) * rec.size(), rec.size() * sizeof(unsigned),
                              hipMemcpyDeviceToHost));
        for (size_t w = 0; w < (size_t)D.last_blocks * 4; w++)
            for (int k = 0; k < kStatWords; k++) tot[k] += rec[w * kStatRecord + k];
    }
    out->kernel_ms = kernel_ms;
    out->gather_ms = ctx->last_gather_ms;
    out->total_ms = ctx->last_total_ms;
    out->width = ctx->last_w;
    out->height = ctx->last_h;
    out->primary_rays = tot[0];
    out->shadow_rays = tot[1];
    out->bounce_rays = tot[2];
    out->primary_steps = tot[3];
    out->shadow_steps = tot[4];
    out->hits = tot[5];
    out->exhausted_rays = tot[6];
    return VRT_OK;
}

int vrt_timing_history(vrt_ctx* ctx, int n, float* kernel_ms_out) {
    if (!ctx || n < 0 || (n > 0 && !kernel_ms_out)) return VRT_ERR_INVALID;
    const uint64_t have = std::min<uint64_t>(ctx->launches, (uint64_t)kRing);
    const int m = (int)std::min<uint64_t>(have, (uint64_t)n);
    DeviceState& D = ctx->dev[0];
    HIP_TRY(hipSetDevice(D.ordinal));
    for (int i = 0; i < m; i++) {
        const uint64_t launch = ctx->launches - (uint64_t)m + (uint64_t)i;
        const int ring = (int)(launch % kRing);
        float ms = 0.f;
        if (D.timed[ring]) {
            HIP_TRY(hipEventSynchronize(D.ev1[ring]));
            HIP_TRY(hipEventElapsedTime(&ms, D.ev0[ring], D.ev1[ring]));
        }
        kernel_ms_out[i] = ms;
    }
    return m;
}

int vrt_launch_history(vrt_ctx* ctx, int n, float* kernel_ms_out, int* frames_out) {
    if (!ctx || n < 0 || (n > 0 && (!kernel_ms_out || !frames_out))) return VRT_ERR_INVALID;
    const int m = vrt_timing_history(ctx, n, kernel_ms_out);
    if (m < 0) return m;
    const DeviceState& D = ctx->dev[0];
    for (int i = 0; i < m; i++) frames_out[i] = D.ring_frames[(int)((ctx->launches - (uint64_t)m + (uint64_t)i) % kRing)];
    return m;
}

int vrt_debug_gather_ceiling(vrt_ctx* ctx, int format, int coherent_lanes, unsigned n_bricks, float* gsamples_per_s_out) {
    if (!ctx || ctx->dev.empty() || !gsamples_per_s_out || (format != VRT_FORMAT_F32 && format != VRT_FORMAT_TEXEL16)) return VRT_ERR_INVALID;
    if (n_bricks < 1 || n_bricks > (1u << 22) || (n_bricks & (n_bricks - 1)) != 0) return VRT_ERR_INVALID;
    DeviceState& D = ctx->dev[0];
    HIP_TRY(hipSetDevice(D.ordinal));
    const int blocks = 256 * 8 * 4, iters = 256; /* 8192 workgroups of 4 waves: every CU at its occupancy limit */
    const size_t pool_bytes = (size_t)n_bricks * (format == VRT_FORMAT_TEXEL16 ? 256 : 512);
    void* pool = nullptr;
    float* out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = VRT_OK;
    float best = 1e30f;
    do {
        if (hipMalloc(&pool, pool_bytes) != hipSuccess || hipMalloc(&out, sizeof(float) * (size_t)blocks * 256) != hipSuccess) { rc = VRT_ERR_OOM; break; }
        if (hipMemset(pool, 0, pool_bytes) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = VRT_ERR_HIP; break; }
        for (int rep = 0; rep < 5 && rc == VRT_OK; rep++) { /* the first repetition warms the caches and the clocks */
            float ms = 0.f;
            if (hipEventRecord(e0, nullptr) != hipSuccess || launch_gather_ceiling(pool, n_bricks, format, coherent_lanes != 0, iters, out, blocks, nullptr) != hipSuccess ||
                hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
                rc = VRT_ERR_HIP;
            else if (rep > 0 && ms < best) best = ms;
        }
    } while (false);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (pool) (void)hipFree(pool);
    if (out) (void)hipFree(out);
    if (rc != VRT_OK) return rc;
    *gsamples_per_s_out = (float)((double)blocks * 256.0 * iters / ((double)best * 1e6));
    return VRT_OK;
}

int vrt_debug_last_kernel_form(vrt_ctx* ctx) {
    if (!ctx || ctx->dev.empty()) return VRT_ERR_INVALID;
    return ctx->dev[0].last_form;
}

long long vrt_debug_wave_records(vrt_ctx* ctx, int which, uint32_t* out, long long max_words) {
    if (!ctx || max_words < 0 || (max_words > 0 && !out) || which < 0 || which > 3) return VRT_ERR_INVALID;
    if (ctx->launches == 0) return VRT_ERR_NOT_READY;
    DeviceState& D = ctx->dev[0];
    const bool diag = (which & 1) != 0, whole = which >= 2;
    if (diag && (!D.last_diag || !D.d_diag)) return VRT_ERR_NOT_READY;
    const long long frame_words = (long long)D.last_blocks * 4 * (diag ? kDiagRecord : kStatRecord);
    const long long words = whole ? frame_words * D.last_frames : frame_words;
    if (out && max_words > 0) {
        HIP_TRY(hipSetDevice(D.ordinal));
        HIP_TRY(hipDeviceSynchronize());
        const uint32_t* src = (diag ? D.d_diag : D.d_stats[D.last_slot]) + (whole ? 0 : (size_t)(D.last_frames - 1) * (size_t)frame_words);
        HIP_TRY(hipMemcpy(out, src, sizeof(uint32_t) * (size_t)std::min(words, max_words), hipMemcpyDeviceToHost));
    }
    return words;
}

const char* vrt_strerror(int status) {
    switch (status) {
        case VRT_OK: return "ok";
        case VRT_ERR_INVALID: return "invalid argument";
        case VRT_ERR_NO_DEVICE: return "no usable HIP device";
        case VRT_ERR_HIP: return "HIP runtime call failed";
        case VRT_ERR_OOM: return "out of device memory";
        case VRT_ERR_SLOT: return "volume slot out of range or empty";
        case VRT_ERR_NOT_READY: return "scene or volume not set";
        case VRT_ERR_UNSUPPORTED: return "render mode / feature not implemented";
        default: return "unknown status";
    }
}

const char* vrt_version(void) { return "0.2.0 gfx950"; }

}  // extern "C"

/*
 * vrt_oracle.cpp — scalar CPU restatement of the ray-march hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see vrt_oracle.h).  Plain C++17, no dependencies, fp32 with
 * contraction disabled at compile time (-ffp-contract=off) and FMA written explicitly as
 * fmaf() where the march contract says so.
 *
 * PARITY UNPINNED by reference tests (the reference has none); pinned by
 * analytic ground truth, by vrto_ref_hit_t and — frames, not just hit distances — by vrto_ref_render below: a second,
 * independent restatement of the reference's OWN intersection (cell walk + exact cubic root + normal at the root) whose
 * frames the sphere-trace's frames must equal in the reference's 8-bit colours (tests/test_reference_pixels.py).
 *
 * Reference lines each function follows (paths relative to
 * /root/reference/VolumetricRaytracer/VolumetricRaytracer/, SH = Renderer/DX/Resources/Shaders):
 *   quat_to_mat        Core/Private/Quat.cpp:28-33,91-95 (Eigen Quaternionf rotate, closed form)
 *   camera_basis       Renderer/DX/Private/RDXScene.cpp:703-724 (XMMatrixLookToRH / PerspectiveFovRH)
 *   camera_ray         SH/Include/Ray.hlsli:36-48 (+ normalisation, DESIGN.md §3 deviation)
 *   build_instance     Renderer/DX/Private/RDXLevelObject.cpp:38-47 (rotation*scale*translation)
 *   slab               SH/Include/Ray.hlsli:111-134
 *   voxel indexing     Core/Private/MathHelpers (2).cpp:43-46, Voxel/Private/VoxelVolume.cpp:19-27,139-146
 *   trilinear          SH/Include/Voxel.hlsli:607-684 (same interpolant, nested-lerp evaluation)
 *   normal             SH/Include/Voxel.hlsli:783-804; entry-face normal SH/Raytracing.hlsl:198-226
 *   march              replaces SH/Raytracing.hlsl:147-336 (sphere-trace per BASELINE.json north_star)
 *   shade              SH/Raytracing_NoTex.hlsl:41-139, SH/Include/Lighting.hlsli:17-101, Constants.hlsli:15-17
 *   env_lookup         SH/Raytracing.hlsl:444-449 (SampleLevel(dir.xzy), point sampler RDXScene.cpp:188-197)
 *   tone-map           SH/Raytracing.hlsl:34-38
 */
#include "vrt_oracle.h"

#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdint>

namespace {

struct V3 {
    float x, y, z;
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline V3 normalize(V3 a) {
    float inv = 1.0f / sqrtf(dot(a, a));
    return a * inv;
}
inline float maxf(float a, float b) { return a > b ? a : b; }
inline float minf(float a, float b) { return a < b ? a : b; }

struct M3 {
    float m[3][3];
};
inline V3 mul(const M3& M, V3 v) {
    return v3((M.m[0][0] * v.x + M.m[0][1] * v.y) + M.m[0][2] * v.z,
              (M.m[1][0] * v.x + M.m[1][1] * v.y) + M.m[1][2] * v.z,
              (M.m[2][0] * v.x + M.m[2][1] * v.y) + M.m[2][2] * v.z);
}

/* Rotation matrix of a unit quaternion (x,y,z,w), column-vector convention v' = R v. */
M3 quat_to_mat(const float q[4]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float xx = x * x, yy = y * y, zz = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    M3 R;
    R.m[0][0] = 1.0f - 2.0f * (yy + zz);
    R.m[0][1] = 2.0f * (xy - wz);
    R.m[0][2] = 2.0f * (xz + wy);
    R.m[1][0] = 2.0f * (xy + wz);
    R.m[1][1] = 1.0f - 2.0f * (xx + zz);
    R.m[1][2] = 2.0f * (yz - wx);
    R.m[2][0] = 2.0f * (xz - wy);
    R.m[2][1] = 2.0f * (yz + wx);
    R.m[2][2] = 1.0f - 2.0f * (xx + yy);
    return R;
}

struct Camera {
    V3 origin;
    V3 r0, r1, r2; /* view-space x, y, z axes in world space (LookToRH rows) */
    float cx, cy;  /* aspect*tan(fov/2), tan(fov/2) */
};

Camera camera_basis(const vrt_scene* s, int width, int height) {
    Camera c;
    M3 R = quat_to_mat(s->cam_rotation);
    V3 fwd = v3(R.m[0][0], R.m[1][0], R.m[2][0]); /* R * (+X) */
    V3 up = v3(R.m[0][2], R.m[1][2], R.m[2][2]);  /* R * (+Z) */
    c.origin = v3(s->cam_position[0], s->cam_position[1], s->cam_position[2]);
    c.r2 = normalize(v3(-fwd.x, -fwd.y, -fwd.z));
    c.r0 = normalize(cross(up, c.r2));
    c.r1 = cross(c.r2, c.r0);
    float aspect = (float)width / (float)height;
    float half = tanf(s->cam_fov_deg * (3.14159265358979323846f / 180.0f) * 0.5f);
    c.cx = aspect * half;
    c.cy = half;
    return c;
}

inline void camera_ray(const Camera& c, int width, int height, int px, int py, V3& o, V3& d, float* len_or_null = nullptr) {
    /* pixel centre -> NDC with the reciprocal of the frame size (one rounding more than a division; the kernel takes the
       reciprocals from the host: two exact divisions per pixel were 8 % of a sky pixel's instructions) */
    const float inv_w = 1.0f / (float)width, inv_h = 1.0f / (float)height;
    float sx = (((float)px + 0.5f) * inv_w) * 2.0f - 1.0f;
    float sy = (((float)py + 0.5f) * inv_h) * 2.0f - 1.0f;
    float tx = sx * c.cx;
    float ty = (-sy) * c.cy;
    V3 dir = v3((tx * c.r0.x + ty * c.r1.x) - c.r2.x,
                (tx * c.r0.y + ty * c.r1.y) - c.r2.y,
                (tx * c.r0.z + ty * c.r1.z) - c.r2.z);
    o = c.origin;
    d = normalize(dir);
    /* |direction| before the normalisation — what the reference's un-normalised WorldRayDirection() is long (VRT_FLAG_REFERENCE_VIEW_VECTOR) */
    if (len_or_null) *len_or_null = sqrtf(dot(dir, dir));
}

/* GenerateCameraRay as the shader leaves it (Ray.hlsli:36-48): target = projInv * (x, -y, 1, 1), direction = viewInv * (target.xyz, 0),
   NOT normalised: |direction| = 1 in the frame's centre, 1.55 in the corner of a 16:9 frame at 60 degrees.  Every t of the reference —
   its 0.01 / 0.1 nudges, the 0.1 back-off of the secondary rays, TMax — is in units of that length, and the closest-hit shader's
   wo = -WorldRayDirection() is not a unit vector either.  (vrto_ref_literal_render only.) */
inline void camera_ray_raw(const Camera& c, int width, int height, int px, int py, V3& o, V3& d) {
    const float sx = (((float)px + 0.5f) / (float)width) * 2.0f - 1.0f;
    const float sy = (((float)py + 0.5f) / (float)height) * 2.0f - 1.0f;
    const float tx = sx * c.cx;
    const float ty = (-sy) * c.cy;
    d = v3((tx * c.r0.x + ty * c.r1.x) - c.r2.x, (tx * c.r0.y + ty * c.r1.y) - c.r2.y, (tx * c.r0.z + ty * c.r1.z) - c.r2.z);
    o = c.origin;
}

struct LitVolume; /* vrt_ref_literal.inl: what the reference's shaders read of a volume (decoded texels, collapsed octree) */

struct Volume {
    const float* den;
    int N;
    float extent, cell, inv_cell, density_scale, step_max;
    float tint[3], roughness, metallic, k;
    float roughness_raw, metallic_raw; /* unclamped material values: the textured modes clamp after the RM texture */
    vrto_texture tex[3];               /* albedo, normal, rm */
    float tex_scale[2];
    bool textured;                     /* any of the three bound */
    /* Two-level empty-space table of volumes with a bounded step (step_max finite); see build_skip_table /
       build_nibble_table.  skip: per 4^3-cell brick the Chebyshev distance D, in bricks, to the nearest brick that
       holds an ACTIVE cell (a cell with a corner closer than step_max to the surface), capped at 255.  nib: per
       brick eight 4-bit fields, one per 2^3-cell sub-block (bit offset 4*((lx>>1)*4 + (lz>>1)*2 + (ly>>1))): the
       floor of the Euclidean distance, in cells, from the sub-block's cells to the nearest active cell, capped at
       15 — the leap of a position in that sub-block.  skip only bounds the active box (and says that the volume has tables).
       Null: no leaping. */
    int nb = 0;
    const uint8_t* skip = nullptr;
    const uint32_t* nib = nullptr;
    /* Cube modes: per brick the Chebyshev distance, in bricks, to the nearest brick that holds a solid
       voxel (density <= 0 at one of its 4^3 cell-origin voxels); 0 = this brick holds one.  Built on demand. */
    const uint8_t* cube_skip = nullptr;
    /* Active box (with skip): object-space bounding box of the near bricks; the sphere-trace is clipped to it. */
    float alo[3] = {0, 0, 0}, ahi[3] = {0, 0, 0};
    std::shared_ptr<const struct Derived> derived; /* owner of the three tables and of the quantised field */
    const LitVolume* lit = nullptr;                /* vrto_ref_literal_render only */
};

/* Everything derived from a volume's samples and metric, cached across vrto_* calls (keyed by a hash of the
   samples, never by their address): building the tables of a 256^3 volume takes seconds on one core. */
struct Derived {
    uint64_t key_hash = 0;
    int N = 0, format = 0;
    float density_scale = 0.f, step_max = 0.f;
    std::vector<float> field;      /* VRT_FORMAT_TEXEL16: the integer-valued field the march sees */
    std::vector<uint8_t> skip;
    std::vector<uint32_t> nib;
    std::vector<uint8_t> cube_skip;
    int abox[6] = {0, 0, 0, -1, -1, -1}; /* bounding box of the near bricks {min x, z, y, max x, z, y} */
    bool has_cube = false;
    std::shared_ptr<LitVolume> lit; /* vrto_ref_literal_render only, built on demand */
};
void ensure_literal(Derived& d, const float* den, int N, int resolution);

struct Instance {
    int slot;
    M3 o2w;  /* S * R   */
    M3 w2o;  /* R^T * S^-1 */
    V3 pos;
};

struct Packed {
    Camera cam;
    Volume vol[VRT_MAX_VOLUMES];
    Instance inst[VRT_MAX_INSTANCES];
    int n_inst;
    const uint8_t* env;
    int env_size;
    V3 light_dir;
    float light_strength;
    const vrt_scene* scene;
    vrt_params prm;
    bool ref_intersection = false; /* vrto_ref_render: every ray is intersected by the REFERENCE's hit search (ref_march_instance) */
    bool literal = false;          /* vrto_ref_literal_render: ... by the literal restatement of its shaders (vrt_ref_literal.inl) */
    unsigned lit_options = 0;
    uint32_t* lead_img = nullptr; /* optional debug output, rows*width: positions the primary ray skipped before its first sample */
    uint32_t* steps_img; /* optional debug output, rows*width: march positions of the primary ray (low 16 bits) and of the rays after it (high 16) */
};

Instance build_instance(const vrt_instance& in) {
    Instance r;
    M3 R = quat_to_mat(in.rotation);
    for (int i = 0; i < 3; i++) {
        float inv_sj[3] = {1.0f / in.scale[0], 1.0f / in.scale[1], 1.0f / in.scale[2]};
        for (int j = 0; j < 3; j++) {
            r.o2w.m[i][j] = in.scale[i] * R.m[i][j];
            r.w2o.m[i][j] = R.m[j][i] * inv_sj[j];
        }
    }
    r.pos = v3(in.position[0], in.position[1], in.position[2]);
    r.slot = in.volume_slot;
    return r;
}

/* In place: entries 0 are seeds, 255 unknown; afterwards D = k for the bricks first covered by the k-th
   3x3x3 dilation of the seed set (exact Chebyshev distance in bricks, capped at 255). */
void chebyshev_dilate(std::vector<uint8_t>& cur, int nb) {
    std::vector<uint8_t> nxt(cur.size());
    for (int k = 1; k < 255; k++) {
        bool changed = false;
        for (int bx = 0; bx < nb; bx++)
            for (int bz = 0; bz < nb; bz++)
                for (int by = 0; by < nb; by++) {
                    const size_t i = ((size_t)bx * nb + bz) * nb + by;
                    uint8_t d = cur[i];
                    if (d == 255) {
                        bool hit = false;
                        for (int dx = -1; dx <= 1 && !hit; dx++)
                            for (int dz = -1; dz <= 1 && !hit; dz++)
                                for (int dy = -1; dy <= 1 && !hit; dy++) {
                                    int x = bx + dx, y = by + dy, z = bz + dz;
                                    if (x < 0 || y < 0 || z < 0 || x >= nb || y >= nb || z >= nb) continue;
                                    hit = cur[((size_t)x * nb + z) * nb + y] == (uint8_t)(k - 1);
                                }
                        if (hit) {
                            d = (uint8_t)k;
                            changed = true;
                        }
                    }
                    nxt[i] = d;
                }
        cur.swap(nxt);
        if (!changed) break;
    }
}

int table_threads() {
    unsigned n = std::thread::hardware_concurrency();
    return (int)(n < 1 ? 1 : (n > 32 ? 32 : n));
}

/* Runs fn(x0, x1) over [0, n) split into contiguous chunks on table_threads() threads. */
template <class F>
void parallel_slabs(int n, F fn) {
    const int T = std::min(table_threads(), n < 1 ? 1 : n);
    if (T <= 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (int k = 0; k < T; k++) th.emplace_back([=]() { fn((int)((long long)n * k / T), (int)((long long)n * (k + 1) / T)); });
    for (auto& t : th) t.join();
}

/*
 * Empty-space table, level 1 (restates, for the sphere-trace, what the reference's collapsed octree did for
 * its DDA: Voxel/Private/Octree.cpp:70-107,181-262 merges cells without surface).  A cell is ACTIVE when one of
 * its 8 corners satisfies density*density_scale < step_max, i.e. holds a trustworthy distance below the clamp;
 * the interpolant is below the clamp only inside active cells.  A brick is "near" when it holds an active cell
 * (equivalently: when any of its 5^3 samples is below the clamp).  D[b] = Chebyshev distance in bricks from b
 * to the nearest near brick.  From any point of a brick with D >= 2 the ray may advance (D-1) brick edges: that
 * cannot reach a near brick.
 */
void build_skip_table(const float* den, int N, int nb, float density_scale, float step_max, std::vector<uint8_t>& out, int abox[6]) {
    std::vector<uint8_t> cur((size_t)nb * nb * nb, 255);
    parallel_slabs(nb, [&](int b0, int b1) {
        for (int bx = b0; bx < b1; bx++)
            for (int bz = 0; bz < nb; bz++)
                for (int by = 0; by < nb; by++) {
                    bool near = false;
                    for (int lx = 0; lx < 5 && !near; lx++)
                        for (int lz = 0; lz < 5 && !near; lz++)
                            for (int ly = 0; ly < 5 && !near; ly++) {
                                int x = bx * 4 + lx, y = by * 4 + ly, z = bz * 4 + lz;
                                x = x > N - 1 ? N - 1 : x;
                                y = y > N - 1 ? N - 1 : y;
                                z = z > N - 1 ? N - 1 : z;
                                near = den[((size_t)x * N + z) * N + y] * density_scale < step_max;
                            }
                    if (near) cur[((size_t)bx * nb + bz) * nb + by] = 0;
                }
    });
    abox[0] = abox[1] = abox[2] = nb;
    abox[3] = abox[4] = abox[5] = -1;
    for (int bx = 0; bx < nb; bx++)
        for (int bz = 0; bz < nb; bz++)
            for (int by = 0; by < nb; by++)
                if (cur[((size_t)bx * nb + bz) * nb + by] == 0) {
                    abox[0] = bx < abox[0] ? bx : abox[0];
                    abox[1] = bz < abox[1] ? bz : abox[1];
                    abox[2] = by < abox[2] ? by : abox[2];
                    abox[3] = bx > abox[3] ? bx : abox[3];
                    abox[4] = bz > abox[4] ? bz : abox[4];
                    abox[5] = by > abox[5] ? by : abox[5];
                }
    chebyshev_dilate(cur, nb);
    out.swap(cur);
}

/*
 * Empty-space table, level 2: inside and next to near bricks (D <= 1) the brick table says nothing, yet most
 * of their cells are empty too.  For every cell the squared Euclidean distance, in cells, to the nearest active
 * cell is computed exactly within a window of kNibWindow cells (cube-to-cube distance: per axis max(|d|-1, 0)),
 * as three separable min-plus passes (y, z, x).  A 2^3-cell sub-block keeps min over its cells of
 * floor(sqrt(d2)), capped at 15: from any point of the sub-block the ray may advance that many cell edges
 * without entering an active cell.
 */
const int kNibWindow = 16;
void build_nibble_table(const float* den, int N, int nb, float density_scale, float step_max, std::vector<uint32_t>& out) {
    const int C = N - 1; /* cells per axis */
    const uint16_t INF = 0xffff;
    std::vector<uint8_t> act((size_t)C * C * C);
    parallel_slabs(C, [&](int x0, int x1) {
        for (int x = x0; x < x1; x++)
            for (int z = 0; z < C; z++)
                for (int y = 0; y < C; y++) {
                    bool a = false;
                    for (int k = 0; k < 8 && !a; k++) {
                        const int xx = x + (k >> 2), zz = z + ((k >> 1) & 1), yy = y + (k & 1);
                        a = den[((size_t)xx * N + zz) * N + yy] * density_scale < step_max;
                    }
                    act[((size_t)x * C + z) * C + y] = a ? 1 : 0;
                }
    });
    auto gap2 = [](int d) {
        d = d < 0 ? -d : d;
        d = d > 0 ? d - 1 : 0;
        return d * d;
    };
    std::vector<uint16_t> g((size_t)C * C * C), h((size_t)C * C * C);
    parallel_slabs(C, [&](int x0, int x1) { /* pass y */
        for (int x = x0; x < x1; x++)
            for (int z = 0; z < C; z++)
                for (int y = 0; y < C; y++) {
                    int best = INF;
                    for (int d = -kNibWindow; d <= kNibWindow; d++) {
                        const int yy = y + d;
                        if (yy < 0 || yy >= C || !act[((size_t)x * C + z) * C + yy]) continue;
                        const int q = gap2(d);
                        best = q < best ? q : best;
                    }
                    g[((size_t)x * C + z) * C + y] = (uint16_t)best;
                }
    });
    parallel_slabs(C, [&](int x0, int x1) { /* pass z */
        for (int x = x0; x < x1; x++)
            for (int z = 0; z < C; z++)
                for (int y = 0; y < C; y++) {
                    int best = INF;
                    for (int d = -kNibWindow; d <= kNibWindow; d++) {
                        const int zz = z + d;
                        if (zz < 0 || zz >= C) continue;
                        const int q = gap2(d) + (int)g[((size_t)x * C + zz) * C + y];
                        best = q < best ? q : best;
                    }
                    h[((size_t)x * C + z) * C + y] = (uint16_t)best;
                }
    });
    parallel_slabs(C, [&](int x0, int x1) { /* pass x */
        for (int x = x0; x < x1; x++)
            for (int z = 0; z < C; z++)
                for (int y = 0; y < C; y++) {
                    int best = INF;
                    for (int d = -kNibWindow; d <= kNibWindow; d++) {
                        const int xx = x + d;
                        if (xx < 0 || xx >= C) continue;
                        const int q = gap2(d) + (int)h[((size_t)xx * C + z) * C + y];
                        best = q < best ? q : best;
                    }
                    g[((size_t)x * C + z) * C + y] = (uint16_t)best;
                }
    });
    out.assign((size_t)nb * nb * nb, 0u);
    parallel_slabs(nb, [&](int b0, int b1) {
        for (int bx = b0; bx < b1; bx++)
            for (int bz = 0; bz < nb; bz++)
                for (int by = 0; by < nb; by++) {
                    int e[8];
                    for (int k = 0; k < 8; k++) e[k] = 15;
                    for (int lx = 0; lx < 4; lx++)
                        for (int lz = 0; lz < 4; lz++)
                            for (int ly = 0; ly < 4; ly++) {
                                const int x = bx * 4 + lx, y = by * 4 + ly, z = bz * 4 + lz;
                                if (x >= C || y >= C || z >= C) continue;
                                const int d2 = g[((size_t)x * C + z) * C + y];
                                int r = 0;
                                while (r < 15 && (r + 1) * (r + 1) <= d2) r++;
                                const int k = (lx >> 1) * 4 + (lz >> 1) * 2 + (ly >> 1);
                                e[k] = r < e[k] ? r : e[k];
                            }
                    uint32_t w = 0;
                    for (int k = 0; k < 8; k++) w |= (uint32_t)e[k] << (4 * k);
                    out[((size_t)bx * nb + bz) * nb + by] = w;
                }
    });
}

/* Cube modes (SH/Raytracing_Cube*.hlsl): voxel (x,y,z) is the cube [x,x+1)x[y,y+1)x[z,z+1) cells, solid when
   its density is <= 0 (GetVoxelDensity(currentVoxelPos) <= 0, Raytracing_Cube.hlsl:242).  The table plays the
   role of the reference's collapsed octree (big empty nodes are crossed in one step). */
void build_cube_table(const float* den, int N, int nb, std::vector<uint8_t>& out) {
    std::vector<uint8_t> cur((size_t)nb * nb * nb, 255);
    for (int bx = 0; bx < nb; bx++)
        for (int bz = 0; bz < nb; bz++)
            for (int by = 0; by < nb; by++) {
                bool solid = false;
                for (int lx = 0; lx < 4 && !solid; lx++)
                    for (int lz = 0; lz < 4 && !solid; lz++)
                        for (int ly = 0; ly < 4 && !solid; ly++) {
                            const int x = bx * 4 + lx, y = by * 4 + ly, z = bz * 4 + lz;
                            if (x > N - 2 || y > N - 2 || z > N - 2) continue;
                            solid = den[((size_t)x * N + z) * N + y] <= 0.0f;
                        }
                if (solid) cur[((size_t)bx * nb + bz) * nb + by] = 0;
            }
    chebyshev_dilate(cur, nb);
    out.swap(cur);
}

/* VRT_FORMAT_TEXEL16 (R6): the reference keeps a voxel as sign + 15-bit trunc(|d| * 100) (VDXVoxelVolume::EncodeVoxel,
   Renderer/DX/Private/RDXVoxelVolume.cpp:399-421) and decodes (float)q * 0.01 (DecodeDensity, SH/Include/Voxel.hlsli:254-266).
   The march works on the integer field i = +-q (held as floats; int16 on the device) with the density unit 0.01
   folded into the volume's density scale: the same field, 0.01 * i, with one rounding less per tap. */
inline float texel16_value(float d) {
    const float a = fabsf(d) * 100.0f;
    uint32_t q = 0;
    if (a >= 4294967040.0f) q = 0xffffffffu;
    else if (a >= 0.0f) q = (uint32_t)a; /* NaN -> 0 */
    q &= 0x7fffu;
    const float v = (float)q;
    return d < 0.0f ? -v : v;
}

uint64_t hash_floats(const float* p, size_t n) {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)n;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
    for (size_t i = 0; i < n; i++) {
        h ^= w[i];
        h *= 0x100000001b3ull;
        h = (h << 29) | (h >> 35);
    }
    return h;
}

std::mutex g_cache_mutex;
std::vector<std::shared_ptr<Derived>> g_cache; /* most recently used last */

/* The derived data of one volume for one metric; `want_cube` adds the Cube modes' table. */
std::shared_ptr<const Derived> derive(const vrto_volume& s, int N, int nb, bool want_cube, bool want_literal = false) {
    const size_t count = (size_t)N * N * N;
    const uint64_t hsh = hash_floats(s.density, count);
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    std::shared_ptr<Derived> d;
    for (size_t i = 0; i < g_cache.size(); i++) {
        Derived& c = *g_cache[i];
        if (c.key_hash == hsh && c.N == N && c.format == s.format && c.density_scale == s.density_scale && c.step_max == s.step_max) {
            d = g_cache[i];
            g_cache.erase(g_cache.begin() + (long)i);
            break;
        }
    }
    if (!d) {
        d = std::make_shared<Derived>();
        d->key_hash = hsh;
        d->N = N;
        d->format = s.format;
        d->density_scale = s.density_scale;
        d->step_max = s.step_max;
        const float* field = s.density;
        float scale = s.density_scale;
        if (s.format == VRT_FORMAT_TEXEL16) {
            d->field.resize(count);
            for (size_t i = 0; i < count; i++) d->field[i] = texel16_value(s.density[i]);
            field = d->field.data();
            scale = s.density_scale * 0.01f;
        }
        if (s.step_max > 0.0f) {
            build_skip_table(field, N, nb, scale, s.step_max, d->skip, d->abox);
            build_nibble_table(field, N, nb, scale, s.step_max, d->nib);
        }
    }
    if (want_cube && !d->has_cube) {
        build_cube_table(s.format == VRT_FORMAT_TEXEL16 ? d->field.data() : s.density, N, nb, d->cube_skip);
        d->has_cube = true;
    }
    if (want_literal && !d->lit) ensure_literal(*d, s.density, N, s.resolution);
    g_cache.push_back(d);
    while (g_cache.size() > 6) g_cache.erase(g_cache.begin());
    return d;
}

bool pack(const vrt_scene* scene, const vrto_volume* volumes, const uint8_t* env, int env_size,
          const vrt_params* prm, Packed& P, bool literal = false) {
    if (!scene || !volumes || !prm) return false;
    if (prm->width <= 0 || prm->height <= 0) return false;
    if (scene->n_instances < 0 || scene->n_instances > VRT_MAX_INSTANCES) return false;
    P.cam = camera_basis(scene, prm->width, prm->height);
    for (int i = 0; i < VRT_MAX_VOLUMES; i++) {
        Volume& v = P.vol[i];
        const vrto_volume& s = volumes[i];
        v.den = s.density;
        if (!s.density) continue;
        v.N = (1 << s.resolution) + 1;
        v.extent = s.extent;
        v.cell = (s.extent * 2.0f) / (float)(v.N - 1);
        v.inv_cell = 1.0f / v.cell;
        v.density_scale = s.format == VRT_FORMAT_TEXEL16 ? s.density_scale * 0.01f : s.density_scale;
        v.step_max = s.step_max > 0.0f ? s.step_max : std::numeric_limits<float>::infinity();
        v.tint[0] = s.material.tint[0];
        v.tint[1] = s.material.tint[1];
        v.tint[2] = s.material.tint[2];
        v.roughness = minf(maxf(s.material.roughness, 0.0f), 1.0f);
        v.metallic = minf(maxf(s.material.metallic, 0.0f), 1.0f);
        v.roughness_raw = s.material.roughness;
        v.metallic_raw = s.material.metallic;
        v.tex[0] = s.albedo_tex;
        v.tex[1] = s.normal_tex;
        v.tex[2] = s.rm_tex;
        v.tex_scale[0] = s.tex_scale[0];
        v.tex_scale[1] = s.tex_scale[1];
        v.textured = false;
        for (int ti = 0; ti < 3; ti++) {
            if (v.tex[ti].rgba8 && (v.tex[ti].width <= 0 || v.tex[ti].height <= 0)) return false;
            v.textured = v.textured || v.tex[ti].rgba8 != nullptr;
        }
        float r1 = s.material.roughness + 1.0f;
        v.k = (r1 * r1) / 8.0f; /* RDXVoxelVolume.cpp:383, from the unclamped roughness */
        v.nb = (v.N - 1 + 3) / 4;
        if (s.format != VRT_FORMAT_F32 && s.format != VRT_FORMAT_TEXEL16) return false;
        if (literal && s.resolution > 8) return false; /* the reference's octree walk ends at depth 8 (Voxel.hlsli:316) */
        v.derived = derive(s, v.N, v.nb, prm->mode >= VRT_MODE_CUBE, literal);
        v.lit = literal ? v.derived->lit.get() : nullptr;
        if (s.format == VRT_FORMAT_TEXEL16) v.den = v.derived->field.data();
        v.skip = s.step_max > 0.0f ? v.derived->skip.data() : nullptr;
        v.nib = s.step_max > 0.0f ? v.derived->nib.data() : nullptr;
        for (int a = 0; a < 3; a++) { /* brick box {x, z, y} -> object-space box per axis x, y, z */
            const int ax = a == 0 ? 0 : (a == 1 ? 2 : 1);
            const int lo_cell = v.derived->abox[ax] * 4;
            int hi_cell = (v.derived->abox[3 + ax] + 1) * 4;
            hi_cell = hi_cell < v.N - 1 ? hi_cell : v.N - 1;
            v.alo[a] = (float)lo_cell * v.cell - v.extent;
            v.ahi[a] = (float)hi_cell * v.cell - v.extent;
        }
        v.cube_skip = prm->mode >= VRT_MODE_CUBE ? v.derived->cube_skip.data() : nullptr;
    }
    P.n_inst = scene->n_instances;
    for (int i = 0; i < P.n_inst; i++) {
        int slot = scene->instances[i].volume_slot;
        if (slot < 0 || slot >= VRT_MAX_VOLUMES || !volumes[slot].density) return false;
        P.inst[i] = build_instance(scene->instances[i]);
    }
    P.env = env;
    P.env_size = env ? env_size : 0;
    P.light_dir = v3(scene->light_dir[0], scene->light_dir[1], scene->light_dir[2]);
    P.light_strength = scene->light_strength;
    P.scene = scene;
    P.prm = *prm;
    P.steps_img = nullptr;
    P.literal = literal;
    return true;
}

inline float lerp1(float a, float b, float w) { return fmaf(w, b - a, a); }

const float kRelaxGate = 0.8f; /* over-relaxation: a step is stretched only when the sample is at least this fraction of the one before */
const int kRefine = 3; /* secant samples spent on a hit that overshot into the surface */

/* Trilinear interpolant of cell (cx,cy,cz) at fractional position (fx,fy,fz). */
inline float trilinear(const Volume& v, int cx, int cy, int cz, float fx, float fy, float fz) {
    const size_t N = (size_t)v.N;
    const float* b0 = v.den + ((size_t)cx * N + (size_t)cz) * N + (size_t)cy; /* x0, z0 */
    const float* b1 = b0 + N;                                                  /* x0, z1 */
    const float* b2 = b0 + N * N;                                              /* x1, z0 */
    const float* b3 = b2 + N;                                                  /* x1, z1 */
    /* z, then x, then y (the order the kernel's packed arithmetic takes: the two y values of a load travel together) */
    float a0y0 = lerp1(b0[0], b1[0], fz), a0y1 = lerp1(b0[1], b1[1], fz); /* x0 */
    float a1y0 = lerp1(b2[0], b3[0], fz), a1y1 = lerp1(b2[1], b3[1], fz); /* x1 */
    float c0 = lerp1(a0y0, a1y0, fx);
    float c1 = lerp1(a0y1, a1y1, fx);
    return lerp1(c0, c1, fy);
}

struct Stats {
    uint64_t primary_rays = 0, shadow_rays = 0, bounce_rays = 0;
    uint64_t primary_steps = 0, shadow_steps = 0, hits = 0, exhausted = 0;
};
/* marches (ray x instance) that ran out of budget inside the volume, counted per thread and collected by vrto_render */
thread_local uint64_t g_exhausted = 0;
/* debug only (vrto_debug_set_steps_image): march positions visited, sampled or skipped, by the primary ray [0] and by the
   rays that follow it [1] — what the length of a GPU lane's dependent chain is made of */
thread_local uint64_t g_positions[2] = {0, 0};
thread_local int g_ray_class = 0;
/* study only (vrto_debug_unnormalised_offsets): the reference does not normalise its camera direction (Ray.hlsli:44-45), so its 0.1 back-off of
   the camera ray's secondary rays is 0.1 * |direction| (1 ... 1.5 towards the frame's corners); 1 everywhere else */
thread_local float g_dir_scale = 1.0f;
bool g_unnormalised_offsets = false;
/* debug only (vrto_debug_set_lead_image): positions the primary ray SKIPPED before its first sample — what a beam pre-pass could take over */
thread_local uint64_t g_leading_skips = 0;
/* debug only (vrto_debug_set_position_log): one record {t, s or NaN when skipped, leap, step taken} per position */
float* g_pos_log = nullptr;
int g_pos_log_cap = 0, g_pos_log_n = 0;
inline void log_position(float t, float s, float leap, float step) {
    if (g_pos_log && g_pos_log_n < g_pos_log_cap) {
        float* r = g_pos_log + 4 * (size_t)g_pos_log_n++;
        r[0] = t; r[1] = s; r[2] = leap; r[3] = step;
    }
}

struct HitRec {
    float t;
    int inst;
    V3 n_world;
    bool unlit = false; /* literal reference intersection only: n_world is the pixel's colour (attr.unlit, Raytracing.hlsl:44-48) */
};

/* Ray-box slab test, box = [-e,+e]^3, inf-safe reciprocals. */
inline bool slab(V3 o, V3 d, float e, float t_cur, float& t_enter, float& t_exit) {
    const float inf = std::numeric_limits<float>::infinity();
    float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    float tmin[3], tmax[3];
    for (int a = 0; a < 3; a++) {
        bool pos = dd[a] > 0.0f;
        float inv = dd[a] != 0.0f ? 1.0f / dd[a] : (pos ? inf : -inf);
        float lo = pos ? -e : e;
        float hi = pos ? e : -e;
        tmin[a] = (lo - oo[a]) * inv;
        tmax[a] = (hi - oo[a]) * inv;
    }
    t_enter = maxf(maxf(tmin[0], tmin[1]), tmin[2]);
    t_exit = minf(minf(tmax[0], tmax[1]), tmax[2]);
    return t_exit > t_enter && t_exit >= 0.0f && t_enter <= t_cur;
}

/*
 * Cube modes: exact traversal of the voxel grid (SH/Raytracing_Cube.hlsl:142-295; GoToNextVoxel with face
 * normal, Voxel.hlsli:133-187).  A node is one cell, or — where the table says the nearest solid voxel is D
 * bricks away — the box of (2D-1)^3 bricks around the current brick, crossed in one step like one of the
 * reference's merged octree nodes.  The hit is the entry point of the first solid voxel; the normal is the
 * face the ray came through (the AABB face for the first voxel, (0,0,0) when the ray starts inside the
 * volume).  One "step" = one node visit.  The crossing axis is chosen in the reference's order (x if
 * strictly smallest, else y if smaller than z, else z); on that axis the next voxel index is an integer
 * step, the other two are re-derived from the crossing point and clamped into the node's cross-section.
 */
bool march_cube(const Packed& P, int ii, V3 o, V3 d, float t_cur, bool want_normal, float& t_hit, V3& n_world,
                uint64_t& steps) {
    const Instance& I = P.inst[ii];
    const Volume& V = P.vol[I.slot];
    const float inf = std::numeric_limits<float>::infinity();
    V3 oo = mul(I.w2o, o - I.pos);
    V3 od = mul(I.w2o, d);
    float t_enter, t_exit;
    if (!slab(oo, od, V.extent, t_cur, t_enter, t_exit)) return false;
    const float uo[3] = {(oo.x + V.extent) * V.inv_cell, (oo.y + V.extent) * V.inv_cell, (oo.z + V.extent) * V.inv_cell};
    const V3 udv = od * V.inv_cell;
    const float ud[3] = {udv.x, udv.y, udv.z};
    float inv[3];
    for (int a = 0; a < 3; a++) inv[a] = ud[a] != 0.0f ? 1.0f / ud[a] : inf;
    const int cmax = V.N - 2;
    float t = (t_enter > 0.0f ? t_enter : 0.0f) + P.prm.eps_in;
    const float t_end = minf(t_exit, t_cur);
    int c[3];
    for (int a = 0; a < 3; a++) c[a] = (int)minf(maxf(floorf(fmaf(ud[a], t, uo[a])), 0.0f), (float)cmax);
    int axis_in = -1;
    for (int i = 0; i < P.prm.max_steps; i++) {
        if (t > t_end) return false;
        steps++;
        const int dist = V.cube_skip[((size_t)(c[0] >> 2) * V.nb + (size_t)(c[2] >> 2)) * V.nb + (size_t)(c[1] >> 2)];
        int lo[3], hi[3];
        if (dist == 0) {
            if (V.den[((size_t)c[0] * V.N + (size_t)c[2]) * V.N + (size_t)c[1]] <= 0.0f) {
                t_hit = t;
                if (want_normal) {
                    V3 n = v3(0.0f, 0.0f, 0.0f);
                    if (axis_in >= 0) {
                        const float f = ud[axis_in] > 0.0f ? -1.0f : 1.0f;
                        n = v3(axis_in == 0 ? f : 0.0f, axis_in == 1 ? f : 0.0f, axis_in == 2 ? f : 0.0f);
                    } else if (t_enter >= 0.0f) {
                        /* first voxel of a ray that entered from outside: AABB-face normal (Raytracing_Cube.hlsl:195-212) */
                        const float tb = t_enter - 0.1f;
                        V3 rp = v3(fmaf(od.x, tb, oo.x), fmaf(od.y, tb, oo.y), fmaf(od.z, tb, oo.z));
                        const float e = V.extent;
                        n.x = rp.x > e ? 1.0f : (rp.x < -e ? -1.0f : 0.0f);
                        n.y = rp.y > e ? 1.0f : (rp.y < -e ? -1.0f : 0.0f);
                        n.z = rp.z > e ? 1.0f : (rp.z < -e ? -1.0f : 0.0f);
                        /* not normalised (the reference normalises only in the interpolated modes) */
                    }
                    n_world = mul(I.o2w, n);
                }
                return true;
            }
            for (int a = 0; a < 3; a++) lo[a] = hi[a] = c[a];
        } else {
            for (int a = 0; a < 3; a++) {
                const int b = c[a] >> 2;
                lo[a] = (b - (dist - 1)) * 4;
                hi[a] = (b + (dist - 1)) * 4 + 3;
                lo[a] = lo[a] < 0 ? 0 : lo[a];
                hi[a] = hi[a] > cmax ? cmax : hi[a];
            }
        }
        float tx[3];
        for (int a = 0; a < 3; a++) {
            const float bound = (float)(ud[a] > 0.0f ? hi[a] + 1 : lo[a]);
            tx[a] = ud[a] != 0.0f ? (bound - uo[a]) * inv[a] : inf;
        }
        const int axis = tx[0] < tx[1] ? (tx[0] < tx[2] ? 0 : 2) : (tx[1] < tx[2] ? 1 : 2);
        const float t_new = tx[axis];
        if (!(t_new <= t_end)) return false; /* leaves the march interval (or NaN) before the next node */
        for (int a = 0; a < 3; a++) {
            if (a == axis) {
                c[a] = ud[a] > 0.0f ? hi[a] + 1 : lo[a] - 1;
            } else {
                const float u = floorf(fmaf(ud[a], t_new, uo[a]));
                c[a] = (int)minf(maxf(u, (float)lo[a]), (float)hi[a]);
            }
        }
        if (c[axis] < 0 || c[axis] > cmax) return false;
        t = maxf(t_new, t);
        axis_in = axis;
    }
    if (P.prm.max_steps > 0 && !(t > t_end)) g_exhausted++;
    return false;
}

bool ref_march_instance(const Packed& P, int ii, V3 o, V3 d, float t_cur, bool want_normal, float& t_hit, V3& n_world);

/* March one instance.  Returns true on hit with t (ray parameter, shared with world space). */
bool march_instance(const Packed& P, int ii, V3 o, V3 d, float t_cur, float t_base, bool want_normal,
                    float& t_hit, V3& n_world, uint64_t& steps) {
    if (P.ref_intersection) return ref_march_instance(P, ii, o, d, t_cur, want_normal, t_hit, n_world);
    if (P.prm.mode >= VRT_MODE_CUBE) return march_cube(P, ii, o, d, t_cur, want_normal, t_hit, n_world, steps);
    const Instance& I = P.inst[ii];
    const Volume& V = P.vol[I.slot];
    V3 oo = mul(I.w2o, o - I.pos);
    V3 od = mul(I.w2o, d);
    float t_enter, t_exit;
    if (!slab(oo, od, V.extent, t_cur, t_enter, t_exit)) return false;

    float inv_len = 1.0f / sqrtf(dot(od, od));
    float ds = V.density_scale * inv_len;
    float smax = V.step_max * inv_len;
    /* ray in voxel units */
    V3 uo = v3((oo.x + V.extent) * V.inv_cell, (oo.y + V.extent) * V.inv_cell, (oo.z + V.extent) * V.inv_cell);
    V3 ud = od * V.inv_cell;
    const float cmax = (float)(V.N - 2);

    float t = (t_enter > 0.0f ? t_enter : 0.0f) + P.prm.eps_in;
    float t_end = minf(t_exit, t_cur);
    bool clipped = false;
    /* up to here the hit threshold eps_hit + cone_eps*t is at most smax/2 (the margin covers rounding) */
    const float t_skip_end = P.prm.cone_eps > 0.0f ? (0.5f * smax - P.prm.eps_hit) / P.prm.cone_eps
                                                   : (P.prm.eps_hit + P.prm.eps_hit <= smax ? std::numeric_limits<float>::infinity()
                                                                                             : -std::numeric_limits<float>::infinity());
    if (V.skip && t_end <= t_skip_end) {
        /* The active box: bounding box of the bricks that can hold surface.  Outside it the table would only leap; the
           march is clipped to it (same slab arithmetic, same reciprocals as the volume box) — as long as the whole interval
           lies where an inactive cell cannot produce a hit (t <= t_skip_end): a volume so far away that a pixel's footprint
           exceeds half the step clamp is marched over its whole box. */
        const float inf = std::numeric_limits<float>::infinity();
        const float oo3[3] = {oo.x, oo.y, oo.z}, od3[3] = {od.x, od.y, od.z};
        float tmin[3], tmax[3];
        for (int a = 0; a < 3; a++) {
            const bool pos = od3[a] > 0.0f;
            const float inv = od3[a] != 0.0f ? 1.0f / od3[a] : (pos ? inf : -inf);
            tmin[a] = ((pos ? V.alo[a] : V.ahi[a]) - oo3[a]) * inv;
            tmax[a] = ((pos ? V.ahi[a] : V.alo[a]) - oo3[a]) * inv;
        }
        const float ta = maxf(maxf(tmin[0], tmin[1]), tmin[2]);
        const float tb = minf(minf(tmax[0], tmax[1]), tmax[2]);
        if (!(tb > ta) || !(tb >= 0.0f)) return false;
        if (ta > t) {
            t = ta;
            clipped = true;
        }
        t_end = minf(t_end, tb);
    }
    /* smallest step: one pixel-footprint radius at the total path length t_base + t (t_base = length of
       the path that led to this ray's origin: 0 for camera rays, the hit distance for shadow rays) */
    const float base_min = fmaf(t_base, P.prm.cone_eps, P.prm.step_min);
    const float leap_unit = (4.0f * V.cell) * inv_len; /* one brick edge in ray-parameter units */
    const float cell_unit = leap_unit * 0.25f;         /* one cell edge */
    float t_prev = t, s_prev = 0.0f;
    /* Over-relaxation (k_relax > 1; Keinert et al., "Enhanced Sphere Tracing", 2014): a distance-driven step is stretched by
       k_relax; at the next sample the empty spheres around the two samples must still overlap, else something may have
       been jumped over: the ray goes back and takes the plain step from the previous sample.  relaxed: the step that led to
       the current position was a stretched one. */
    bool relaxed = false;
    bool leading = true; /* (debug statistics) no position of this march has been sampled yet */
    for (int i = 0; i < P.prm.max_steps; i++) {
        if (t > t_end) return false;
        g_positions[g_ray_class]++;
        float ux = fmaf(ud.x, t, uo.x), uy = fmaf(ud.y, t, uo.y), uz = fmaf(ud.z, t, uo.z);
        float cxf = minf(maxf(floorf(ux), 0.0f), cmax);
        float cyf = minf(maxf(floorf(uy), 0.0f), cmax);
        float czf = minf(maxf(floorf(uz), 0.0f), cmax);
        float fx = ux - cxf, fy = uy - cyf, fz = uz - czf;
        int cx = (int)cxf, cy = (int)cyf, cz = (int)czf;
        /* Empty-space leap L from the sub-block table: the sub-block's distance to the nearest active cell in cell edges (at most
           15).  Such a move cannot enter an active cell, and the interpolant is below the step clamp only inside active cells.
           (The brick-level table only bounds the active box since round 3: ONE table word per new brick for the GPU's march.) */
        float leap = 0.0f;
        if (V.skip) {
            const size_t brick = ((size_t)(cx >> 2) * V.nb + (size_t)(cz >> 2)) * V.nb + (size_t)(cy >> 2);
            const int k = ((cx >> 1) & 1) * 4 + ((cz >> 1) & 1) * 2 + ((cy >> 1) & 1);
            leap = (float)((V.nib[brick] >> (4 * k)) & 15u) * cell_unit;
            /* No active cell here: a sample would be >= smax, so it cannot be a hit while the threshold is below smax
               (factor 2: rounding margin), and max(min(s*k, smax), footprint, leap) = max(footprint, leap) once
               leap >= smax.  The ray advances without sampling (no tap is read, no sample is counted). */
            if (leap > 0.0f && leap >= smax && t <= t_skip_end) {
                if (leading && g_ray_class == 0) g_leading_skips++;
                t_prev = t;
                s_prev = smax;
                relaxed = false;
                log_position(t, std::numeric_limits<float>::quiet_NaN(), leap, fmaxf(fmaf(t, P.prm.cone_eps, base_min), leap));
                t = t + fmaxf(fmaf(t, P.prm.cone_eps, base_min), leap);
                continue;
            }
        }
        float s = trilinear(V, cx, cy, cz, fx, fy, fz) * ds;
        steps++;
        leading = false;
        if (relaxed && fmaxf(fminf(s, smax), 0.0f) + s_prev < t - t_prev) {
            /* the spheres do not overlap: back to the previous sample's plain step (that sample stays the "previous" one) */
            relaxed = false;
            log_position(t, s, leap, t_prev + fmaxf(s_prev, fmaf(t_prev, P.prm.cone_eps, base_min)) - t); /* negative: went back */
            t = t_prev + fmaxf(s_prev, fmaf(t_prev, P.prm.cone_eps, base_min));
            continue;
        }
        if (s < fmaf(t, P.prm.cone_eps, P.prm.eps_hit)) {
            if (s < 0.0f && i > 0) {
                /* the step overshot into the surface (band-edge cells of shell volumes interpolate towards the
                   background value; trilinear SDFs are not exactly 1-Lipschitz): walk back to the crossing with
                   kRefine secant (regula falsi) samples between the last outside and the first inside sample */
                float ta = t_prev, sa = s_prev, tb = t, sb = s, tm = t;
                for (int r = 0; r < kRefine; r++) {
                    tm = fmaf(tb - ta, sa / (sa - sb), ta);
                    float mx = fmaf(ud.x, tm, uo.x), my = fmaf(ud.y, tm, uo.y), mz = fmaf(ud.z, tm, uo.z);
                    float mcx = minf(maxf(floorf(mx), 0.0f), cmax);
                    float mcy = minf(maxf(floorf(my), 0.0f), cmax);
                    float mcz = minf(maxf(floorf(mz), 0.0f), cmax);
                    float sm = trilinear(V, (int)mcx, (int)mcy, (int)mcz, mx - mcx, my - mcy, mz - mcz) * ds;
                    steps++;
                    if (sm < 0.0f) { tb = tm; sb = sm; } else { ta = tm; sa = sm; }
                }
                /* report the last secant point: within the residual of the crossing, on either side */
                t = tm;
                ux = fmaf(ud.x, t, uo.x); uy = fmaf(ud.y, t, uo.y); uz = fmaf(ud.z, t, uo.z);
                cxf = minf(maxf(floorf(ux), 0.0f), cmax);
                cyf = minf(maxf(floorf(uy), 0.0f), cmax);
                czf = minf(maxf(floorf(uz), 0.0f), cmax);
                fx = ux - cxf; fy = uy - cyf; fz = uz - czf;
                cx = (int)cxf; cy = (int)cyf; cz = (int)czf;
            }
            else if (want_normal && i > 0 && !(P.prm.flags & VRT_FLAG_NO_HIT_POLISH)) {
                /* Hit polish: the cone threshold stops the ray up to a few pixel footprints IN FRONT of the
                   surface; the reference reports the zero crossing itself and takes its normal there (Voxel.hlsli:691-804).  The
                   hit DECISION above stands; its position moves on to the crossing by the secant rule: first estimate from the
                   march's own last two samples (when the previous one is a real, unclamped sample), else one sphere step;
                   VRT_HIT_POLISH_SAMPLES samples, each followed by the secant through the last two points; never behind the
                   stop point, never more than 8 thresholds ahead, never beyond the interval. */
                const float span = 8.0f * fmaf(t, P.prm.cone_eps, P.prm.eps_hit);
                const float t_far = t + span;
                float ta = t, sa = s;
                float tb = (s_prev > s && s_prev < smax) ? t + (s * (t - t_prev)) / (s_prev - s) : t + s;
                tb = fminf(fmaxf(tb, t), t_far);
                for (int r = 0; r < VRT_HIT_POLISH_SAMPLES; r++) {
                    const float mx = fmaf(ud.x, tb, uo.x), my = fmaf(ud.y, tb, uo.y), mz = fmaf(ud.z, tb, uo.z);
                    const float mcx = minf(maxf(floorf(mx), 0.0f), cmax);
                    const float mcy = minf(maxf(floorf(my), 0.0f), cmax);
                    const float mcz = minf(maxf(floorf(mz), 0.0f), cmax);
                    const float sb = trilinear(V, (int)mcx, (int)mcy, (int)mcz, mx - mcx, my - mcy, mz - mcz) * ds;
                    steps++;
                    /* zero of the line through (ta, sa), (tb, sb); equal samples (the estimate has converged: tb rounds onto ta): stay */
                    const float tm = sa != sb ? fminf(fmaxf(fmaf(tb - ta, sb / (sa - sb), tb), t), t_far) : tb;
                    ta = tb;
                    sa = sb;
                    tb = tm;
                }
                if (tb <= t_end) {
                    t = tb;
                    ux = fmaf(ud.x, t, uo.x); uy = fmaf(ud.y, t, uo.y); uz = fmaf(ud.z, t, uo.z);
                    cxf = minf(maxf(floorf(ux), 0.0f), cmax);
                    cyf = minf(maxf(floorf(uy), 0.0f), cmax);
                    czf = minf(maxf(floorf(uz), 0.0f), cmax);
                    fx = ux - cxf; fy = uy - cyf; fz = uz - czf;
                    cx = (int)cxf; cy = (int)cyf; cz = (int)czf;
                }
            }
            t_hit = t;
            if (want_normal) {
                V3 n;
                if (i == 0 && t_enter >= 0.0f && !clipped) {
                    /* surface cut by the volume boundary: AABB-face normal (Raytracing.hlsl:198-226) */
                    float tb = t_enter - 0.1f;
                    V3 rp = v3(fmaf(od.x, tb, oo.x), fmaf(od.y, tb, oo.y), fmaf(od.z, tb, oo.z));
                    float e = V.extent;
                    n.x = rp.x > e ? 1.0f : (rp.x < -e ? -1.0f : 0.0f);
                    n.y = rp.y > e ? 1.0f : (rp.y < -e ? -1.0f : 0.0f);
                    n.z = rp.z > e ? 1.0f : (rp.z < -e ? -1.0f : 0.0f);
                } else {
                    int N2 = V.N - 2;
                    int xp = cx + 1 > N2 ? N2 : cx + 1, xm = cx - 1 < 0 ? 0 : cx - 1;
                    int yp = cy + 1 > N2 ? N2 : cy + 1, ym = cy - 1 < 0 ? 0 : cy - 1;
                    int zp = cz + 1 > N2 ? N2 : cz + 1, zm = cz - 1 < 0 ? 0 : cz - 1;
                    if (P.prm.flags & VRT_FLAG_REFERENCE_BOUNDARY_TEXELS) {
                        /* A neighbour cell beyond the grid: the reference's Load returns 0 for its samples outside the volume texture
                           (Voxel.hlsli:607-617), so its interpolant is the boundary plane's, weighted with the in-texture side's weight:
                           cell N-1: (1 - f) x the plane of samples N-1 = (1 - f) x cell N-2 at fraction 1; cell -1: f x cell 0 at fraction 0. */
                        n.x = (cx + 1 > N2 ? (1.0f - fx) * trilinear(V, N2, cy, cz, 1.0f, fy, fz) : trilinear(V, xp, cy, cz, fx, fy, fz)) -
                              (cx - 1 < 0 ? fx * trilinear(V, 0, cy, cz, 0.0f, fy, fz) : trilinear(V, xm, cy, cz, fx, fy, fz));
                        n.y = (cy + 1 > N2 ? (1.0f - fy) * trilinear(V, cx, N2, cz, fx, 1.0f, fz) : trilinear(V, cx, yp, cz, fx, fy, fz)) -
                              (cy - 1 < 0 ? fy * trilinear(V, cx, 0, cz, fx, 0.0f, fz) : trilinear(V, cx, ym, cz, fx, fy, fz));
                        n.z = (cz + 1 > N2 ? (1.0f - fz) * trilinear(V, cx, cy, N2, fx, fy, 1.0f) : trilinear(V, cx, cy, zp, fx, fy, fz)) -
                              (cz - 1 < 0 ? fz * trilinear(V, cx, cy, 0, fx, fy, 0.0f) : trilinear(V, cx, cy, zm, fx, fy, fz));
                    } else {
                    n.x = trilinear(V, xp, cy, cz, fx, fy, fz) - trilinear(V, xm, cy, cz, fx, fy, fz);
                    n.y = trilinear(V, cx, yp, cz, fx, fy, fz) - trilinear(V, cx, ym, cz, fx, fy, fz);
                    n.z = trilinear(V, cx, cy, zp, fx, fy, fz) - trilinear(V, cx, cy, zm, fx, fy, fz);
                    }
                }
                float l2 = dot(n, n);
                if (!(l2 > 0.0f)) { /* zero or NaN gradient → (0,0,0) (Voxel.hlsli:794-798) */
                    n = v3(0.0f, 0.0f, 0.0f);
                } else {
                    n = n * (1.0f / sqrtf(l2));
                }
                n_world = mul(I.o2w, n);
            }
            return true;
        }
        const float s_old = s_prev;
        t_prev = t;
        s_prev = fminf(s, smax); /* what a skipped sample would have recorded: the secant of an overshoot repair starts from it */
        const float adv_min = fmaxf(fmaf(t, P.prm.cone_eps, base_min), leap);
        const float om = P.prm.k_relax;
        const float plain = fmaxf(s_prev, adv_min);
        const float stretched = fmaxf(fminf(s * om, om > 1.0f ? smax * om : smax), adv_min);
        /* stretched only while (a) the distance is not falling fast — a ray running at a surface would only overshoot and
           come back, it is the grazing ray whose chain the stretch shortens — and (b) the next sample stays inside the
           interval: past its end there is no sample to check the overlap with */
        relaxed = stretched > plain && s >= kRelaxGate * s_old && t + stretched <= t_end;
        log_position(t, s, leap, (relaxed || om < 1.0f) ? stretched : plain);
        /* k_relax < 1: every distance-driven step is scaled down (a field steeper than a distance), never stretched */
        t = t + ((relaxed || om < 1.0f) ? stretched : plain);
    }
    if (P.prm.max_steps > 0 && !(t > t_end)) g_exhausted++;
    return false;
}

#include "vrt_ref_literal.inl"

/* Closest hit over all instances (ascending index; strict '<' keeps the lower index on ties). */
bool trace_closest(const Packed& P, V3 o, V3 d, float t_max, float t_base, HitRec& h, uint64_t& steps) {
    if (P.literal) return lit_trace_closest(P, o, d, t_max, h);
    bool any = false;
    float best = t_max;
    /* A sphere-trace runs over the instance's whole interval whatever has been hit before: where its stretched steps fall
       depends on the interval's end, so cutting it at the closest hit so far would make the result depend on the order the
       instances are visited in.  (The cell walk of the Cube modes has no such state and stops at the closest hit.) */
    const bool cube = P.prm.mode >= VRT_MODE_CUBE;
    for (int i = 0; i < P.n_inst; i++) {
        float t;
        V3 n;
        if (march_instance(P, i, o, d, cube ? best : t_max, t_base, true, t, n, steps)) {
            if (!any || t < best) {
                any = true;
                best = t;
                h.t = t;
                h.inst = i;
                h.n_world = n;
            }
        }
    }
    return any;
}

bool trace_any(const Packed& P, V3 o, V3 d, float t_max, float t_base, uint64_t& steps) {
    if (P.literal) return lit_trace_any(P, o, d, t_max);
    for (int i = 0; i < P.n_inst; i++) {
        float t;
        V3 n;
        if (march_instance(P, i, o, d, t_max, t_base, false, t, n, steps)) return true;
    }
    return false;
}

void env_lookup(const uint8_t* env, int S, V3 dir, float rgb[3]) {
    if (!env || S <= 0) {
        rgb[0] = rgb[1] = rgb[2] = 0.0f;
        return;
    }
    /* SampleLevel(dir.xzy) */
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
    const uint8_t* px = env + (((size_t)face * S + iy) * S + ix) * 4;
    const float k = 1.0f / 255.0f;
    rgb[0] = (float)px[0] * k;
    rgb[1] = (float)px[1] * k;
    rgb[2] = (float)px[2] * k;
}

const float PI_REF = 3.141592f; /* Constants.hlsli:15 */

/* Radiance(), Lighting.hlsli:50-101 (note F enters twice: inside cook and as its multiplier). */
void radiance(V3 Li, V3 wi, V3 wo, V3 n, V3 albedo, float rough, float metal, float k, V3& out) {
    V3 hv = wi + wo;
    V3 h = normalize(hv);
    V3 f0 = v3(0.04f + (albedo.x - 0.04f) * metal, 0.04f + (albedo.y - 0.04f) * metal, 0.04f + (albedo.z - 0.04f) * metal);
    float a2 = rough * rough;
    float ndoth = maxf(dot(n, h), 0.0f);
    float c = (ndoth * ndoth) * (a2 - 1.0f) + 1.0f;
    float D = a2 / maxf((PI_REF * c) * c, 0.001f);
    float wdoth = maxf(dot(wo, h), 0.0f);
    float m = maxf(-wdoth + 1.0f, 0.0f);
    float m2 = m * m;
    float m5 = (m2 * m2) * m;
    V3 F = v3(f0.x + (-f0.x + 1.0f) * m5, f0.y + (-f0.y + 1.0f) * m5, f0.z + (-f0.z + 1.0f) * m5);
    float dwo = maxf(dot(n, wo), 0.0f);
    float dwi = maxf(dot(n, wi), 0.0f);
    float G = (dwo / (dwo * (1.0f - k) + k)) * (dwi / (dwi * (1.0f - k) + k));
    float den = maxf((4.0f * dwo) * dwi, 0.0001f);
    V3 cook = v3(((D * F.x) * G) / den, ((D * F.y) * G) / den, ((D * F.z) * G) / den);
    float km = 1.0f - metal;
    V3 kd = v3((1.0f - F.x) * km, (1.0f - F.y) * km, (1.0f - F.z) * km);
    V3 brdf = v3((albedo.x / PI_REF) * kd.x + cook.x * F.x,
                 (albedo.y / PI_REF) * kd.y + cook.y * F.y,
                 (albedo.z / PI_REF) * kd.z + cook.z * F.z);
    float ndwi = dot(n, wi);
    out = v3((brdf.x * Li.x) * ndwi, (brdf.y * Li.y) * ndwi, (brdf.z * Li.z) * ndwi);
}

/* ---- tri-planar material textures (SH/Include/Textures.hlsli:16-59, Quaternion.hlsli:18-82) ------------- */

/* Geometry sampler: point filter, wrap addressing (RDXScene.cpp:262-270): texel (floor(frac(u)*W), floor(frac(v)*H)). */
inline V3 tex_point_wrap(const vrto_texture& T, float u, float v) {
    float fu = u - floorf(u), fv = v - floorf(v);
    int x = (int)(fu * (float)T.width), y = (int)(fv * (float)T.height);
    x = x > T.width - 1 ? T.width - 1 : (x < 0 ? 0 : x);   /* frac can round to 1.0; NaN -> 0 */
    y = y > T.height - 1 ? T.height - 1 : (y < 0 ? 0 : y);
    const uint8_t* px = T.rgba8 + ((size_t)y * (size_t)T.width + (size_t)x) * 4;
    return v3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}

/* TriSampleTexture: three planar projections of the object-space position, blended by |normal|. */
inline V3 tri_sample(const vrto_texture& T, const float scale[2], V3 op, V3 blend, bool as_normal) {
    V3 tx = tex_point_wrap(T, op.z / scale[0], op.y / scale[1]);
    V3 ty = tex_point_wrap(T, op.x / scale[0], op.z / scale[1]);
    V3 tz = tex_point_wrap(T, op.x / scale[0], op.y / scale[1]);
    if (as_normal) {
        tx = v3(tx.x * 2.0f - 1.0f, tx.y * 2.0f - 1.0f, tx.z * 2.0f - 1.0f);
        ty = v3(ty.x * 2.0f - 1.0f, ty.y * 2.0f - 1.0f, ty.z * 2.0f - 1.0f);
        tz = v3(tz.x * 2.0f - 1.0f, tz.y * 2.0f - 1.0f, tz.z * 2.0f - 1.0f);
    }
    return v3((tx.x * blend.x + ty.x * blend.y) + tz.x * blend.z, (tx.y * blend.x + ty.y * blend.y) + tz.y * blend.z,
              (tx.z * blend.x + ty.z * blend.y) + tz.z * blend.z);
}

struct Q4 { float x, y, z, w; };
inline Q4 qmul(Q4 a, Q4 b) { /* Quaternion.hlsli:18-24 */
    return {(b.x * a.w + a.x * b.w) + (a.y * b.z - a.z * b.y), (b.y * a.w + a.y * b.w) + (a.z * b.x - a.x * b.z),
            (b.z * a.w + a.z * b.w) + (a.x * b.y - a.y * b.x), a.w * b.w - ((a.x * b.x + a.y * b.y) + a.z * b.z)};
}
/* fromX(n) = from_to_rotation((1,0,0), n), Quaternion.hlsli:46-82 */
inline Q4 quat_from_x(V3 n) {
    const float d = n.x;
    if (d < -0.999999f) return {0.0f, 0.0f, -1.0f, -4.371139e-08f}; /* half turn about cross(up, x) = -z: (axis*sin(pi/2), cos(pi/2)) in fp32 */
    if (d > 0.999999f) return {0.0f, 0.0f, 0.0f, 1.0f};
    Q4 q = {0.0f, -n.z, n.y, 1.0f + d}; /* (cross((1,0,0), n), 1 + d) */
    const float inv = 1.0f / sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    return {q.x * inv, q.y * inv, q.z * inv, q.w * inv};
}
inline V3 rotate_vector(V3 v, Q4 r) { /* Quaternion.hlsli:26-30 */
    Q4 rc = {-r.x, -r.y, -r.z, r.w};
    Q4 t = qmul(r, qmul(Q4{v.x, v.y, v.z, 0.0f}, rc));
    return v3(t.x, t.y, t.z);
}

/* Material at a hit in the textured modes.  n_world is the march's normal; the object-space normal the
   projections are blended by is w2o * n_world (the inverse of the transform that produced it).  Unbound slots
   are exact identities (white albedo, (1,1) roughness/metal factors, untouched normal) — the reference's
   8-bit default normal texel (127,127,255) would tilt every normal by 0.3 degrees; not inherited. */
struct Surface {
    V3 albedo, n;
    float roughness, metallic;
};
inline Surface textured_surface(const Volume& V, const Instance& I, V3 hit_world, V3 n_world) {
    Surface s;
    const V3 op = mul(I.w2o, hit_world - I.pos);
    const V3 no = mul(I.w2o, n_world);
    const V3 an = v3(fabsf(no.x), fabsf(no.y), fabsf(no.z));
    const float sum = (an.x + an.y) + an.z;
    const V3 blend = v3(an.x / sum, an.y / sum, an.z / sum);
    s.albedo = v3(V.tint[0], V.tint[1], V.tint[2]);
    if (V.tex[0].rgba8) {
        const V3 t = tri_sample(V.tex[0], V.tex_scale, op, blend, false);
        s.albedo = v3(s.albedo.x * t.x, s.albedo.y * t.y, s.albedo.z * t.z);
    }
    s.roughness = V.roughness;
    s.metallic = V.metallic;
    if (V.tex[2].rgba8) {
        const V3 t = tri_sample(V.tex[2], V.tex_scale, op, blend, false);
        s.roughness = minf(maxf(V.roughness_raw * t.x, 0.0f), 1.0f);
        s.metallic = minf(maxf(V.metallic_raw * t.y, 0.0f), 1.0f);
    }
    s.n = n_world;
    if (V.tex[1].rgba8) {
        V3 t = tri_sample(V.tex[1], V.tex_scale, op, blend, true);
        t = normalize(t);
        const V3 tn = v3(t.z, t.x, t.y); /* tNormal.zxy */
        s.n = mul(I.o2w, rotate_vector(tn, quat_from_x(no)));
    }
    return s;
}

const int MAX_DEPTH = 3; /* MAX_RAY_RECURSION_DEPTH, RaytracingHlsl.h:32 */

/* TraceRadianceRay + VRClosestHit / VRMiss, level = 1 for the primary ray. */
/* view_scale: |WorldRayDirection()| of the reference for this ray — 1, or with VRT_FLAG_REFERENCE_VIEW_VECTOR the camera ray's un-normalised
   length: wo = -view_scale * d, secondary rays start 0.1 * view_scale back (Raytracing.hlsl:51-52,85-95). */
V3 radiance_ray(const Packed& P, V3 o, V3 d, int level, float t_base, Stats& st, float view_scale = 1.0f) {
    HitRec h;
    uint64_t steps = 0;
    bool hit = trace_closest(P, o, d, 10000.0f, t_base, h, steps);
    g_ray_class = 1; /* everything this pixel traces from here on follows its primary ray */
    st.primary_steps += steps;
    if (!hit) {
        float rgb[3];
        env_lookup(P.env, P.env_size, d, rgb);
        return v3(rgb[0], rgb[1], rgb[2]);
    }
    st.hits++;
    if (h.unlit) return h.n_world;
    const Volume& V = P.vol[P.inst[h.inst].slot];
    V3 albedo = v3(V.tint[0], V.tint[1], V.tint[2]);
    int mode = P.prm.mode;
    V3 hit_pos = v3(fmaf(d.x, h.t, o.x), fmaf(d.y, h.t, o.y), fmaf(d.z, h.t, o.z));
    V3 n = h.n_world;
    float rough = V.roughness, metal = V.metallic;
    /* Interp / Cube (and their Unlit variants) sample the material textures; the NoTex variants do not
       (Raytracing.hlsl:64-70 vs Raytracing_NoTex.hlsl:64-68).  A material without textures is the same either way. */
    const bool tex_mode = mode == VRT_MODE_INTERP || mode == VRT_MODE_INTERP_UNLIT || mode == VRT_MODE_CUBE || mode == VRT_MODE_CUBE_UNLIT;
    if (tex_mode && V.textured) {
        const Surface sf = textured_surface(V, P.inst[h.inst], hit_pos, n);
        albedo = sf.albedo;
        n = sf.n;
        rough = sf.roughness;
        metal = sf.metallic;
    }
    if (mode == VRT_MODE_INTERP_UNLIT || mode == VRT_MODE_INTERP_NOTEX_UNLIT || mode == VRT_MODE_CUBE_UNLIT ||
        mode == VRT_MODE_CUBE_NOTEX_UNLIT)
        return albedo;

    /* secondary rays start 0.1 back along the ray (Raytracing.hlsl:52), 0.2 in the Cube modes (Raytracing_Cube.hlsl:52) */
    const float back = (mode >= VRT_MODE_CUBE ? 0.2f : 0.1f) * (g_dir_scale * view_scale); /* g_dir_scale: 1, or (study only) |un-normalised camera direction| */
    V3 so = v3(hit_pos.x - d.x * back, hit_pos.y - d.y * back, hit_pos.z - d.z * back);
    V3 wo = v3(-d.x * view_scale, -d.y * view_scale, -d.z * view_scale);
    bool shadows = level < MAX_DEPTH; /* TraceShadowRay's recursion guard, Ray.hlsli:83-86 */
    V3 diffuse = v3(0.0f, 0.0f, 0.0f); /* SHADOW_BRIGHTNESS */

    const vrt_scene* S = P.scene;
    const int npl = S->n_point_lights < VRT_MAX_POINT_LIGHTS ? S->n_point_lights : VRT_MAX_POINT_LIGHTS;
    const int nsl = S->n_spot_lights < VRT_MAX_SPOT_LIGHTS ? S->n_spot_lights : VRT_MAX_SPOT_LIGHTS;
    const bool bounce = rough < 0.3f && level <= P.prm.max_bounces && level < MAX_DEPTH;
    /* A surface facing away from the directional light gets a contribution <= 0 from it whether the shadow ray is
       blocked (0) or not (BRDF >= 0, Li >= 0, n.wi <= 0).  When that light is the only term of this hit's colour (no
       point / spot light, no mirror bounce) the colour is <= 0 either way and every consumer clamps it to 0 (tone-map;
       the parent's max(0, fade)): the shadow ray cannot change the pixel and is not cast. */
    const bool lone_backfacing = npl == 0 && nsl == 0 && !bounce && !(dot(n, P.light_dir) > 0.0f);
    bool shadowed = false;
    if (P.prm.shadow && shadows && !lone_backfacing) {
        st.shadow_rays++;
        uint64_t ss = 0;
        shadowed = trace_any(P, so, P.light_dir, 5000.0f, t_base + h.t, ss);
        st.shadow_steps += ss;
    }
    if (!shadowed) {
        V3 r;
        V3 Li = v3(P.light_strength, P.light_strength, P.light_strength);
        radiance(Li, P.light_dir, wo, n, albedo, rough, metal, V.k, r);
        diffuse = diffuse + r;
    }

    for (int i = 0; i < npl; i++) {
        const vrt_point_light& L = S->point_lights[i];
        V3 lp = v3(L.position[0], L.position[1], L.position[2]);
        V3 dl = lp - so;
        float dist = sqrtf(dot(dl, dl));
        float inten = L.intensity / ((1.0f + L.att_linear * dist) + (L.att_exp * dist) * dist);
        if (inten > 0.005f) {
            V3 ld = dl * (1.0f / dist);
            bool sh = false;
            if (P.prm.shadow && shadows) {
                st.shadow_rays++;
                uint64_t ss = 0;
                sh = trace_any(P, so, ld, dist, t_base + h.t, ss);
                st.shadow_steps += ss;
            }
            if (!sh) {
                V3 r;
                radiance(v3(L.color[0] * inten, L.color[1] * inten, L.color[2] * inten), ld, wo, n, albedo,
                         rough, metal, V.k, r);
                diffuse = diffuse + r;
            }
        }
    }
    for (int i = 0; i < nsl; i++) {
        const vrt_spot_light& L = S->spot_lights[i];
        V3 lp = v3(L.position[0], L.position[1], L.position[2]);
        V3 dl = lp - so;
        float dist = sqrtf(dot(dl, dl));
        /* ComputeSpotLightIntensity, Lighting.hlsli:30-48 */
        V3 sd = v3(-dl.x, -dl.y, -dl.z) * (1.0f / dist);
        float cs = dot(v3(L.forward[0], L.forward[1], L.forward[2]), sd);
        float inten = 0.0f;
        if (cs >= 0.0f && cs > L.cos_angle) {
            float delta = (cs - L.cos_angle) / (L.cos_falloff_angle - L.cos_angle);
            float fall = minf(delta, 1.0f);
            float base = L.intensity * fall;
            inten = base / ((1.0f + L.att_linear * dist) + (L.att_exp * dist) * dist);
        }
        if (inten > 0.01f) {
            V3 ld = dl * (1.0f / dist);
            bool sh = false;
            if (P.prm.shadow && shadows) {
                st.shadow_rays++;
                uint64_t ss = 0;
                sh = trace_any(P, so, ld, dist, t_base + h.t, ss);
                st.shadow_steps += ss;
            }
            if (!sh) {
                V3 r;
                radiance(v3(L.color[0] * inten, L.color[1] * inten, L.color[2] * inten), ld, wo, n, albedo,
                         rough, metal, V.k, r);
                diffuse = diffuse + r;
            }
        }
    }
    /* Mirror bounce (Raytracing.hlsl:79-90).  The reference adds it before the direct light; here it is
       added last so that the kernel can evaluate the recursion as a loop with the same rounding
       (colour = direct + reflection at every level). */
    if (bounce) {
        float dn = dot(d, n);
        V3 rd = normalize(v3(d.x - (2.0f * dn) * n.x, d.y - (2.0f * dn) * n.y, d.z - (2.0f * dn) * n.z));
        st.bounce_rays++;
        const float keep_scale = g_dir_scale;
        g_dir_scale = 1.0f; /* the mirror ray's direction is normalised (Raytracing.hlsl:84) */
        V3 rc = radiance_ray(P, so, rd, level + 1, t_base + h.t, st);
        g_dir_scale = keep_scale;
        float fade = rough * 2.2f;
        rc = v3(maxf(0.0f, rc.x + (0.0f - rc.x) * fade), maxf(0.0f, rc.y + (0.0f - rc.y) * fade),
                maxf(0.0f, rc.z + (0.0f - rc.z) * fade));
        V3 r;
        radiance(rc, rd, wo, n, albedo, rough, metal, V.k, r);
        diffuse = diffuse + r;
    }
    return diffuse;
}

inline float tonemap(float c) {
    c = c > 0.0f ? c : 0.0f; /* negative / NaN → 0, what the UNORM render target keeps */
    c = c / (c + 1.0f);
    return powf(c, 1.0f / 2.2f);
}

void render_rows(const Packed& P, int y0, int y1, int row0, float* out, Stats& st) {
    int W = P.prm.width, H = P.prm.height;
    const uint64_t exhausted0 = g_exhausted;
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            V3 o, d;
            float view_scale = 1.0f;
            camera_ray(P.cam, W, H, x, y, o, d, &view_scale);
            if (!(P.prm.flags & VRT_FLAG_REFERENCE_VIEW_VECTOR)) view_scale = 1.0f;
            if (P.literal) { /* the literal restatement takes its own direction; its length travels with it */
                view_scale = 1.0f;
                if (!(P.lit_options & VRTO_LIT_NORMALISED_CAMERA)) camera_ray_raw(P.cam, W, H, x, y, o, d);
            }
            if (g_unnormalised_offsets) {
                const float sx = (((float)x + 0.5f) / (float)W) * 2.0f - 1.0f, sy = (((float)y + 0.5f) / (float)H) * 2.0f - 1.0f;
                g_dir_scale = sqrtf(1.0f + (sx * P.cam.cx) * (sx * P.cam.cx) + (sy * P.cam.cy) * (sy * P.cam.cy));
            }
            st.primary_rays++;
            g_positions[0] = g_positions[1] = 0;
            g_ray_class = 0;
            g_leading_skips = 0;
            V3 c = radiance_ray(P, o, d, 1, 0.0f, st, view_scale);
            if (P.steps_img) {
                const uint64_t a = g_positions[0] < 0xffffu ? g_positions[0] : 0xffffu, b = g_positions[1] < 0xffffu ? g_positions[1] : 0xffffu;
                P.steps_img[(size_t)(y - row0) * W + x] = (uint32_t)(a | (b << 16));
            }
            if (P.lead_img) P.lead_img[(size_t)(y - row0) * W + x] = (uint32_t)g_leading_skips;
            float* px = out + ((size_t)(y - row0) * W + x) * 4;
            px[0] = tonemap(c.x);
            px[1] = tonemap(c.y);
            px[2] = tonemap(c.z);
            px[3] = 1.0f;
        }
    }
    st.exhausted += g_exhausted - exhausted0;
}

bool mode_supported(int mode) { return mode >= VRT_MODE_INTERP && mode <= VRT_MODE_CUBE_NOTEX_UNLIT; }

}  // namespace

extern "C" {

static uint32_t* g_steps_img = nullptr;
static uint32_t* g_lead_img = nullptr;
void vrto_debug_set_steps_image(uint32_t* img) { g_steps_img = img; }
void vrto_debug_set_lead_image(uint32_t* img) { g_lead_img = img; }
void vrto_debug_unnormalised_offsets(int on) { g_unnormalised_offsets = on != 0; if (!on) g_dir_scale = 1.0f; }
int vrto_debug_set_position_log(float* records, int capacity) {
    const int n = g_pos_log_n;
    g_pos_log = records;
    g_pos_log_cap = records ? capacity : 0;
    g_pos_log_n = 0;
    return n;
}

int vrto_render(const vrt_scene* scene, const vrto_volume* volumes, const uint8_t* env_rgba8, int env_face_size,
                const vrt_params* params, int row0, int rows, float* out_rgba, vrto_stats* stats_or_null,
                int threads) {
    Packed* P = new Packed;
    if (!pack(scene, volumes, env_rgba8, env_face_size, params, *P) || !out_rgba) {
        delete P;
        return VRT_ERR_INVALID;
    }
    if (!mode_supported(params->mode)) {
        delete P;
        return VRT_ERR_UNSUPPORTED;
    }
    if (row0 < 0 || rows < 0 || row0 + rows > params->height) {
        delete P;
        return VRT_ERR_INVALID;
    }
    P->steps_img = g_steps_img;
    P->lead_img = g_lead_img;
    if (threads < 1) threads = 1;
    if (threads > rows && rows > 0) threads = rows;
    std::vector<Stats> st((size_t)threads);
    if (threads == 1) {
        render_rows(*P, row0, row0 + rows, row0, out_rgba, st[0]);
    } else {
        /* interleaved 4-row strips so that sky rows and object rows spread over the threads */
        std::vector<std::thread> th;
        for (int k = 0; k < threads; k++) {
            th.emplace_back([&, k]() {
                for (int y = row0 + k * 4; y < row0 + rows; y += threads * 4) {
                    int ye = y + 4 < row0 + rows ? y + 4 : row0 + rows;
                    render_rows(*P, y, ye, row0, out_rgba, st[(size_t)k]);
                }
            });
        }
        for (auto& t : th) t.join();
    }
    if (stats_or_null) {
        vrto_stats s;
        memset(&s, 0, sizeof s);
        for (auto& a : st) {
            s.primary_rays += a.primary_rays;
            s.shadow_rays += a.shadow_rays;
            s.bounce_rays += a.bounce_rays;
            s.primary_steps += a.primary_steps;
            s.shadow_steps += a.shadow_steps;
            s.hits += a.hits;
            s.exhausted_rays += a.exhausted;
        }
        *stats_or_null = s;
    }
    delete P;
    return VRT_OK;
}

int vrto_trace(const vrt_scene* scene, const vrto_volume* volumes, const vrt_params* params, const float origin[3],
               const float dir[3], float t_max, float* t_out, float normal_out[3], int* instance_out,
               int* steps_out) {
    Packed* P = new Packed;
    vrt_params prm = *params;
    if (prm.width <= 0) prm.width = 1;
    if (prm.height <= 0) prm.height = 1;
    if (!pack(scene, volumes, nullptr, 0, &prm, *P)) {
        delete P;
        return VRT_ERR_INVALID;
    }
    V3 o = v3(origin[0], origin[1], origin[2]);
    V3 d = normalize(v3(dir[0], dir[1], dir[2]));
    HitRec h;
    uint64_t steps = 0;
    bool hit = trace_closest(*P, o, d, t_max, 0.0f, h, steps);
    if (steps_out) *steps_out = (int)steps;
    if (hit) {
        if (t_out) *t_out = h.t;
        if (normal_out) {
            normal_out[0] = h.n_world.x;
            normal_out[1] = h.n_world.y;
            normal_out[2] = h.n_world.z;
        }
        if (instance_out) *instance_out = h.inst;
    }
    delete P;
    return hit ? 1 : 0;
}

/* vrto_trace for n rays with the scene packed ONCE (packing builds the empty-space tables: 45 ms for a 256^3 shell). */
int vrto_trace_batch(const vrt_scene* scene, const vrto_volume* volumes, const vrt_params* params, int n, const float* origins,
                     const float* dirs, float t_max, uint8_t* hit_out, float* t_out, float* normal_out_or_null, int threads) {
    if (n < 0 || (n > 0 && (!origins || !dirs || !hit_out || !t_out))) return VRT_ERR_INVALID;
    std::unique_ptr<Packed> P(new Packed);
    vrt_params prm = *params;
    if (prm.width <= 0) prm.width = 1;
    if (prm.height <= 0) prm.height = 1;
    if (!pack(scene, volumes, nullptr, 0, &prm, *P)) return VRT_ERR_INVALID;
    if (threads < 1) threads = 1;
    auto work = [&](int k) {
        for (int i = k; i < n; i += threads) {
            const V3 o = v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
            const V3 d = normalize(v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
            HitRec h;
            uint64_t steps = 0;
            const bool hit = trace_closest(*P, o, d, t_max, 0.0f, h, steps);
            hit_out[i] = hit ? 1 : 0;
            t_out[i] = hit ? h.t : 0.0f;
            if (normal_out_or_null) {
                normal_out_or_null[3 * i] = hit ? h.n_world.x : 0.0f;
                normal_out_or_null[3 * i + 1] = hit ? h.n_world.y : 0.0f;
                normal_out_or_null[3 * i + 2] = hit ? h.n_world.z : 0.0f;
            }
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < threads; k++) th.emplace_back(work, k);
        for (auto& t : th) t.join();
    }
    return 0;
}

void vrto_camera_ray(const vrt_scene* scene, int width, int height, int px, int py, float origin[3], float dir[3]) {
    Camera c = camera_basis(scene, width, height);
    V3 o, d;
    camera_ray(c, width, height, px, py, o, d);
    origin[0] = o.x; origin[1] = o.y; origin[2] = o.z;
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

float vrto_sample(const vrto_volume* vol, const float p[3]) {
    Volume V;
    V.den = vol->density;
    V.N = (1 << vol->resolution) + 1;
    V.extent = vol->extent;
    V.cell = (vol->extent * 2.0f) / (float)(V.N - 1);
    V.inv_cell = 1.0f / V.cell;
    float cmax = (float)(V.N - 2);
    float u[3], cf[3], f[3];
    for (int a = 0; a < 3; a++) {
        u[a] = (p[a] + V.extent) * V.inv_cell;
        cf[a] = minf(maxf(floorf(u[a]), 0.0f), cmax);
        f[a] = u[a] - cf[a];
    }
    return trilinear(V, (int)cf[0], (int)cf[1], (int)cf[2], f[0], f[1], f[2]);
}

int vrto_debug_tables(const vrto_volume* vol, uint8_t* skip_out, uint32_t* nib_out, float* field_out) {
    if (!vol || !vol->density || !(vol->step_max > 0.0f)) return VRT_ERR_INVALID;
    const int N = (1 << vol->resolution) + 1, nb = (N - 1 + 3) / 4;
    std::shared_ptr<const Derived> d = derive(*vol, N, nb, false);
    const size_t n = (size_t)nb * nb * nb;
    if (skip_out) memcpy(skip_out, d->skip.data(), n);
    if (nib_out) memcpy(nib_out, d->nib.data(), n * sizeof(uint32_t));
    if (field_out) {
        const float* f = vol->format == VRT_FORMAT_TEXEL16 ? d->field.data() : vol->density;
        memcpy(field_out, f, sizeof(float) * (size_t)N * N * N);
    }
    return VRT_OK;
}

void vrto_env_lookup(const uint8_t* env_rgba8, int face_size, const float dir[3], float rgb_out[3]) {
    env_lookup(env_rgba8, face_size, v3(dir[0], dir[1], dir[2]), rgb_out);
}

}  // extern "C"

/* ---- reference-algorithm cross-check (double precision) -------------------------------- */

namespace {
inline double cubic(double A, double B, double C, double D, double s) { return ((A * s + B) * s + C) * s + D; }

/* first root of the cubic in [s0,1] given f(s0) > 0, or -1 */
double first_root(double A, double B, double C, double D, double s0) {
    /* split [s0,1] at the derivative's roots, then bisect the first bracketing piece */
    double cuts[4];
    int nc = 0;
    cuts[nc++] = s0;
    double dA = 3.0 * A, dB = 2.0 * B, dC = C;
    double e[2];
    int ne = 0;
    if (fabs(dA) > 1e-300) {
        double disc = dB * dB - 4.0 * dA * dC;
        if (disc >= 0.0) {
            double sq = sqrt(disc);
            e[ne++] = (-dB - sq) / (2.0 * dA);
            e[ne++] = (-dB + sq) / (2.0 * dA);
        }
    } else if (fabs(dB) > 1e-300) {
        e[ne++] = -dC / dB;
    }
    if (ne == 2 && e[0] > e[1]) { double t = e[0]; e[0] = e[1]; e[1] = t; }
    for (int i = 0; i < ne; i++)
        if (e[i] > s0 && e[i] < 1.0) cuts[nc++] = e[i];
    cuts[nc++] = 1.0;
    for (int i = 0; i + 1 < nc; i++) {
        double a = cuts[i], b = cuts[i + 1];
        double fa = cubic(A, B, C, D, a), fb = cubic(A, B, C, D, b);
        if (fa <= 0.0) return a;
        if (fb <= 0.0) {
            for (int it = 0; it < 200; it++) {
                double m = 0.5 * (a + b);
                double fm = cubic(A, B, C, D, m);
                if (fm > 0.0) a = m; else b = m;
            }
            return 0.5 * (a + b);
        }
    }
    return -1.0;
}

/* The reference's hit search for one OBJECT-space ray through one density grid (den: N^3, index x*N*N + z*N + y), shared by
   vrto_ref_hit_t and vrto_ref_render.  Cell-by-cell walk (GoToNextVoxel, Voxel.hlsli:80-128, without the +0.1 nudge and
   without the octree: a merged node holds no surface, so skipping it changes nothing); in every cell with a corner <= 0
   (HasIsoSurfaceInsideCell / IsSolidCell, :497-538) the trilinear interpolant along the ray as a cubic in s in [0,1]
   (GetDensityPolynomial, :552-605) and its first root (GetSurfaceIntersectionT, :691-781: the reference splits at the
   derivative's roots and takes 2 regula-falsi steps + 1 secant; here the bracket is bisected to double precision — the
   zero the reference approximates).  A cell whose interpolant is <= 0 where the ray enters it reports the entry point
   (:704-708).  Double precision throughout. */
struct RefHit {
    double t = 0.0;       /* ray parameter of the hit */
    int c[3] = {0, 0, 0}; /* cell that holds it (x, y, z) */
    bool first = false;   /* the hit is the very first position of the walk (max(tEnter, 0)) */
    bool solid = false;   /* ... and that cell is solid: all 8 corners < 0 (IsSolidCell) */
    double t_enter = 0.0; /* slab entry (negative: the origin is inside the volume) */
};
inline double tapd(const float* den, int N, int x, int y, int z) { return (double)den[((size_t)x * N + (size_t)z) * N + (size_t)y]; }

int ref_walk(const float* den, int N, double E, const double o[3], const double d[3], RefHit& h) {
    const double cell = 2.0 * E / (N - 1);
    /* slab (Ray.hlsli:111-134) */
    double t_enter = -1e300, t_exit = 1e300;
    for (int a = 0; a < 3; a++) {
        if (d[a] != 0.0) {
            double t0 = (-E - o[a]) / d[a], t1 = (E - o[a]) / d[a];
            if (t0 > t1) { double s = t0; t0 = t1; t1 = s; }
            if (t0 > t_enter) t_enter = t0;
            if (t1 < t_exit) t_exit = t1;
        } else if (o[a] < -E || o[a] > E) {
            return 0;
        }
    }
    if (!(t_exit > t_enter) || t_exit < 0.0) return 0;
    h.t_enter = t_enter;
    double t = t_enter > 0.0 ? t_enter : 0.0;
    /* cell walk: at each step find the cell containing the midpoint of the next segment */
    for (int guard = 0; guard < 8 * N; guard++) {
        if (t >= t_exit) return 0;
        double eps = 1e-9 * (1.0 + fabs(t));
        double pm[3];
        int c[3];
        for (int a = 0; a < 3; a++) {
            pm[a] = o[a] + d[a] * (t + eps);
            int ci = (int)floor((pm[a] + E) / cell);
            c[a] = ci < 0 ? 0 : (ci > N - 2 ? N - 2 : ci);
        }
        /* exit of this cell (GoToNextVoxel, Voxel.hlsli:80-128, without the +0.1 nudge) */
        double t_out_cell = t_exit;
        for (int a = 0; a < 3; a++) {
            if (d[a] != 0.0) {
                double face = -E + cell * (c[a] + (d[a] > 0.0 ? 1 : 0));
                double tf = (face - o[a]) / d[a];
                if (tf > t && tf < t_out_cell) t_out_cell = tf;
            }
        }
        if (!(t_out_cell > t)) t_out_cell = t + 1e-7 * (1.0 + fabs(t));
        /* corner values, VCell corner order of GetDensityPolynomial (v1..v8: x fastest, then y, then z) */
        double v[8];
        bool neg = false, pos = false;
        for (int k = 0; k < 8; k++) {
            v[k] = tapd(den, N, c[0] + (k & 1), c[1] + ((k >> 1) & 1), c[2] + ((k >> 2) & 1));
            if (v[k] < 0.0) neg = true;
            if (v[k] > 0.0) pos = true;
            if (v[k] == 0.0) { neg = true; pos = true; }
        }
        if (neg) { /* HasIsoSurfaceInsideCell or solid cell (Voxel.hlsli:497-538) */
            /* a = cell-space position at tIn, b = delta to tOut (GetDensityPolynomial, :552-605) */
            double a1[3], a0[3], b1[3], b0[3];
            for (int a = 0; a < 3; a++) {
                double pin = o[a] + d[a] * t, pout = o[a] + d[a] * t_out_cell;
                double origin_cell = -E + cell * c[a];
                a1[a] = (pin - origin_cell) / cell;
                a0[a] = 1.0 - a1[a];
                b1[a] = (pout - origin_cell) / cell - a1[a];
                b0[a] = -b1[a];
            }
            double A = 0, B = 0, C = 0, D = 0;
            for (int k = 0; k < 8; k++) {
                const double* ax = (k & 1) ? a1 : a0;
                const double* ay = ((k >> 1) & 1) ? a1 : a0;
                const double* az = ((k >> 2) & 1) ? a1 : a0;
                const double* bx = (k & 1) ? b1 : b0;
                const double* by = ((k >> 1) & 1) ? b1 : b0;
                const double* bz = ((k >> 2) & 1) ? b1 : b0;
                A += bx[0] * by[1] * bz[2] * v[k];
                D += ax[0] * ay[1] * az[2] * v[k];
                B += (ax[0] * by[1] * bz[2] + bx[0] * ay[1] * bz[2] + bx[0] * by[1] * az[2]) * v[k];
                C += (bx[0] * ay[1] * az[2] + ax[0] * by[1] * az[2] + ax[0] * ay[1] * bz[2]) * v[k];
            }
            if (!pos || D <= 0.0) { /* start inside: reference reports tIn (:704-708) */
                h.t = t;
                h.c[0] = c[0]; h.c[1] = c[1]; h.c[2] = c[2];
                h.first = guard == 0;
                h.solid = guard == 0 && !pos;
                return 1;
            }
            double s = first_root(A, B, C, D, 0.0);
            if (s >= 0.0) {
                h.t = t + (t_out_cell - t) * s;
                h.c[0] = c[0]; h.c[1] = c[1]; h.c[2] = c[2];
                return 1;
            }
        }
        t = t_out_cell;
    }
    return 0;
}

/* GetDensity (Voxel.hlsli:607-684) of cell c at the cell-space position f, on the texels the shader would Load: a texel
   outside the 3D texture reads as 0 (D3D out-of-bounds Load), density 0. */
inline double ref_density(const float* den, int N, const int c[3], const double f[3]) {
    double p = 0.0;
    for (int k = 0; k < 8; k++) {
        const int i = k & 1, j = (k >> 1) & 1, l = (k >> 2) & 1;
        const int x = c[0] + i, y = c[1] + j, z = c[2] + l;
        const double v = (x < 0 || y < 0 || z < 0 || x >= N || y >= N || z >= N) ? 0.0 : tapd(den, N, x, y, z);
        p += fabs((1 - i) - f[0]) * fabs((1 - j) - f[1]) * fabs((1 - l) - f[2]) * v;
    }
    return p;
}

/*
 * VRIntersection / VRIntersectionShadowRay (SH/Raytracing.hlsl:147-442) for one instance: the reference's OWN hit
 * definition, for vrto_ref_render.  Object-space ray (GetLocalRay, Ray.hlsli:21-29), ref_walk for the hit distance, then
 *   - GetNormal (Voxel.hlsli:783-804) evaluated AT THE ROOT: central difference of GetDensity in the cells c -+ 1 at the
 *     root's own cell-space position, NaN -> (0,0,0), normalised — not at a point some footprints in front of it;
 *   - a ray that enters the volume through a face into a solid cell reports the entry point with the AABB-face normal
 *     (Raytracing.hlsl:198-226, normalised there);
 *   - a hit outside [TMin = 0, RayTCurrent] is rejected by ReportHit and the shader returns: a ray that starts inside the
 *     negative region of a volume does not hit that volume at all.
 * Idealised where the shader's own numerics are not a definition: no +0.01 / +0.1 nudges (they are in units of the
 * reference's un-normalised ray direction), no 255-iteration budget, the exact first root instead of three secant steps.
 * Nothing here depends on the sphere-trace's contract (eps_hit, cone_eps, k_relax, tables, step clamp).
 */
bool ref_march_instance(const Packed& P, int ii, V3 o, V3 d, float t_cur, bool want_normal, float& t_hit, V3& n_world) {
    const Instance& I = P.inst[ii];
    const Volume& V = P.vol[I.slot];
    const V3 oo = mul(I.w2o, o - I.pos);
    const V3 od = mul(I.w2o, d);
    const double od3[3] = {od.x, od.y, od.z}, oo3[3] = {oo.x, oo.y, oo.z};
    const double E = V.extent;
    RefHit h;
    if (!ref_walk(V.den, V.N, E, oo3, od3, h)) return false;
    if (h.t_enter > (double)t_cur) return false;           /* DetermineRayAABBIntersection: tEnter <= TCurrent */
    if (h.first && h.t_enter < 0.0) return false;          /* origin inside the solid: reported t <= 0, rejected */
    if (!(h.t > 0.0) || h.t > (double)t_cur) return false; /* ReportHit's interval */
    t_hit = (float)h.t;
    if (!want_normal) return true;
    double n[3] = {0.0, 0.0, 0.0};
    if (h.solid) {
        const double tb = h.t_enter - 0.1;
        for (int a = 0; a < 3; a++) {
            const double rp = oo3[a] + od3[a] * tb;
            n[a] = rp > E ? 1.0 : (rp < -E ? -1.0 : 0.0);
        }
    } else {
        const double cell = 2.0 * E / (V.N - 1);
        double f[3];
        for (int a = 0; a < 3; a++) f[a] = ((oo3[a] + od3[a] * h.t) - (-E + cell * h.c[a])) / cell;
        for (int a = 0; a < 3; a++) {
            int cp[3] = {h.c[0], h.c[1], h.c[2]}, cm[3] = {h.c[0], h.c[1], h.c[2]};
            cp[a] += 1;
            cm[a] -= 1;
            n[a] = ref_density(V.den, V.N, cp, f) - ref_density(V.den, V.N, cm, f);
        }
    }
    const double l2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    V3 nf = v3(0.0f, 0.0f, 0.0f);
    if (l2 > 0.0) {
        const double inv = 1.0 / sqrt(l2);
        nf = v3((float)(n[0] * inv), (float)(n[1] * inv), (float)(n[2] * inv));
    }
    n_world = mul(I.o2w, nf);
    return true;
}
}  // namespace

extern "C" {

int vrto_ref_hit_t(const vrto_volume* vol, const float origin[3], const float dir[3], double* t_out) {
    const int N = (1 << vol->resolution) + 1;
    const double o[3] = {origin[0], origin[1], origin[2]}, d[3] = {dir[0], dir[1], dir[2]};
    RefHit h;
    if (!ref_walk(vol->density, N, (double)vol->extent, o, d, h)) return 0;
    *t_out = h.t;
    return 1;
}

/* vrto_ref_hit_t for n object-space rays of one volume. */
int vrto_ref_hit_batch(const vrto_volume* vol, int n, const float* origins, const float* dirs, uint8_t* hit_out, double* t_out, int threads) {
    if (!vol || n < 0 || (n > 0 && (!origins || !dirs || !hit_out || !t_out))) return VRT_ERR_INVALID;
    if (threads < 1) threads = 1;
    auto work = [&](int k) {
        for (int i = k; i < n; i += threads) {
            double t = 0.0;
            hit_out[i] = vrto_ref_hit_t(vol, origins + 3 * i, dirs + 3 * i, &t) ? 1 : 0;
            t_out[i] = t;
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < threads; k++) th.emplace_back(work, k);
        for (auto& t : th) t.join();
    }
    return 0;
}


/* The frame the REFERENCE's intersection would shade: every ray (camera, shadow, mirror) is intersected by
   ref_march_instance; camera ray, closest-hit shading, miss and tone-map are the oracle's restatements of the reference's
   (radiance_ray).  Independent of the sphere-trace contract: params' march fields (eps_hit, cone_eps, k_relax, step_*, max_steps)
   are not read.  t_out_or_null (rows*width): the camera ray's hit distance, or -1 for a miss. */
int vrto_ref_render(const vrt_scene* scene, const vrto_volume* volumes, const uint8_t* env_rgba8, int env_face_size,
                    const vrt_params* params, int row0, int rows, float* out_rgba, float* t_out_or_null, int threads) {
    std::unique_ptr<Packed> P(new Packed);
    if (!pack(scene, volumes, env_rgba8, env_face_size, params, *P) || !out_rgba) return VRT_ERR_INVALID;
    if (!(params->mode >= VRT_MODE_INTERP && params->mode < VRT_MODE_CUBE)) return VRT_ERR_UNSUPPORTED;
    if (row0 < 0 || rows < 0 || row0 + rows > params->height) return VRT_ERR_INVALID;
    P->ref_intersection = true;
    if (threads < 1) threads = 1;
    const int W = params->width;
    auto work = [&](int k) {
        Stats st;
        for (int y = row0 + k; y < row0 + rows; y += threads) {
            render_rows(*P, y, y + 1, row0, out_rgba, st);
            if (t_out_or_null)
                for (int x = 0; x < W; x++) {
                    V3 o, d;
                    camera_ray(P->cam, W, params->height, x, y, o, d);
                    HitRec h;
                    uint64_t steps = 0;
                    t_out_or_null[(size_t)(y - row0) * W + x] = trace_closest(*P, o, d, 10000.0f, 0.0f, h, steps) ? h.t : -1.0f;
                }
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < threads; k++) th.emplace_back(work, k);
        for (auto& t : th) t.join();
    }
    return VRT_OK;
}


/* The frame the reference's shaders compute, LITERALLY (vrt_ref_literal.inl): fp32, the un-normalised camera direction with every
   offset and the shading's wo in its units, +0.01 / +0.1 nudges, the collapsed octree's leaves, the cubic on [cellEnter, cellExit]
   with 2 regula-falsi steps + 1 secant, abs()-weighted GetNormal with out-of-bounds texels = 0, 255 leaves then the red hit.
   Volumes are read through the reference's 16-bit texel whatever vrto_volume::format says (that is all its GPU ever sees).
   t_out_or_null: the camera ray's hit distance in WORLD units (t * |direction|), -1 = miss.  Interp modes only. */
int vrto_ref_literal_render(const vrt_scene* scene, const vrto_volume* volumes, const uint8_t* env_rgba8, int env_face_size,
                            const vrt_params* params, int row0, int rows, float* out_rgba, float* t_out_or_null, unsigned options,
                            vrto_literal_stats* stats_or_null, int threads) {
    std::unique_ptr<Packed> P(new Packed);
    if (!params || !(params->mode >= VRT_MODE_INTERP && params->mode < VRT_MODE_CUBE)) return VRT_ERR_UNSUPPORTED;
    if (!pack(scene, volumes, env_rgba8, env_face_size, params, *P, true) || !out_rgba) return VRT_ERR_INVALID;
    if (row0 < 0 || rows < 0 || row0 + rows > params->height) return VRT_ERR_INVALID;
    P->lit_options = options;
    if (threads < 1) threads = 1;
    const int W = params->width;
    std::vector<LitCounters> counters((size_t)threads);
    auto work = [&](int k) {
        Stats st;
        g_lit = LitCounters();
        for (int y = row0 + k; y < row0 + rows; y += threads) {
            render_rows(*P, y, y + 1, row0, out_rgba, st);
            if (t_out_or_null) {
                const LitCounters keep = g_lit; /* the distance probe below repeats the camera rays: not counted twice */
                for (int x = 0; x < W; x++) {
                    V3 o, d;
                    camera_ray(P->cam, W, params->height, x, y, o, d);
                    if (!(options & VRTO_LIT_NORMALISED_CAMERA)) camera_ray_raw(P->cam, W, params->height, x, y, o, d);
                    HitRec h;
                    t_out_or_null[(size_t)(y - row0) * W + x] = lit_trace_closest(*P, o, d, 10000.0f, h) ? h.t * sqrtf(dot(d, d)) : -1.0f;
                }
                g_lit = keep;
            }
        }
        counters[(size_t)k] = g_lit;
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int k = 0; k < threads; k++) th.emplace_back(work, k);
        for (auto& t : th) t.join();
    }
    if (stats_or_null) {
        vrto_literal_stats s;
        memset(&s, 0, sizeof s);
        for (const LitCounters& c : counters) {
            s.rays += c.rays;
            s.iterations += c.iterations;
            s.solid_start_hits += c.solid_start;
            s.entry_hits += c.entry_hits;
            s.root_hits += c.root_hits;
            s.tail_hits += c.tail_hits;
            s.red_hits += c.red_hits;
            s.rejected_reports += c.rejected;
        }
        *stats_or_null = s;
    }
    return VRT_OK;
}

/* The collapsed octree the reference builds for `vol` (VCellOctree, Voxel/Private/Octree.cpp): node count, texture edge S
   (the traversal texture is (2S)^3 RGBA8 texels whose pointer texels hold block coordinates 2c < 2S in 8 bits), leaves per depth. */
int vrto_literal_octree_info(const vrto_volume* vol, vrto_octree_info* out) {
    if (!vol || !vol->density || !out || vol->resolution < 1 || vol->resolution > 8) return VRT_ERR_INVALID;
    const int N = (1 << vol->resolution) + 1, nb = (N - 1 + 3) / 4;
    std::shared_ptr<const Derived> d = derive(*vol, N, nb, false, true);
    memset(out, 0, sizeof *out);
    out->nodes = d->lit->nodes;
    out->texture_edge = 2 * d->lit->S;
    out->pointer_overflow = 2 * (d->lit->S - 1) > 255 ? 1 : 0;
    for (int k = 0; k < 9; k++) out->leaves_at_depth[k] = d->lit->leaves_at[k];
    return VRT_OK;
}

}  // extern "C"

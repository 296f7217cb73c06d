/*
 * vrt_ref_literal.inl — the reference's intersection shaders restated LITERALLY (included by vrt_oracle.cpp inside its
 * anonymous namespace).  TEST INFRASTRUCTURE ONLY, CPU only: a measuring instrument, never a product path.
 *
 * vrto_ref_render (ref_march_instance) is an IDEALISED restatement: double precision, the exact first root, no nudges, no
 * octree, no budget.  This file is the other one: what VRIntersection / VRIntersectionShadowRay compute, statement by
 * statement, in fp32, so that the distance between "the reference's frames" and the idealisation — and between the HIP
 * frames and either — is a measured number (tests/ref_pixels.py):
 *
 *   lit_slab               DetermineRayAABBIntersection            SH/Include/Ray.hlsli:111-134
 *   lit_w2v / lit_v2w      WorldSpaceToVoxelSpace / VoxelIndexToWorldSpace / WorldSpaceToBottomLevelCellSpace / IsValidCell
 *                                                                  SH/Include/Voxel.hlsli:21-61
 *   build_literal          VDXVoxelVolume::EncodeVoxel + DecodeDensity (the texel the shader reads)
 *                                                                  Renderer/DX/Private/RDXVoxelVolume.cpp:399-421, Voxel.hlsli:254-270
 *                          VCellOctree: leaves per cell, bottom-up build, CollapseTree / TryToMergeNodes (a node whose cells
 *                          all lack a surface becomes ONE leaf), node count and texture edge S
 *                                                                  Voxel/Private/Octree.cpp:70-107,181-262,548-580, Voxel.cpp:17-41
 *   lit_octree_node        GetOctreeNode: leaf origin (VoxelIndexToWorldSpace of the leaf's first cell) and size
 *                          pow(2, maxDepth - depth) * distanceBtwVoxels, root-leaf shortcut, invalid cell -> {0, (-1,-1,-1)}
 *                                                                  Voxel.hlsli:293-495 (texture layout RDXVoxelVolume.cpp:221-292)
 *   lit_cell_exit          CalculateCellExit / GoToNextVoxel (+0.1)  Voxel.hlsli:80-128,185-230
 *   lit_has_iso/lit_solid  HasIsoSurfaceInsideCell / IsSolidCell    Voxel.hlsli:497-538
 *   lit_polynomial         GetDensityPolynomial                     Voxel.hlsli:552-605
 *   lit_surface_t          GetSurfaceIntersectionT (t0 = max(0, -tIn/(tOut-tIn)), split at the derivative's roots,
 *                          2 regula-falsi steps + 1 secant, tHit > 0)  Voxel.hlsli:686-781
 *   lit_density/lit_normal GetDensity with abs() weights (texels outside the 3D texture read 0) / GetNormal
 *                                                                  Voxel.hlsli:607-684,783-804
 *   lit_march_instance     VRIntersection / VRIntersectionShadowRay: tEnter += 0.01, origin-inside start (ReverseRay returns
 *                          its argument: the "backwards" exit is the forward one, negated), solid start cell -> AABB-face
 *                          normal, 255 leaves, then the red unlit hit at t = 10; ReportHit accepts TMin = 0 <= t <= RayTCurrent
 *                                                                  SH/Raytracing.hlsl:147-442
 *
 * The texture walk itself is not replayed texel by texel: GetOctreeNode's result is a pure function of (cell, depth of its
 * leaf), which build_literal computes with the merge rule; what the walk could add are its failure modes, which are
 * reported instead (lit_info: pointer texels are 8 bits, a texture edge 2S > 256 wraps; the walk stops at depth 8).
 * The voxel material is (density <= 0) in both of the reference's producers (Voxelizer/Private/VolumeConverter.cpp:236,245,
 * App/Private/RendererEngineInstance.cpp:300), so VCell::HasSurface reduces to "the 8 corner signs differ".
 * fp32, no contraction; operations in the shader's order.  What a D3D driver may still do differently (fused multiply-adds,
 * 1-ulp divisions) is below anything an 8-bit target keeps.  PARITY UNPINNED like everything else here: the reference holds
 * no frame to check this restatement against.
 */

struct LitVolume {
    int r = 0, N = 0;
    std::vector<float> dec;          /* DecodeDensity(EncodeVoxel(sample)) */
    std::vector<uint8_t> leaf_depth; /* per cell [(x*C + z)*C + y]: depth of the collapsed octree's leaf that holds it */
    uint64_t nodes = 0;              /* nodes of the collapsed tree (GetAllNodes) */
    uint64_t leaves_at[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int S = 0;                       /* ceil(cbrtf(nodes)): the traversal texture is (2S)^3 texels */
};

struct LitCounters {
    uint64_t rays = 0, iterations = 0;
    uint64_t solid_start = 0; /* accepted hits of a solid start cell (AABB-face normal) */
    uint64_t entry_hits = 0;  /* accepted hits reported at the interval's START: the cubic is <= 0 where the search begins (Voxel.hlsli:707-711) */
    uint64_t root_hits = 0;   /* accepted hits at a root of the cubic */
    uint64_t tail_hits = 0;   /* ... of which beyond the leaf's true exit, in the 0.1 the interval overhangs into the next cell */
    uint64_t red_hits = 0;    /* accepted budget-exhaustion hits (unlit red at t = 10, Raytracing.hlsl:325-334) */
    uint64_t rejected = 0;    /* ReportHit outside [0, RayTCurrent]: the shader returns without a hit */
};
thread_local LitCounters g_lit;

inline int lit_sign(float v) { return (0.0f < v) - (v < 0.0f); } /* HLSL sign(); NaN -> 0 */

/* float -> int the way a GPU converts: NaN -> 0, saturating */
inline int lit_ftoi(float f) {
    if (!(f == f)) return 0;
    if (f >= 2147483520.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

void build_literal(const float* den, int N, int r, LitVolume& L) {
    const int C = N - 1;
    const size_t count = (size_t)N * N * N;
    L.r = r;
    L.N = N;
    L.dec.resize(count);
    for (size_t i = 0; i < count; i++) {
        const float d = den[i];
        const float q = fabsf(texel16_value(d)); /* (uint16_t)(|d| * 100) & 0x7fff */
        const float res = q * 0.01f;
        L.dec[i] = d < 0.0f ? -res : res; /* sign bit from the float, also when q == 0: -0.0 */
    }
    /* mergeable[d]: the node (x, y, z) of depth d holds no cell with a surface */
    std::vector<std::vector<uint8_t>> m((size_t)r + 1);
    m[(size_t)r].resize((size_t)C * C * C);
    parallel_slabs(C, [&](int x0, int x1) {
        for (int x = x0; x < x1; x++)
            for (int z = 0; z < C; z++)
                for (int y = 0; y < C; y++) {
                    const int s0 = lit_sign(den[((size_t)x * N + z) * N + y]);
                    bool same = true;
                    for (int k = 1; k < 8 && same; k++)
                        same = lit_sign(den[((size_t)(x + (k & 1)) * N + (z + ((k >> 2) & 1))) * N + (y + ((k >> 1) & 1))]) == s0;
                    m[(size_t)r][((size_t)x * C + z) * C + y] = same ? 1 : 0;
                }
    });
    for (int d = r - 1; d >= 0; d--) {
        const int n = 1 << d, n2 = n * 2;
        m[(size_t)d].resize((size_t)n * n * n);
        for (int x = 0; x < n; x++)
            for (int z = 0; z < n; z++)
                for (int y = 0; y < n; y++) {
                    uint8_t all = 1;
                    for (int k = 0; k < 8; k++)
                        all &= m[(size_t)d + 1][((size_t)(2 * x + (k & 1)) * n2 + (2 * z + ((k >> 2) & 1))) * n2 + (2 * y + ((k >> 1) & 1))];
                    m[(size_t)d][((size_t)x * n + z) * n + y] = all;
                }
    }
    uint64_t branches = 0;
    for (int d = 0; d < r; d++) {
        /* a branch: not mergeable itself, and no ancestor merged (a mergeable ancestor implies a mergeable node) */
        for (uint8_t v : m[(size_t)d]) branches += v ? 0 : 1;
    }
    L.nodes = 1 + 8 * branches;
    L.S = (int)std::ceil(cbrtf((float)L.nodes));
    L.leaf_depth.resize((size_t)C * C * C);
    uint64_t cells_at[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int x = 0; x < C; x++)
        for (int z = 0; z < C; z++)
            for (int y = 0; y < C; y++) {
                int dd = r;
                for (int d = 0; d < r; d++) {
                    const int sh = r - d, n = 1 << d;
                    if (m[(size_t)d][((size_t)(x >> sh) * n + (z >> sh)) * n + (y >> sh)]) {
                        dd = d;
                        break;
                    }
                }
                L.leaf_depth[((size_t)x * C + z) * C + y] = (uint8_t)dd;
                if (dd < 10) cells_at[dd]++;
            }
    for (int d = 0; d <= r && d < 10; d++) L.leaves_at[d] = cells_at[d] >> (3 * (r - d)); /* a leaf of depth d holds 8^(r-d) cells */
}

void ensure_literal(Derived& d, const float* den, int N, int resolution) {
    auto L = std::make_shared<LitVolume>();
    build_literal(den, N, resolution, *L);
    d.lit = L;
}

struct LitNode {
    float size;
    V3 pos;
};

inline bool lit_valid_cell(const Volume& V, const int c[3]) {
    return c[0] >= 0 && c[1] >= 0 && c[2] >= 0 && (c[0] + 1) < V.N && (c[1] + 1) < V.N && (c[2] + 1) < V.N;
}

inline void lit_w2v(const Volume& V, V3 p, int c[3]) {
    const float org = -V.extent;
    c[0] = lit_ftoi(floorf((p.x - org) / V.cell));
    c[1] = lit_ftoi(floorf((p.y - org) / V.cell));
    c[2] = lit_ftoi(floorf((p.z - org) / V.cell));
}

inline V3 lit_v2w(const Volume& V, const int c[3]) {
    const float org = -V.extent;
    return v3((float)c[0] * V.cell + org, (float)c[1] * V.cell + org, (float)c[2] * V.cell + org);
}

inline V3 lit_cell_space(const Volume& V, const int c[3], float cell_size, V3 world) {
    const V3 vp = lit_v2w(V, c);
    return v3((world.x - vp.x) / cell_size, (world.y - vp.y) / cell_size, (world.z - vp.z) / cell_size);
}

inline V3 lit_pos(V3 o, V3 d, float t) { return v3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t); }

LitNode lit_octree_node(const Volume& V, const LitVolume& L, const int c[3]) {
    LitNode n = {0.0f, v3(-1.0f, -1.0f, -1.0f)};
    if (!lit_valid_cell(V, c)) return n;
    const int C = V.N - 1;
    const int dd = L.leaf_depth[((size_t)c[0] * C + (size_t)c[2]) * C + (size_t)c[1]];
    if (dd > 8) return n; /* the walk's loop ends at depth 8 (Voxel.hlsli:316) */
    n.size = powf(2.0f, (float)(L.r - dd)) * V.cell; /* GetNodeSize */
    if (dd == 0) {
        const float h = n.size * -0.5f; /* root leaf: float3(1,1,1) * size * -0.5 (Voxel.hlsli:309-314) */
        n.pos = v3(h, h, h);
        return n;
    }
    const int mask = ~((1 << (L.r - dd)) - 1);
    const int first[3] = {c[0] & mask, c[1] & mask, c[2] & mask}; /* the leaf's CellIndex: its first cell */
    n.pos = lit_v2w(V, first);
    return n;
}

inline float lit_tap(const LitVolume& L, int x, int y, int z) {
    const int N = L.N;
    if (x < 0 || y < 0 || z < 0 || x >= N || y >= N || z >= N) return 0.0f; /* Load outside the texture */
    return L.dec[((size_t)x * N + (size_t)z) * N + (size_t)y];
}

inline void lit_corners(const LitVolume& L, const int c[3], float v[8]) {
    for (int k = 0; k < 8; k++) v[k] = lit_tap(L, c[0] + (k & 1), c[1] + ((k >> 1) & 1), c[2] + ((k >> 2) & 1));
}

inline bool lit_has_iso(const LitVolume& L, const int c[3]) {
    float v[8];
    lit_corners(L, c, v);
    const int s = lit_sign(v[0]);
    for (int k = 1; k < 8; k++)
        if (lit_sign(v[k]) != s) return true;
    return false;
}

inline bool lit_solid(const LitVolume& L, const int c[3]) {
    float v[8];
    lit_corners(L, c, v);
    for (int k = 0; k < 8; k++)
        if (!(v[k] < 0.0f)) return false;
    return true;
}

/* exit of the box [pos, pos + size] along the ray: CalculateCellExit; GoToNextVoxel adds 0.1 to it */
inline float lit_cell_exit(V3 o, V3 d, V3 pos, float size) {
    const float inf = std::numeric_limits<float>::infinity();
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, lo[3] = {pos.x, pos.y, pos.z};
    float tmax[3] = {100000.0f, 100000.0f, 100000.0f};
    for (int a = 0; a < 3; a++) {
        const float inv = dd[a] != 0.0f ? 1.0f / dd[a] : (dd[a] > 0.0f ? inf : -inf);
        const float face = dd[a] > 0.0f ? lo[a] + size : lo[a];
        if (dd[a] != 0.0f) tmax[a] = (face - oo[a]) * inv;
    }
    if (tmax[0] < tmax[1]) return tmax[0] < tmax[2] ? tmax[0] : tmax[2];
    return tmax[1] < tmax[2] ? tmax[1] : tmax[2];
}

inline float lit_poly(float t, float A, float B, float C, float D) { return ((((A * t) * t) * t + (B * t) * t) + C * t) + D; }

void lit_polynomial(const Volume& V, const LitVolume& L, V3 o, V3 d, const int c[3], float cell_size, float t_in, float t_out,
                    float& A, float& B, float& C, float& D) {
    const V3 a1v = lit_cell_space(V, c, cell_size, lit_pos(o, d, t_in));
    const V3 outv = lit_cell_space(V, c, cell_size, lit_pos(o, d, t_out));
    const float a1[3] = {a1v.x, a1v.y, a1v.z};
    float a0[3], b1[3], b0[3];
    const float ov[3] = {outv.x, outv.y, outv.z};
    for (int a = 0; a < 3; a++) {
        a0[a] = 1.0f - a1[a];
        b1[a] = ov[a] - a1[a];
        b0[a] = -b1[a];
    }
    float v[8];
    lit_corners(L, c, v);
    A = B = C = D = 0.0f;
    for (int k = 0; k < 8; k++) {
        const float ax = (k & 1) ? a1[0] : a0[0], ay = ((k >> 1) & 1) ? a1[1] : a0[1], az = ((k >> 2) & 1) ? a1[2] : a0[2];
        const float bx = (k & 1) ? b1[0] : b0[0], by = ((k >> 1) & 1) ? b1[1] : b0[1], bz = ((k >> 2) & 1) ? b1[2] : b0[2];
        const float ta = ((bx * by) * bz) * v[k];
        const float td = ((ax * ay) * az) * v[k];
        const float tb = ((((ax * by) * bz) + ((bx * ay) * bz)) + ((bx * by) * az)) * v[k];
        const float tc = ((((bx * ay) * az) + ((ax * by) * az)) + ((ax * ay) * bz)) * v[k];
        A = k ? A + ta : ta;
        D = k ? D + td : td;
        B = k ? B + tb : tb;
        C = k ? C + tc : tc;
    }
}

/* GetSurfaceIntersectionT.  at_start: the hit is tIn itself (the cubic is <= 0 where the search starts). */
bool lit_surface_t(const Volume& V, const LitVolume& L, V3 o, V3 d, const int c[3], float cell_size, float t_in, float t_out,
                   float& t_hit, bool& at_start) {
    float A, B, C, D;
    float t0 = fmaxf(0.0f, -t_in / (t_out - t_in));
    float t1 = 1.0f;
    at_start = false;
    lit_polynomial(V, L, o, d, c, cell_size, t_in, t_out, A, B, C, D);
    const float dA = 3.0f * A, dB = 2.0f * B;
    float ex1 = (-dB + sqrtf(dB * dB - (4.0f * dA) * C)) / (2.0f * dA);
    float ex2 = (-dB - sqrtf(dB * dB - (4.0f * dA) * C)) / (2.0f * dA);
    float f0 = lit_poly(t0, A, B, C, D);
    if (lit_sign(f0) <= 0) {
        t_hit = t_in;
        at_start = true;
        return true;
    }
    float f1 = lit_poly(t1, A, B, C, D);
    if (ex1 > ex2) {
        const float f = ex1;
        ex1 = ex2;
        ex2 = f;
    }
    if (ex1 >= t0 && ex1 <= t1) {
        const float fe = lit_poly(ex1, A, B, C, D);
        if (lit_sign(fe) == lit_sign(f0)) { t0 = ex1; f0 = fe; } else { t1 = ex1; f1 = fe; }
    }
    if (ex2 >= t0 && ex2 <= t1) {
        const float fe = lit_poly(ex2, A, B, C, D);
        if (lit_sign(fe) == lit_sign(f0)) { t0 = ex2; f0 = fe; } else { t1 = ex2; f1 = fe; }
    }
    if (lit_sign(f0) == lit_sign(f1)) return false;
    for (int i = 0; i < 2; i++) {
        const float t = t0 + (t1 - t0) * (-f0 / (f1 - f0));
        const float f = lit_poly(t, A, B, C, D);
        if (lit_sign(f) == lit_sign(f0)) { t0 = t; f0 = f; } else { t1 = t; f1 = f; }
    }
    t_hit = t0 + (t1 - t0) * (-f0 / (f1 - f0));
    t_hit = t_in + t_hit * (t_out - t_in); /* lerp(tIn, tOut, tHit) */
    return t_hit > 0.0f;
}

inline float lit_density(const LitVolume& L, const int c[3], V3 f) {
    float v[8];
    lit_corners(L, c, v);
    float p = 0.0f;
    for (int k = 0; k < 8; k++) {
        const float u = fabsf((float)(1 - (k & 1)) - f.x);
        const float w2 = fabsf((float)(1 - ((k >> 1) & 1)) - f.y);
        const float w3 = fabsf((float)(1 - ((k >> 2) & 1)) - f.z);
        p += ((u * w2) * w3) * v[k];
    }
    return p;
}

inline V3 lit_normal(const LitVolume& L, const int c[3], V3 f) {
    float n[3];
    for (int a = 0; a < 3; a++) {
        int cp[3] = {c[0], c[1], c[2]}, cm[3] = {c[0], c[1], c[2]};
        cp[a] += 1;
        cm[a] -= 1;
        n[a] = lit_density(L, cp, f) - lit_density(L, cm, f);
    }
    if (n[0] != n[0] || n[1] != n[1] || n[2] != n[2]) return v3(0.0f, 0.0f, 0.0f);
    const float inv = 1.0f / sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]); /* normalize(): a zero gradient becomes NaN */
    return v3(n[0] * inv, n[1] * inv, n[2] * inv);
}

/* VRIntersection (shadow = false) / VRIntersectionShadowRay (shadow = true) for one instance.  Returns true when a hit was
   ACCEPTED (ReportHit: 0 <= t <= t_cur).  n_obj: attr.normal (object space; the colour itself when unlit). */
bool lit_march_instance(const Packed& P, int ii, V3 o, V3 d, float t_cur, bool shadow, float& t_hit, V3& n_obj, bool& unlit) {
    const Instance& I = P.inst[ii];
    const Volume& V = P.vol[I.slot];
    const LitVolume& L = *V.lit;
    const float inf = std::numeric_limits<float>::infinity();
    const V3 oo = mul(I.w2o, o - I.pos);
    const V3 od = mul(I.w2o, d);
    g_lit.rays++;
    unlit = shadow;
    n_obj = v3(0.0f, 0.0f, 0.0f);
    /* DetermineRayAABBIntersection */
    float t_enter, t_exit;
    {
        const float oa[3] = {oo.x, oo.y, oo.z}, da[3] = {od.x, od.y, od.z};
        float tmin[3], tmax[3];
        for (int a = 0; a < 3; a++) {
            const bool pos = da[a] > 0.0f;
            const float inv = da[a] != 0.0f ? 1.0f / da[a] : (pos ? inf : -inf);
            tmin[a] = ((pos ? -V.extent : V.extent) - oa[a]) * inv;
            tmax[a] = ((pos ? V.extent : -V.extent) - oa[a]) * inv;
        }
        t_enter = fmaxf(fmaxf(tmin[0], tmin[1]), tmin[2]);
        t_exit = fminf(fminf(tmax[0], tmax[1]), tmax[2]);
        if (!(t_exit > t_enter && t_exit >= 0.0f && t_enter <= t_cur)) return false;
    }
    auto report = [&](float t) {
        if (t >= 0.0f && t <= t_cur) {
            t_hit = t;
            return true;
        }
        g_lit.rejected++;
        return false;
    };
    int cur[3], next[3];
    float cell_exit, cell_enter;
    LitNode node;
    if (t_enter >= 0.0f) {
        t_enter += 0.01f;
        lit_w2v(V, lit_pos(oo, od, t_enter), cur);
        cell_exit = t_enter;
        node = lit_octree_node(V, L, cur);
    } else {
        lit_w2v(V, oo, cur);
        node = lit_octree_node(V, L, cur);
        cell_exit = lit_cell_exit(oo, od, node.pos, node.size); /* ReverseRay returns its argument (Ray.hlsli:50-58) */
        cell_exit = -cell_exit;
        cell_exit += 0.01f;
    }
    if (lit_valid_cell(V, cur) && lit_solid(L, cur)) {
        if (!shadow) {
            const V3 rp = lit_pos(oo, od, t_enter - 0.1f);
            const float e = V.extent;
            const float r3[3] = {rp.x, rp.y, rp.z};
            float n[3];
            for (int a = 0; a < 3; a++) {
                n[a] = (float)lit_sign(r3[a] - e);
                if (n[a] < 0.0f) n[a] = r3[a] < -e ? -1.0f : 0.0f;
            }
            const float inv = 1.0f / sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
            n_obj = v3(n[0] * inv, n[1] * inv, n[2] * inv);
            unlit = false;
        }
        if (report(t_enter)) {
            g_lit.solid_start++;
            return true;
        }
        return false;
    }
    int it = 255;
    for (; it > 0; it--) {
        if (cell_exit > t_exit) break;
        g_lit.iterations++;
        cell_enter = cell_exit;
        if (!lit_valid_cell(V, cur)) return false;
        const float true_exit = lit_cell_exit(oo, od, node.pos, node.size);
        cell_exit = true_exit + 0.1f; /* GoToNextVoxel */
        lit_w2v(V, lit_pos(oo, od, cell_exit), next);
        if (lit_has_iso(L, cur)) {
            float th;
            bool at_start;
            if (lit_surface_t(V, L, oo, od, cur, node.size, cell_enter, cell_exit, th, at_start)) {
                if (!shadow) {
                    n_obj = lit_normal(L, cur, lit_cell_space(V, cur, node.size, lit_pos(oo, od, th)));
                    unlit = false;
                }
                if (report(th)) {
                    if (at_start) g_lit.entry_hits++;
                    else {
                        g_lit.root_hits++;
                        if (th > true_exit) g_lit.tail_hits++;
                    }
                    return true;
                }
                return false;
            }
        }
        cur[0] = next[0]; cur[1] = next[1]; cur[2] = next[2];
        node = lit_octree_node(V, L, next);
    }
    if (it <= 0) {
        n_obj = shadow ? v3(0.0f, 0.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
        unlit = true;
        if (report(10.0f)) {
            g_lit.red_hits++;
            return true;
        }
    }
    return false;
}

/* TraceRay over the instances in index order; RayTCurrent is the closest accepted hit so far (DXR leaves the order open;
   the closest hit does not depend on it except through the t = 10 budget hit). */
bool lit_trace_closest(const Packed& P, V3 o, V3 d, float t_max, HitRec& h) {
    bool any = false;
    float best = t_max;
    for (int i = 0; i < P.n_inst; i++) {
        float t;
        V3 n;
        bool unlit;
        if (lit_march_instance(P, i, o, d, best, false, t, n, unlit)) {
            any = true;
            best = t;
            h.t = t;
            h.inst = i;
            h.unlit = unlit;
            h.n_world = unlit ? n : mul(P.inst[i].o2w, n); /* unlit: the colour is attr.normal itself (Raytracing.hlsl:44-48) */
        }
    }
    return any;
}

bool lit_trace_any(const Packed& P, V3 o, V3 d, float t_max) {
    for (int i = 0; i < P.n_inst; i++) {
        float t;
        V3 n;
        bool unlit;
        if (lit_march_instance(P, i, o, d, t_max, true, t, n, unlit)) return true;
    }
    return false;
}

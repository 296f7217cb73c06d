#include "VolumeConverter.h"

#include <cmath>
#include <iostream>

namespace VolumeRaytracer {
namespace Voxelizer {

namespace {

/* Everything about one triangle that the per-voxel classification needs. */
struct TriangleFrame {
    VVector v[3];       /* V1, V2, V3 */
    VVector normal;     /* A: unit face normal */
    VVector along[3];   /* unit edge directions  B: V1→V3, C: V3→V2, D: V2→V1 */
    float length[3];    /* |B|, |C|, |D| */
    VVector inward[3];  /* in-plane unit normals of the edges, pointing into the triangle: E (of B), F (of C), G (of D) */
};

bool make_frame(const VVector& v1, const VVector& v2, const VVector& v3, TriangleFrame& t) {
    t.v[0] = v1;
    t.v[1] = v2;
    t.v[2] = v3;
    VVector n = VVector::Cross(v2 - v1, v3 - v1);
    const float area2 = n.Length();
    if (!(area2 > 0.f)) return false; /* degenerate: the reference would hit its unreachable assert (:779) */
    t.normal = n / area2;
    const VVector e[3] = {v3 - v1, v2 - v3, v1 - v2};
    for (int k = 0; k < 3; k++) {
        t.length[k] = e[k].Length();
        if (!(t.length[k] > 0.f)) return false;
        t.along[k] = e[k] / t.length[k];
        t.inward[k] = VVector::Cross(t.along[k], t.normal).GetNormalized();
    }
    return true;
}

/* Distance from p to the triangle, by the region p projects into (face, 3 edges, 3 vertices). */
float region_distance(const TriangleFrame& t, const VVector& p) {
    const VVector r1 = p - t.v[0], r2 = p - t.v[1], r3 = p - t.v[2];
    const float a = r1.Dot(t.normal);                              /* signed plane distance */
    const float b = r1.Dot(t.along[0]), e = r1.Dot(t.inward[0]);   /* edge V1→V3 */
    const float c = r3.Dot(t.along[1]), f = r3.Dot(t.inward[1]);   /* edge V3→V2 */
    const float d = r2.Dot(t.along[2]), g = r2.Dot(t.inward[2]);   /* edge V2→V1 */
    if (e >= 0.f && f >= 0.f && g >= 0.f) return std::fabs(a);                       /* R1: over the face */
    if (d >= t.length[2] && b <= 0.f) return r1.Length();                            /* R5: vertex V1 */
    if (b >= t.length[0] && c <= 0.f) return r3.Length();                            /* R7: vertex V3 */
    if (c >= t.length[1] && d <= 0.f) return r2.Length();                            /* R6: vertex V2 */
    if (g <= 0.f && d >= 0.f && d <= t.length[2]) return std::sqrt(a * a + g * g);   /* R2: edge V2→V1 */
    if (e <= 0.f && b >= 0.f && b <= t.length[0]) return std::sqrt(a * a + e * e);   /* R4: edge V1→V3 */
    if (f <= 0.f && c >= 0.f && c <= t.length[1]) return std::sqrt(a * a + f * f);   /* R3: edge V3→V2 */
    /* numerically between regions (the reference asserts here): nearest of the three vertices */
    return std::fmin(r1.Length(), std::fmin(r2.Length(), r3.Length()));
}

void voxelize_face(Voxel::VVoxelVolume& volume, const TriangleFrame& t, float threshold) {
    /* index box: triangle bounds, grown by the threshold, rounded to voxels, grown by one voxel
       (GetTriangleBoundingBox + GetVoxelizedBoundingBox, VolumeConverter.cpp:681-701) */
    const VVector lo = VVector::Min(t.v[0], VVector::Min(t.v[1], t.v[2]));
    const VVector hi = VVector::Max(t.v[0], VVector::Max(t.v[1], t.v[2]));
    const VVector half = (hi - lo) * 0.5f;
    const VAABB box(half + lo, half.Abs());
    VIntVector imin = volume.RelativePositionToVoxelIndex(box.GetMin() - VVector::ONE * threshold) - VIntVector(1, 1, 1);
    VIntVector imax = volume.RelativePositionToVoxelIndex(box.GetMax() + VVector::ONE * threshold) + VIntVector(1, 1, 1);
    const int last = (int)volume.GetSize() - 1;
    imin = VIntVector(std::max(imin.X, 0), std::max(imin.Y, 0), std::max(imin.Z, 0));
    imax = VIntVector(std::min(imax.X, last), std::min(imax.Y, last), std::min(imax.Z, last));
    for (int x = imin.X; x <= imax.X; x++)
        for (int y = imin.Y; y <= imax.Y; y++)
            for (int z = imin.Z; z <= imax.Z; z++) {
                const VIntVector idx(x, y, z);
                const float dist = region_distance(t, volume.VoxelIndexToRelativePosition(idx));
                float density = 1.f - (dist / threshold);
                density = -1.f * density + 0.5f;
                Voxel::VVoxel voxel = volume.GetVoxel(idx);
                if (density < voxel.Density) {
                    voxel.Density = density;
                    voxel.Material = voxel.Density <= 0.f ? 1 : 0;
                    volume.SetVoxel(idx, voxel);
                }
            }
}

}  // namespace

bool VVolumeConverter::ExtractResolutionFromName(const std::string& name, uint8_t& outResolution) {
    const size_t at = name.rfind('_');
    if (at == std::string::npos) return false;
    try {
        outResolution = (uint8_t)std::stoi(name.substr(at + 1));
        return true;
    } catch (...) {
        return false;
    }
}

float VVolumeConverter::ExtractionThreshold(const Voxel::VVoxelVolume& volume) { return volume.GetCellSize() * std::sqrt(3.f); }

std::shared_ptr<Voxel::VVoxelVolume> VVolumeConverter::ConvertMeshInfoToVoxelVolume(const VMeshInfo& meshInfo, const VTextureLibrary& textureLib) {
    const VVector be = meshInfo.Bounds.GetExtends();
    float extends = std::fmax(be.X, std::fmax(be.Y, be.Z));
    extends += extends * 0.25f;

    uint8_t resolution = 5;
    if (!ExtractResolutionFromName(meshInfo.MeshName, resolution)) {
        resolution = 5;
        std::cout << "[WARNING] Mesh with name " << meshInfo.MeshName
                  << " has no or invalid resolution specifier. Correct syntax is meshName_resolution (cubeMesh_6). Using default resolution of 5!" << std::endl;
    }
    if (resolution > 8) {
        std::cout << "[WARNING] Mesh with name " << meshInfo.MeshName << " has invalid resolution. Resolution needs to be between or equal than 0 and 8." << std::endl;
        resolution = 5;
    }

    auto volume = std::make_shared<Voxel::VVoxelVolume>(resolution, extends);
    Voxel::VVoxel background;
    background.Material = 0;
    background.Density = extends * 2.f;
    volume->FillVolume(background);

    const float threshold = ExtractionThreshold(*volume);
    size_t skipped = 0;
    for (size_t i = 0; i + 3 <= meshInfo.Indices.size(); i += 3) {
        const size_t a = meshInfo.Indices[i], b = meshInfo.Indices[i + 1], c = meshInfo.Indices[i + 2];
        if (a >= meshInfo.Vertices.size() || b >= meshInfo.Vertices.size() || c >= meshInfo.Vertices.size()) {
            skipped++;
            continue;
        }
        TriangleFrame t;
        if (!make_frame(meshInfo.Vertices[a].Position, meshInfo.Vertices[b].Position, meshInfo.Vertices[c].Position, t)) {
            skipped++;
            continue;
        }
        voxelize_face(*volume, t, threshold);
    }
    if (skipped) std::cout << "[WARNING] Skipped " << skipped << " degenerate or out-of-range triangle(s) of " << meshInfo.MeshName << std::endl;

    VMaterial material = meshInfo.Material;
    auto it = textureLib.Materials.find(meshInfo.MaterialName);
    if (it != textureLib.Materials.end()) {
        material.AlbedoTexturePath = it->second.Albedo;
        material.NormalTexturePath = it->second.Normal;
        material.RMTexturePath = it->second.RM;
        material.TextureScale = it->second.TextureTiling;
    }
    volume->SetMaterial(material);
    /* metric for the sphere-trace: density·thr is the distance to the shell wherever it is below
       thr (a voxel nearer than thr to a triangle lies inside that triangle's index box); larger
       values are only upper bounds, so no step from any sample may exceed thr/2 */
    volume->DensityScale = threshold;
    volume->StepMax = 0.5f * threshold;
    return volume;
}

}  // namespace Voxelizer
}  // namespace VolumeRaytracer

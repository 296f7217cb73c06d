#include "VolumeConverter.h"

#include "../../../include/vrt.h"
#include "../voxelize_core.h"

#include <cmath>
#include <iostream>
#include <vector>

namespace VolumeRaytracer {
namespace Voxelizer {

namespace {

using vrt_vox::TriangleFrame;

inline vrt_vox::V3 to_v3(const VVector& v) { return vrt_vox::v3(v.X, v.Y, v.Z); }

/* One triangle into the volume: every voxel of its index box keeps the smaller of its density and the
   shell density of its distance to this triangle (VoxelizeFace, VolumeConverter.cpp:161-252).  The
   arithmetic lives in ../voxelize_core.h, shared with the HIP kernel. */
void voxelize_face(Voxel::VVoxelVolume& volume, const TriangleFrame& t, float threshold) {
    const float cell = volume.GetCellSize(), extent = volume.GetVolumeExtends();
    for (int x = t.lo[0]; x <= t.hi[0]; x++)
        for (int y = t.lo[1]; y <= t.hi[1]; y++)
            for (int z = t.lo[2]; z <= t.hi[2]; z++) {
                const VIntVector idx(x, y, z);
                const float dist = vrt_vox::region_distance(t, vrt_vox::voxel_position(x, y, z, cell, extent));
                const float density = vrt_vox::shell_density(dist, threshold);
                Voxel::VVoxel voxel = volume.GetVoxel(idx);
                if (density < voxel.Density) {
                    voxel.Density = density;
                    voxel.Material = voxel.Density <= 0.f ? 1 : 0;
                    volume.SetVoxel(idx, voxel);
                }
            }
}

}  // namespace

static vrt_ctx* g_device_ctx = nullptr;
void VVolumeConverter::UseDevice(vrt_ctx* ctx) { g_device_ctx = ctx; }

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
        std::cout << "[voxelizer] mesh '" << meshInfo.MeshName << "': no '_<resolution>' suffix in its name (e.g. cube_6): resolution 5" << std::endl;
    }
    if (resolution > 8) {
        std::cout << "[voxelizer] mesh '" << meshInfo.MeshName << "': resolution " << (int)resolution << " is outside 0..8: resolution 5" << std::endl;
        resolution = 5;
    }

    auto volume = std::make_shared<Voxel::VVoxelVolume>(resolution, extends);
    Voxel::VVoxel background;
    background.Material = 0;
    background.Density = extends * 2.f;
    volume->FillVolume(background);

    const float threshold = ExtractionThreshold(*volume);
    size_t skipped = 0;
    bool on_device = false;
    if (g_device_ctx) {
        /* the same loop on the GPU: upload the mesh, voxelize into a scratch slot, read the voxels back */
        constexpr int kScratchSlot = VRT_MAX_VOLUMES - 1;
        std::vector<float> pos(meshInfo.Vertices.size() * 3);
        for (size_t i = 0; i < meshInfo.Vertices.size(); i++) {
            pos[3 * i] = meshInfo.Vertices[i].Position.X;
            pos[3 * i + 1] = meshInfo.Vertices[i].Position.Y;
            pos[3 * i + 2] = meshInfo.Vertices[i].Position.Z;
        }
        std::vector<uint32_t> idx(meshInfo.Indices.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = meshInfo.Indices[i] > 0xfffffffeull ? 0xffffffffu : (uint32_t)meshInfo.Indices[i];
        static_assert(sizeof(Voxel::VVoxel) == sizeof(vrt_voxel), "VVoxel must match the wire record");
        int rc = vrt_voxelize_mesh(g_device_ctx, kScratchSlot, resolution, extends, pos.data(), meshInfo.Vertices.size(), idx.data(), idx.size(), &skipped);
        if (rc == VRT_OK) rc = vrt_volume_download(g_device_ctx, kScratchSlot, reinterpret_cast<vrt_voxel*>(volume->GetVoxels().data()));
        if (rc == VRT_OK) {
            (void)vrt_volume_free(g_device_ctx, kScratchSlot);
            on_device = true;
        } else {
            std::cout << "[WARNING] device Voxelizer failed (" << vrt_strerror(rc) << "); converting " << meshInfo.MeshName << " on the host" << std::endl;
            volume->FillVolume(background);
            skipped = 0;
        }
    }
    for (size_t i = 0; !on_device && i + 3 <= meshInfo.Indices.size(); i += 3) {
        const size_t a = meshInfo.Indices[i], b = meshInfo.Indices[i + 1], c = meshInfo.Indices[i + 2];
        if (a >= meshInfo.Vertices.size() || b >= meshInfo.Vertices.size() || c >= meshInfo.Vertices.size()) {
            skipped++;
            continue;
        }
        TriangleFrame t;
        if (!vrt_vox::make_frame(to_v3(meshInfo.Vertices[a].Position), to_v3(meshInfo.Vertices[b].Position), to_v3(meshInfo.Vertices[c].Position), t)) {
            skipped++;
            continue;
        }
        vrt_vox::index_box(t, threshold, volume->GetVolumeExtends(), volume->GetCellSize(), (int)volume->GetSize());
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

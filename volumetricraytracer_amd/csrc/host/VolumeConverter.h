/*
 * VolumeConverter.h — triangle mesh → VVoxelVolume (the Voxelizer's hot loop, BASELINE config 1).
 *
 * Restates Voxelizer/Private/VolumeConverter.cpp:30-84 (ConvertMeshInfoToVoxelVolume),
 * :161-252 (VoxelizeFace), :656-679 (resolution from the mesh-name suffix), :681-781 (triangle
 * bounding box, 7-region point/triangle classification).  For every triangle, every voxel in the
 * triangle's (bbox ± thr ± 1 voxel) index box gets  density = dist/thr − 0.5  (thr = cell·√3),
 * where dist is the distance to the triangle's face / edge / vertex region the voxel projects
 * into, keeping the minimum over triangles; untouched voxels keep 2·extent.  The result is an
 * UNSIGNED shell: |surface distance| < thr/2 ⇔ density < 0.
 */
#pragma once
#include <memory>
#include "HostVoxel.h"
#include "VoxelizerTypes.h"

struct vrt_ctx; /* include/vrt.h */

namespace VolumeRaytracer {
namespace Voxelizer {

class VVolumeConverter {
public:
    static std::shared_ptr<Voxel::VVoxelVolume> ConvertMeshInfoToVoxelVolume(const VMeshInfo& meshInfo, const VTextureLibrary& textureLib);
    /* Run the per-triangle loop on the GPU (vrt_voxelize_mesh, include/vrt.h) instead of on the host: same
       volume, bit for bit (both builds compile csrc/voxelize_core.h).  ctx = a live vrt_ctx, or nullptr to go
       back to the CPU loop.  The converter uses (and overwrites) volume slot 19 of that context. */
    static void UseDevice(::vrt_ctx* ctx);
    static bool ExtractResolutionFromName(const std::string& name, uint8_t& outResolution);
    /* extraction threshold of a volume: cell size · √3 (VolumeConverter.cpp:57) */
    static float ExtractionThreshold(const Voxel::VVoxelVolume& volume);
};

}  // namespace Voxelizer
}  // namespace VolumeRaytracer

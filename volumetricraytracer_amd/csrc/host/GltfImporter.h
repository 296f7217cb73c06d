/*
 * GltfImporter.h — minimal glTF 2.0 reader (.gltf JSON + external .bin or base64 data URIs; .glb
 * is not supported) producing the Voxelizer's VSceneInfo.
 *
 * Restates Voxelizer/Private/GLTFImporter.cpp:21-271: one VMeshInfo per mesh (all primitives
 * appended; POSITION and NORMAL are both required; indices u16/u32); positions are scaled by 100
 * and re-centred on the POSITION accessor's min/max midpoint, bounds = half-size + 5; material =
 * pbrMetallicRoughness of the first primitive; nodes with a mesh become objects (translation
 * x100); nodes whose name starts with "Light" become lights, typed by a "_Point" / "_Spot"
 * suffix, parameters from the node's `extras` (strength, color_r/g/b, attl, attexp, fangle, angle).
 */
#pragma once
#include <memory>
#include <string>
#include "VoxelizerTypes.h"

namespace VolumeRaytracer {
namespace Voxelizer {

class VGLTFImporter {
public:
    /* throws std::runtime_error on unreadable / malformed input */
    static std::shared_ptr<VSceneInfo> ImportScene(const std::string& gltfPath);
};

class VTextureLibraryImporter {
public:
    /* Voxelizer/Private/TextureLibraryImporter.cpp:22-96: {"materials":[{"material","tiling-x",
     * "tiling-y","albedo","normal","rm"}]}.  Paths are kept as written. */
    static bool Import(const std::string& jsonPath, VTextureLibrary& out);
};

}  // namespace Voxelizer
}  // namespace VolumeRaytracer

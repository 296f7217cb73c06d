/* SceneConverter.h — VSceneInfo → VScene (one VVoxelVolume per mesh, one VVoxelObject per node,
 * lights by type).  Restates Voxelizer/Private/SceneConverter.cpp:25-91. */
#pragma once
#include "HostScene.h"
#include "VoxelizerTypes.h"

namespace VolumeRaytracer {
namespace Voxelizer {

class VSceneConverter {
public:
    static VObjectPtr<Scene::VScene> ConvertSceneInfoToScene(const VSceneInfo& sceneInfo, const VTextureLibrary& textureLib);
};

/* The Voxelizer executable's whole job (Voxelizer/Private/Voxelizer.cpp:36-117): import
 * `gltfPath` (+ optional texture library), convert, write "<stem>.vox" next to it (or to
 * outPathOrEmpty).  Returns the written path; throws std::runtime_error on failure. */
std::string VoxelizeFile(const std::string& gltfPath, const std::string& textureLibraryOrEmpty, const std::string& outPathOrEmpty);

}  // namespace Voxelizer
}  // namespace VolumeRaytracer

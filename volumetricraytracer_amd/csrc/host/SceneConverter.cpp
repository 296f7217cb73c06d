#include "SceneConverter.h"

#include <iostream>
#include <stdexcept>

#include "GltfImporter.h"
#include "HostSerialization.h"
#include "VolumeConverter.h"

namespace VolumeRaytracer {
namespace Voxelizer {

namespace {

/* What a light of the imported scene turns into, by type: the .vox scene contract keeps three lists —
   directional lights {Color, Strength}, point lights + {AttL, AttExp}, spot lights + {AngleF, Angle}. */
void spawn_light(Scene::VScene& scene, const VLightInfo& in) {
    auto common = [&](Scene::VLight& out) {
        out.Color = in.Color;
        out.IlluminationStrength = in.Intensity;
    };
    if (in.LightType == ELightType::SPOT) {
        auto spot = scene.SpawnObject<Scene::VSpotLight>(in.Position, in.Rotation, VVector::ONE);
        common(*spot);
        spot->AttenuationLinear = in.AttL;
        spot->AttenuationExp = in.AttExp;
        spot->FalloffAngle = in.FalloffAngle;
        spot->Angle = in.Angle;
    } else if (in.LightType == ELightType::POINT) {
        auto point = scene.SpawnObject<Scene::VPointLight>(in.Position, in.Rotation, VVector::ONE);
        common(*point);
        point->AttenuationLinear = in.AttL;
        point->AttenuationExp = in.AttExp;
    } else {
        common(*scene.SpawnObject<Scene::VLight>(in.Position, in.Rotation, VVector::ONE));
    }
}

}  // namespace

/* One VVoxelVolume per imported mesh (shared by every node that places it), one VVoxelObject per node, one light object per
   imported light.  Written from the scene contract the .vox file holds, not from the reference converter's text
   (Voxelizer/Private/SceneConverter.cpp does the same job). */
VObjectPtr<Scene::VScene> VSceneConverter::ConvertSceneInfoToScene(const VSceneInfo& sceneInfo, const VTextureLibrary& textureLib) {
    auto scene = std::make_shared<Scene::VScene>();
    std::cout << "[voxelizer] " << sceneInfo.Meshes.size() << " mesh(es) -> voxel volumes, " << sceneInfo.Objects.size() << " object(s), "
              << sceneInfo.Lights.size() << " light(s)" << std::endl;
    /* meshes first: a volume is converted once, however many nodes place it */
    std::map<std::string, VObjectPtr<Voxel::VVoxelVolume>> volume_of_mesh;
    for (const auto& [mesh_id, mesh] : sceneInfo.Meshes) volume_of_mesh.emplace(mesh_id, VVolumeConverter::ConvertMeshInfoToVoxelVolume(mesh, textureLib));
    for (const VObjectInfo& node : sceneInfo.Objects) {
        const auto found = volume_of_mesh.find(node.MeshID);
        auto placed = scene->SpawnObject<Scene::VVoxelObject>(node.Position, node.Rotation, node.Scale);
        placed->SetVoxelVolume(found != volume_of_mesh.end() ? found->second : nullptr); /* (a node without a mesh: an empty object) */
    }
    for (const VLightInfo& light : sceneInfo.Lights) spawn_light(*scene, light);
    return scene;
}

std::string VoxelizeFile(const std::string& gltfPath, const std::string& textureLibraryOrEmpty, const std::string& outPathOrEmpty) {
    VTextureLibrary textureLib;
    if (!textureLibraryOrEmpty.empty() && !VTextureLibraryImporter::Import(textureLibraryOrEmpty, textureLib))
        std::cerr << "[ERROR] Unable to load specified texture library!" << std::endl;
    std::shared_ptr<VSceneInfo> info = VGLTFImporter::ImportScene(gltfPath);
    VObjectPtr<Scene::VScene> scene = VSceneConverter::ConvertSceneInfoToScene(*info, textureLib);
    std::string out = outPathOrEmpty;
    if (out.empty()) {
        const size_t dot = gltfPath.find_last_of('.');
        const size_t slash = gltfPath.find_last_of("/\\");
        out = (dot != std::string::npos && (slash == std::string::npos || dot > slash) ? gltfPath.substr(0, dot) : gltfPath) + ".vox";
    }
    if (!VSerializationManager::SaveToFile(*scene, out)) throw std::runtime_error("cannot write " + out);
    return out;
}

}  // namespace Voxelizer
}  // namespace VolumeRaytracer

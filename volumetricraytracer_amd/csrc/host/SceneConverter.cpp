#include "SceneConverter.h"

#include <iostream>
#include <stdexcept>

#include "GltfImporter.h"
#include "HostSerialization.h"
#include "VolumeConverter.h"

namespace VolumeRaytracer {
namespace Voxelizer {

VObjectPtr<Scene::VScene> VSceneConverter::ConvertSceneInfoToScene(const VSceneInfo& sceneInfo, const VTextureLibrary& textureLib) {
    auto scene = std::make_shared<Scene::VScene>();
    std::cout << "Convert imported scene to voxel scene" << std::endl;
    std::cout << "Converting meshes to voxel volumes" << std::endl;
    std::map<std::string, VObjectPtr<Voxel::VVoxelVolume>> volumes;
    for (const auto& mesh : sceneInfo.Meshes) volumes[mesh.first] = VVolumeConverter::ConvertMeshInfoToVoxelVolume(mesh.second, textureLib);
    std::cout << "Converting scene objects" << std::endl;
    for (const auto& object : sceneInfo.Objects) {
        auto obj = scene->SpawnObject<Scene::VVoxelObject>(object.Position, object.Rotation, object.Scale);
        obj->SetVoxelVolume(volumes[object.MeshID]);
    }
    std::cout << "Converting lights" << std::endl;
    for (const auto& light : sceneInfo.Lights) {
        switch (light.LightType) {
            case ELightType::POINT: {
                auto l = scene->SpawnObject<Scene::VPointLight>(light.Position, light.Rotation, VVector::ONE);
                l->Color = light.Color;
                l->IlluminationStrength = light.Intensity;
                l->AttenuationExp = light.AttExp;
                l->AttenuationLinear = light.AttL;
            } break;
            case ELightType::SPOT: {
                auto l = scene->SpawnObject<Scene::VSpotLight>(light.Position, light.Rotation, VVector::ONE);
                l->Color = light.Color;
                l->IlluminationStrength = light.Intensity;
                l->AttenuationExp = light.AttExp;
                l->AttenuationLinear = light.AttL;
                l->Angle = light.Angle;
                l->FalloffAngle = light.FalloffAngle;
            } break;
            default: {
                auto l = scene->SpawnObject<Scene::VLight>(light.Position, light.Rotation, VVector::ONE);
                l->Color = light.Color;
                l->IlluminationStrength = light.Intensity;
            } break;
        }
    }
    std::cout << "Scene conversion finished" << std::endl;
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

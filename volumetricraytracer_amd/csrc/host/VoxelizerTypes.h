/* VoxelizerTypes.h — intermediate scene description between the glTF importer and the voxel
 * converter.  Reference: Voxelizer/Public/SceneInfo.h:26-110. */
#pragma once
#include <map>
#include <string>
#include <vector>
#include "HostCore.h"

namespace VolumeRaytracer {
namespace Voxelizer {

struct VVertex {
    VVector Position;
    VVector Normal;
};

struct VMeshInfo {
    std::string MeshName;
    std::vector<VVertex> Vertices;
    std::vector<size_t> Indices;
    VAABB Bounds;
    std::string MaterialName;
    VMaterial Material;
};

struct VObjectInfo {
    std::string MeshID;
    VVector Position;
    VVector Scale;
    VQuat Rotation;
};

enum class ELightType { DIRECTIONAL, POINT, SPOT };

struct VLightInfo {
    VVector Position;
    VQuat Rotation;
    ELightType LightType = ELightType::DIRECTIONAL;
    VColor Color = VColor::WHITE;
    float Intensity = 0.f, AttL = 0.f, AttExp = 0.f, FalloffAngle = 0.f, Angle = 0.f;
};

struct VMaterialTextures {
    VVector2D TextureTiling = VVector2D(100.f, 100.f);
    std::string Albedo, Normal, RM;
};
struct VTextureLibrary {
    std::map<std::string, VMaterialTextures> Materials;
};

struct VSceneInfo {
    std::map<std::string, VMeshInfo> Meshes; /* key: glTF mesh index as a string (the glTF SDK's id) */
    std::vector<VObjectInfo> Objects;
    std::vector<VLightInfo> Lights;
};

}  // namespace Voxelizer
}  // namespace VolumeRaytracer

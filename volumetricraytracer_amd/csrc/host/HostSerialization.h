/*
 * HostSerialization.h — the reference's recursive property-bag archive and the .vox scene file.
 *
 * File layout (Core/Private/SerializationManager.cpp:24-100; size_t is 8 bytes, little endian):
 *   archive := u64 BufferSize, BufferSize bytes, u64 numProps,
 *              numProps x { u64 nameLen (including the NUL), name bytes, archive }
 * Property order on disk is boost::unordered_map iteration order, i.e. arbitrary: readers must be
 * order-agnostic (this one is; the writer emits names sorted so that files are reproducible).
 *
 * Scene (Scene/Private/Scene.cpp:392-458): VCount, V_i, OCount, OI_i, O_i, LDCount, LD_i, LPCount,
 * LP_i, LSCount, LS_i.  Volume (Voxel/Private/VoxelVolume.cpp:178-219): buffer = N^3 VVoxel records,
 * props Resolution(u8), Extends(f32), Material{Color 4f, Roughness, Metallic, TextureScale 2f,
 * AlbedoTexture/NormalTexture/RMTexture C strings} (Core/Private/Material.cpp:19-70).  Objects and
 * lights: Scene/Private/VoxelObject.cpp:37-71, Light.cpp:17-57, PointLight.cpp:17-33, SpotLight.cpp:17-37.
 */
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "HostScene.h"

namespace VolumeRaytracer {

struct VSerializationArchive {
    std::vector<char> Buffer;
    std::map<std::string, std::shared_ptr<VSerializationArchive>> Properties;

    template <typename T> static std::shared_ptr<VSerializationArchive> From(const T* src) {
        auto a = std::make_shared<VSerializationArchive>();
        a->Buffer.resize(sizeof(T));
        memcpy(a->Buffer.data(), src, sizeof(T));
        return a;
    }
    static std::shared_ptr<VSerializationArchive> FromString(const std::string& s) {
        auto a = std::make_shared<VSerializationArchive>();
        a->Buffer.assign(s.c_str(), s.c_str() + s.size() + 1);
        return a;
    }
    /* Reads sizeof(T) bytes (the reference copies BufferSize bytes — ISerializable.h:47-58 — a quirk
     * that is not inherited). */
    template <typename T> T To() const {
        T res{};
        if (Buffer.size() >= sizeof(T)) memcpy(&res, Buffer.data(), sizeof(T));
        return res;
    }
    bool Has(const std::string& name) const { return Properties.find(name) != Properties.end(); }
    const VSerializationArchive& At(const std::string& name) const;
};

class VSerializationManager {
public:
    static bool WriteArchive(const VSerializationArchive& archive, const std::string& filePath);
    static std::shared_ptr<VSerializationArchive> ReadArchive(const std::string& filePath);
    static bool SaveToFile(const Scene::VScene& scene, const std::string& filePath);
    static VObjectPtr<Scene::VScene> LoadSceneFromFile(const std::string& filePath);

    static std::shared_ptr<VSerializationArchive> Serialize(const Scene::VScene& scene);
    static std::shared_ptr<VSerializationArchive> Serialize(const Voxel::VVoxelVolume& volume);
    static std::shared_ptr<VSerializationArchive> Serialize(const VMaterial& material);
    /* sourcePath: the .vox file the archive was read from; material texture paths that are not absolute are resolved
       against its folder (VMaterial::Deserialize, Core/Private/Material.cpp:72-100).  Empty: paths stay as stored. */
    static void Deserialize(const VSerializationArchive& a, Voxel::VVoxelVolume& volume, const std::string& sourcePath = std::string());
    static void Deserialize(const VSerializationArchive& a, VMaterial& material, const std::string& sourcePath = std::string());
    static void Deserialize(const VSerializationArchive& a, Scene::VScene& scene, const std::string& sourcePath = std::string());
};

}  // namespace VolumeRaytracer

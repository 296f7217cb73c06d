#include "HostSerialization.h"

#include <filesystem>
#include <fstream>
#include <stdexcept>

namespace VolumeRaytracer {

const VSerializationArchive& VSerializationArchive::At(const std::string& name) const {
    auto it = Properties.find(name);
    if (it == Properties.end() || !it->second) throw std::runtime_error("archive has no property '" + name + "'");
    return *it->second;
}

namespace {

void write_u64(std::ostream& s, uint64_t v) { s.write(reinterpret_cast<const char*>(&v), 8); }
bool read_u64(std::istream& s, uint64_t& v) { return (bool)s.read(reinterpret_cast<char*>(&v), 8); }

void write_archive(const VSerializationArchive& a, std::ostream& s) {
    write_u64(s, a.Buffer.size());
    if (!a.Buffer.empty()) s.write(a.Buffer.data(), (std::streamsize)a.Buffer.size());
    write_u64(s, a.Properties.size());
    for (const auto& kv : a.Properties) {
        write_u64(s, kv.first.size() + 1);
        s.write(kv.first.c_str(), (std::streamsize)kv.first.size() + 1);
        write_archive(*kv.second, s);
    }
}

std::shared_ptr<VSerializationArchive> read_archive(std::istream& s, uint64_t remaining_hint, int depth) {
    if (depth > 64) throw std::runtime_error("archive nesting too deep");
    auto a = std::make_shared<VSerializationArchive>();
    uint64_t size = 0, props = 0;
    if (!read_u64(s, size)) throw std::runtime_error("truncated archive (buffer size)");
    if (size > remaining_hint) throw std::runtime_error("archive buffer larger than the file");
    a->Buffer.resize((size_t)size);
    if (size && !s.read(a->Buffer.data(), (std::streamsize)size)) throw std::runtime_error("truncated archive (buffer)");
    if (!read_u64(s, props)) throw std::runtime_error("truncated archive (property count)");
    if (props > (1u << 24)) throw std::runtime_error("implausible property count");
    for (uint64_t i = 0; i < props; i++) {
        uint64_t len = 0;
        if (!read_u64(s, len) || len == 0 || len > 4096) throw std::runtime_error("bad property name length");
        std::string name((size_t)len, '\0');
        if (!s.read(&name[0], (std::streamsize)len)) throw std::runtime_error("truncated archive (property name)");
        name.resize(strnlen(name.c_str(), (size_t)len));
        a->Properties[name] = read_archive(s, remaining_hint, depth + 1);
    }
    return a;
}

template <typename T> std::shared_ptr<VSerializationArchive> level_object(const T& o) {
    auto a = std::make_shared<VSerializationArchive>();
    a->Properties["Position"] = VSerializationArchive::From(&o.Position);
    a->Properties["Scale"] = VSerializationArchive::From(&o.Scale);
    a->Properties["Rotation"] = VSerializationArchive::From(&o.Rotation);
    return a;
}
void read_level_object(const VSerializationArchive& a, Scene::VLevelObject& o) {
    o.Position = a.At("Position").To<VVector>();
    o.Scale = a.At("Scale").To<VVector>();
    o.Rotation = a.At("Rotation").To<VQuat>();
}
std::shared_ptr<VSerializationArchive> light(const Scene::VLight& l) {
    auto a = level_object(l);
    a->Properties["Color"] = VSerializationArchive::From(&l.Color);
    a->Properties["Strength"] = VSerializationArchive::From(&l.IlluminationStrength);
    return a;
}
void read_light(const VSerializationArchive& a, Scene::VLight& l) {
    read_level_object(a, l);
    l.Color = a.At("Color").To<VColor>();
    l.IlluminationStrength = a.At("Strength").To<float>();
}
std::string idx(const char* prefix, size_t i) { return std::string(prefix) + std::to_string(i); }

}  // namespace

bool VSerializationManager::WriteArchive(const VSerializationArchive& archive, const std::string& filePath) {
    std::ofstream s(filePath, std::ios::binary);
    if (!s) return false;
    write_archive(archive, s);
    return (bool)s;
}

std::shared_ptr<VSerializationArchive> VSerializationManager::ReadArchive(const std::string& filePath) {
    std::ifstream s(filePath, std::ios::binary | std::ios::ate);
    if (!s) return nullptr;
    const uint64_t file_size = (uint64_t)s.tellg();
    s.seekg(0);
    return read_archive(s, file_size, 0);
}

std::shared_ptr<VSerializationArchive> VSerializationManager::Serialize(const VMaterial& m) {
    auto a = std::make_shared<VSerializationArchive>();
    a->Properties["Color"] = VSerializationArchive::From(&m.AlbedoColor);
    a->Properties["Roughness"] = VSerializationArchive::From(&m.Roughness);
    a->Properties["Metallic"] = VSerializationArchive::From(&m.Metallic);
    a->Properties["TextureScale"] = VSerializationArchive::From(&m.TextureScale);
    a->Properties["AlbedoTexture"] = VSerializationArchive::FromString(m.AlbedoTexturePath);
    a->Properties["NormalTexture"] = VSerializationArchive::FromString(m.NormalTexturePath);
    a->Properties["RMTexture"] = VSerializationArchive::FromString(m.RMTexturePath); /* the reference writes the albedo path here (Material.cpp:59); not inherited */
    return a;
}

/* VMaterial::Deserialize(sourcePath, archive), Core/Private/Material.cpp:72-100: texture paths that are not absolute are
   relative to the folder of the .vox file they came from. */
void VSerializationManager::Deserialize(const VSerializationArchive& a, VMaterial& m, const std::string& sourcePath) {
    m.AlbedoColor = a.At("Color").To<VColor>();
    m.Roughness = a.At("Roughness").To<float>();
    /* files written before these properties existed do not carry them: read when present */
    if (a.Has("Metallic")) m.Metallic = a.At("Metallic").To<float>();
    if (a.Has("TextureScale")) m.TextureScale = a.At("TextureScale").To<VVector2D>();
    const std::filesystem::path folder = sourcePath.empty() ? std::filesystem::path() : std::filesystem::path(sourcePath).parent_path();
    auto str = [&](const char* k, std::string& out) {
        if (!a.Has(k) || a.At(k).Buffer.empty()) return;
        out.assign(a.At(k).Buffer.data(), strnlen(a.At(k).Buffer.data(), a.At(k).Buffer.size()));
        if (!out.empty() && !folder.empty() && !std::filesystem::path(out).is_absolute()) out = (folder / out).string();
    };
    str("AlbedoTexture", m.AlbedoTexturePath);
    str("NormalTexture", m.NormalTexturePath);
    str("RMTexture", m.RMTexturePath);
}

std::shared_ptr<VSerializationArchive> VSerializationManager::Serialize(const Voxel::VVoxelVolume& v) {
    auto a = std::make_shared<VSerializationArchive>();
    const auto& vox = v.GetVoxels();
    a->Buffer.resize(vox.size() * sizeof(Voxel::VVoxel));
    /* write the records with zeroed padding so that files are byte-reproducible */
    for (size_t i = 0; i < vox.size(); i++) {
        char rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        rec[0] = (char)vox[i].Material;
        memcpy(rec + 4, &vox[i].Density, 4);
        memcpy(a->Buffer.data() + i * 8, rec, 8);
    }
    const uint8_t res = v.GetResolution();
    const float ext = v.GetVolumeExtends();
    a->Properties["Resolution"] = VSerializationArchive::From(&res);
    a->Properties["Extends"] = VSerializationArchive::From(&ext);
    a->Properties["Material"] = Serialize(v.GetMaterial());
    return a;
}

void VSerializationManager::Deserialize(const VSerializationArchive& a, Voxel::VVoxelVolume& v, const std::string& sourcePath) {
    const uint8_t res = a.At("Resolution").To<uint8_t>();
    const float ext = a.At("Extends").To<float>();
    if (res > 10) throw std::runtime_error("volume resolution out of range");
    v.Reset(res, ext);
    VMaterial m;
    Deserialize(a.At("Material"), m, sourcePath);
    v.SetMaterial(m);
    if (a.Buffer.size() < v.GetVoxelCount() * sizeof(Voxel::VVoxel)) throw std::runtime_error("volume buffer too small");
    auto& vox = v.GetVoxels();
    for (size_t i = 0; i < vox.size(); i++) {
        vox[i].Material = (uint8_t)a.Buffer[i * 8];
        memcpy(&vox[i].Density, a.Buffer.data() + i * 8 + 4, 4);
    }
    v.MakeDirty();
}

std::shared_ptr<VSerializationArchive> VSerializationManager::Serialize(const Scene::VScene& scene) {
    auto a = std::make_shared<VSerializationArchive>();
    const auto volumes = scene.GetAllRegisteredVolumes();
    uint64_t n = volumes.size();
    a->Properties["VCount"] = VSerializationArchive::From(&n);
    for (size_t i = 0; i < volumes.size(); i++) a->Properties[idx("V_", i)] = Serialize(*volumes[i]);
    std::vector<std::shared_ptr<Scene::VVoxelObject>> objects;
    std::vector<std::shared_ptr<Scene::VLight>> dir;
    std::vector<std::shared_ptr<Scene::VPointLight>> point;
    std::vector<std::shared_ptr<Scene::VSpotLight>> spot;
    for (const auto& o : scene.GetAllPlacedObjects()) {
        if (auto p = std::dynamic_pointer_cast<Scene::VPointLight>(o)) point.push_back(p);
        else if (auto s = std::dynamic_pointer_cast<Scene::VSpotLight>(o)) spot.push_back(s);
        else if (auto l = std::dynamic_pointer_cast<Scene::VLight>(o)) dir.push_back(l);
        else if (auto v = std::dynamic_pointer_cast<Scene::VVoxelObject>(o)) { if (v->GetVoxelVolume()) objects.push_back(v); }
    }
    n = objects.size();
    a->Properties["OCount"] = VSerializationArchive::From(&n);
    for (size_t i = 0; i < objects.size(); i++) {
        uint64_t vi = 0;
        for (size_t k = 0; k < volumes.size(); k++) if (volumes[k] == objects[i]->GetVoxelVolume()) vi = k;
        a->Properties[idx("OI_", i)] = VSerializationArchive::From(&vi);
        a->Properties[idx("O_", i)] = level_object(*objects[i]);
    }
    n = dir.size();
    a->Properties["LDCount"] = VSerializationArchive::From(&n);
    for (size_t i = 0; i < dir.size(); i++) a->Properties[idx("LD_", i)] = light(*dir[i]);
    n = point.size();
    a->Properties["LPCount"] = VSerializationArchive::From(&n);
    for (size_t i = 0; i < point.size(); i++) {
        auto l = light(*point[i]);
        l->Properties["AttL"] = VSerializationArchive::From(&point[i]->AttenuationLinear);
        l->Properties["AttExp"] = VSerializationArchive::From(&point[i]->AttenuationExp);
        a->Properties[idx("LP_", i)] = l;
    }
    n = spot.size();
    a->Properties["LSCount"] = VSerializationArchive::From(&n);
    for (size_t i = 0; i < spot.size(); i++) {
        auto l = light(*spot[i]);
        l->Properties["AttL"] = VSerializationArchive::From(&spot[i]->AttenuationLinear);
        l->Properties["AttExp"] = VSerializationArchive::From(&spot[i]->AttenuationExp);
        l->Properties["AngleF"] = VSerializationArchive::From(&spot[i]->FalloffAngle);
        l->Properties["Angle"] = VSerializationArchive::From(&spot[i]->Angle);
        a->Properties[idx("LS_", i)] = l;
    }
    return a;
}

void VSerializationManager::Deserialize(const VSerializationArchive& a, Scene::VScene& scene, const std::string& sourcePath) {
    const uint64_t volumesCount = a.At("VCount").To<uint64_t>();
    const uint64_t objectsCount = a.At("OCount").To<uint64_t>();
    const uint64_t dirCount = a.Has("LDCount") ? a.At("LDCount").To<uint64_t>() : 0;
    const uint64_t pointCount = a.Has("LPCount") ? a.At("LPCount").To<uint64_t>() : 0;
    const uint64_t spotCount = a.Has("LSCount") ? a.At("LSCount").To<uint64_t>() : 0;
    if (volumesCount > 4096 || objectsCount > (1u << 20)) throw std::runtime_error("implausible scene counts");
    std::vector<VObjectPtr<Voxel::VVoxelVolume>> volumes;
    for (uint64_t i = 0; i < volumesCount; i++) {
        auto v = std::make_shared<Voxel::VVoxelVolume>(1, 1.f);
        Deserialize(a.At(idx("V_", i)), *v, sourcePath);
        volumes.push_back(v);
    }
    for (uint64_t i = 0; i < objectsCount; i++) {
        const uint64_t vi = a.At(idx("OI_", i)).To<uint64_t>();
        if (vi >= volumes.size()) throw std::runtime_error("object references a missing volume");
        auto obj = scene.SpawnObject<Scene::VVoxelObject>(VVector::ZERO, VQuat::IDENTITY, VVector::ONE);
        read_level_object(a.At(idx("O_", i)), *obj);
        obj->SetVoxelVolume(volumes[(size_t)vi]);
    }
    for (uint64_t i = 0; i < dirCount; i++) {
        auto l = scene.SpawnObject<Scene::VLight>(VVector::ZERO, VQuat::IDENTITY, VVector::ONE);
        read_light(a.At(idx("LD_", i)), *l);
        scene.SetActiveDirectionalLight(l);
    }
    for (uint64_t i = 0; i < pointCount; i++) {
        auto l = scene.SpawnObject<Scene::VPointLight>(VVector::ZERO, VQuat::IDENTITY, VVector::ONE);
        const auto& la = a.At(idx("LP_", i));
        read_light(la, *l);
        l->AttenuationLinear = la.At("AttL").To<float>();
        l->AttenuationExp = la.At("AttExp").To<float>();
    }
    for (uint64_t i = 0; i < spotCount; i++) {
        auto l = scene.SpawnObject<Scene::VSpotLight>(VVector::ZERO, VQuat::IDENTITY, VVector::ONE);
        const auto& la = a.At(idx("LS_", i));
        read_light(la, *l);
        l->AttenuationLinear = la.At("AttL").To<float>();
        l->AttenuationExp = la.At("AttExp").To<float>();
        l->FalloffAngle = la.At("AngleF").To<float>();
        l->Angle = la.At("Angle").To<float>();
    }
}

bool VSerializationManager::SaveToFile(const Scene::VScene& scene, const std::string& filePath) {
    return WriteArchive(*Serialize(scene), filePath);
}

VObjectPtr<Scene::VScene> VSerializationManager::LoadSceneFromFile(const std::string& filePath) {
    auto a = ReadArchive(filePath);
    if (!a) return nullptr;
    auto scene = std::make_shared<Scene::VScene>();
    Deserialize(*a, *scene, filePath);
    return scene;
}

}  // namespace VolumeRaytracer

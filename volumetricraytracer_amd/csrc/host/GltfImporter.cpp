#include "GltfImporter.h"

#include <cstring>
#include <algorithm>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "MiniJson.h"

namespace VolumeRaytracer {
namespace Voxelizer {

namespace {

std::string read_file(const std::string& path, bool binary) {
    std::ifstream f(path, binary ? std::ios::binary : std::ios::in);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

std::string dir_of(const std::string& path) {
    const size_t at = path.find_last_of("/\\");
    return at == std::string::npos ? std::string() : path.substr(0, at + 1);
}

std::string base64_decode(const std::string& in) {
    std::string out;
    unsigned acc = 0;
    int bits = 0;
    for (char ch : in) {
        int v;
        if (ch >= 'A' && ch <= 'Z') v = ch - 'A';
        else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26;
        else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
        else if (ch == '+') v = 62;
        else if (ch == '/') v = 63;
        else continue; /* '=' padding, whitespace */
        acc = (acc << 6) | (unsigned)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out += (char)((acc >> bits) & 0xff);
        }
    }
    return out;
}

struct Buffers {
    std::vector<std::string> data;
};

/* Reads `count` elements of `components` scalars of type T from an accessor (tightly packed or strided). */
template <typename T> std::vector<T> read_accessor(const minijson::Value& doc, const Buffers& buffers, long long accessorIndex, int components) {
    const minijson::Value& acc = doc["accessors"][(size_t)accessorIndex];
    if (!acc.IsObject()) throw std::runtime_error("accessor index out of range");
    const long long viewIndex = acc["bufferView"].GetInt(-1);
    const minijson::Value& view = doc["bufferViews"][(size_t)viewIndex];
    if (viewIndex < 0 || !view.IsObject()) throw std::runtime_error("accessor without a bufferView (sparse accessors are not supported)");
    const long long bufferIndex = view["buffer"].GetInt(-1);
    if (bufferIndex < 0 || (size_t)bufferIndex >= buffers.data.size()) throw std::runtime_error("bufferView references a missing buffer");
    const std::string& bytes = buffers.data[(size_t)bufferIndex];
    const size_t count = (size_t)acc["count"].GetInt(0);
    const size_t elem = sizeof(T) * (size_t)components;
    const size_t stride = view.Has("byteStride") ? (size_t)view["byteStride"].GetInt(0) : elem;
    const size_t base = (size_t)view["byteOffset"].GetInt(0) + (size_t)acc["byteOffset"].GetInt(0);
    if (count && base + (count - 1) * stride + elem > bytes.size()) throw std::runtime_error("accessor reads past the end of its buffer");
    std::vector<T> out(count * (size_t)components);
    for (size_t i = 0; i < count; i++) memcpy(&out[i * (size_t)components], bytes.data() + base + i * stride, elem);
    return out;
}

bool starts_with(const std::string& s, const char* prefix) { return s.compare(0, strlen(prefix), prefix) == 0; }

VLightInfo light_from_node(const minijson::Value& node) {
    VLightInfo li;
    const minijson::Value& t = node["translation"];
    const minijson::Value& r = node["rotation"];
    li.Position = VVector(t[0].GetFloat(), t[1].GetFloat(), t[2].GetFloat()) * 100.f;
    li.Rotation = r.IsArray() ? VQuat(r[0].GetFloat(), r[1].GetFloat(), r[2].GetFloat(), r[3].GetFloat(1.f)) : VQuat::IDENTITY;
    const std::string name = node["name"].GetString();
    const size_t at = name.find('_');
    if (at != std::string::npos) {
        const std::string type = name.substr(at + 1);
        if (starts_with(type, "Point")) li.LightType = ELightType::POINT;
        else if (starts_with(type, "Spot")) li.LightType = ELightType::SPOT;
    }
    const minijson::Value& ex = node["extras"];
    if (ex.IsObject()) {
        if (ex["strength"].IsNumber()) li.Intensity = ex["strength"].GetFloat();
        if (ex["color_r"].IsNumber() && ex["color_g"].IsNumber() && ex["color_b"].IsNumber())
            li.Color = VColor(ex["color_r"].GetFloat(), ex["color_g"].GetFloat(), ex["color_b"].GetFloat(), 1.f);
        if (ex["attl"].IsNumber()) li.AttL = ex["attl"].GetFloat();
        if (ex["attexp"].IsNumber()) li.AttExp = ex["attexp"].GetFloat();
        if (ex["fangle"].IsNumber()) li.FalloffAngle = ex["fangle"].GetFloat();
        if (ex["angle"].IsNumber()) li.Angle = ex["angle"].GetFloat();
    }
    return li;
}

}  // namespace

/* Binary glTF container (.glb, glTF 2.0 spec §4.4): 12-byte header (magic "glTF", version 2, total length), then
   chunks of (u32 length, u32 type, payload): the first is JSON, an optional second one is BIN = buffer 0 when that
   buffer has no uri.  The reference gets this from the Microsoft glTF SDK's GLBResourceReader. */
bool split_glb(const std::string& file, std::string& json, std::string& bin) {
    auto u32 = [&](size_t at) {
        return (uint32_t)(unsigned char)file[at] | (uint32_t)(unsigned char)file[at + 1] << 8 | (uint32_t)(unsigned char)file[at + 2] << 16 |
               (uint32_t)(unsigned char)file[at + 3] << 24;
    };
    if (file.size() < 12 || u32(0) != 0x46546C67u) return false; /* "glTF" */
    if (u32(4) != 2) throw std::runtime_error("unsupported .glb container version");
    const size_t total = std::min<size_t>(u32(8), file.size());
    size_t at = 12;
    bool have_json = false;
    while (at + 8 <= total) {
        const size_t len = u32(at);
        const uint32_t type = u32(at + 4);
        if (at + 8 + len > total) throw std::runtime_error("truncated .glb chunk");
        if (type == 0x4E4F534Au && !have_json) { /* "JSON" */
            json = file.substr(at + 8, len);
            have_json = true;
        } else if (type == 0x004E4942u && bin.empty()) { /* "BIN\0" */
            bin = file.substr(at + 8, len);
        }
        at += 8 + ((len + 3) & ~(size_t)3);
    }
    if (!have_json) throw std::runtime_error(".glb without a JSON chunk");
    return true;
}

std::shared_ptr<VSceneInfo> VGLTFImporter::ImportScene(const std::string& gltfPath) {
    const std::string file = read_file(gltfPath, true);
    std::string json, glb_bin;
    const bool is_glb = split_glb(file, json, glb_bin);
    const minijson::ValuePtr root = minijson::Parse(is_glb ? json : file);
    const minijson::Value& doc = *root;
    if (!doc.IsObject()) throw std::runtime_error("glTF manifest is not a JSON object");

    Buffers buffers;
    for (size_t i = 0; i < doc["buffers"].Size(); i++) {
        const std::string uri = doc["buffers"][i]["uri"].GetString();
        if (starts_with(uri, "data:")) {
            const size_t comma = uri.find(',');
            if (comma == std::string::npos) throw std::runtime_error("malformed data URI");
            buffers.data.push_back(base64_decode(uri.substr(comma + 1)));
        } else if (!uri.empty()) {
            buffers.data.push_back(read_file(dir_of(gltfPath) + uri, true));
        } else if (is_glb && i == 0) {
            buffers.data.push_back(glb_bin); /* the container's BIN chunk */
        } else {
            throw std::runtime_error("buffer without a uri outside a .glb container");
        }
    }

    auto scene = std::make_shared<VSceneInfo>();
    std::cout << "[INFO] Importing meshes" << std::endl;
    for (size_t mi = 0; mi < doc["meshes"].Size(); mi++) {
        const minijson::Value& mesh = doc["meshes"][mi];
        VMeshInfo info;
        info.MeshName = mesh["name"].GetString();
        std::cout << "[INFO] Importing mesh: " << info.MeshName << std::endl;
        for (size_t pi = 0; pi < mesh["primitives"].Size(); pi++) {
            const minijson::Value& prim = mesh["primitives"][pi];
            const long long posAcc = prim["attributes"]["POSITION"].GetInt(-1);
            const long long nrmAcc = prim["attributes"]["NORMAL"].GetInt(-1);
            const long long idxAcc = prim["indices"].GetInt(-1);
            if (posAcc < 0 || nrmAcc < 0) {
                std::cout << "[WARNING] Invalid mesh primtive detected. Either no vertices or normals." << std::endl;
                continue;
            }
            if (idxAcc < 0 || !doc["accessors"][(size_t)idxAcc].IsObject() || !doc["accessors"][(size_t)posAcc].IsObject() ||
                !doc["accessors"][(size_t)nrmAcc].IsObject()) {
                std::cerr << "[ERROR] Invalid accessor data inside gltf file. File may be corrupted!" << std::endl;
                continue;
            }
            const minijson::Value& pa = doc["accessors"][(size_t)posAcc];
            VVector volumeOffset = VVector::ZERO;
            if (pa["min"].Size() >= 3 && pa["max"].Size() >= 3) {
                const VVector mn = VVector(pa["min"][0].GetFloat(), pa["min"][1].GetFloat(), pa["min"][2].GetFloat()) * 100.f;
                const VVector mx = VVector(pa["max"][0].GetFloat(), pa["max"][1].GetFloat(), pa["max"][2].GetFloat()) * 100.f;
                const VVector ext = (mx - mn) * 0.5f;
                volumeOffset = mx - ext;
                info.Bounds = VAABB(volumeOffset, ext + VVector::ONE * 5.f);
            } else {
                std::cout << "[WARNING] No bounds found for primitive!" << std::endl;
            }
            const long long idxType = doc["accessors"][(size_t)idxAcc]["componentType"].GetInt();
            const size_t firstVertex = info.Vertices.size();
            (void)firstVertex; /* like the reference, indices are NOT rebased per primitive */
            if (idxType == 5123) {
                for (unsigned short v : read_accessor<unsigned short>(doc, buffers, idxAcc, 1)) info.Indices.push_back(v);
            } else if (idxType == 5125) {
                for (unsigned int v : read_accessor<unsigned int>(doc, buffers, idxAcc, 1)) info.Indices.push_back(v);
            } else {
                std::cerr << "[ERROR] Unsupported indices format!" << std::endl;
                continue;
            }
            if (pa["componentType"].GetInt() != 5126 || doc["accessors"][(size_t)nrmAcc]["componentType"].GetInt() != 5126 ||
                pa["type"].GetString() != "VEC3" || doc["accessors"][(size_t)nrmAcc]["type"].GetString() != "VEC3") {
                std::cerr << "[ERROR] Unsupported vertex format!" << std::endl;
                continue;
            }
            const std::vector<float> positions = read_accessor<float>(doc, buffers, posAcc, 3);
            const std::vector<float> normals = read_accessor<float>(doc, buffers, nrmAcc, 3);
            if (positions.size() != normals.size()) {
                std::cerr << "[ERROR] Vertex and normal data are not the same size!" << std::endl;
                continue;
            }
            for (size_t c = 0; c + 2 < positions.size(); c += 3) {
                VVertex v;
                v.Position = VVector(positions[c], positions[c + 1], positions[c + 2]) * 100.f - volumeOffset;
                v.Normal = VVector(normals[c], normals[c + 1], normals[c + 2]);
                info.Vertices.push_back(v);
            }
        }
        if (info.Indices.empty()) {
            std::cout << "[WARNING] Mesh has no index data, skipping." << std::endl;
            continue;
        }
        if (info.Vertices.empty()) {
            std::cout << "[WARNING] Mesh has no vertices, skipping." << std::endl;
            continue;
        }
        const long long matIndex = mesh["primitives"][0]["material"].GetInt(-1);
        const minijson::Value& mat = doc["materials"][(size_t)(matIndex < 0 ? 0 : matIndex)];
        if (matIndex >= 0 && mat.IsObject()) {
            const minijson::Value& pbr = mat["pbrMetallicRoughness"];
            const minijson::Value& bc = pbr["baseColorFactor"];
            info.Material.AlbedoColor = bc.IsArray() ? VColor(bc[0].GetFloat(1.f), bc[1].GetFloat(1.f), bc[2].GetFloat(1.f), bc[3].GetFloat(1.f))
                                                     : VColor(1.f, 1.f, 1.f, 1.f); /* glTF defaults */
            info.Material.Metallic = pbr["metallicFactor"].GetFloat(1.f);
            info.Material.Roughness = pbr["roughnessFactor"].GetFloat(1.f);
            info.MaterialName = mat["name"].GetString();
        } else {
            std::cout << "[WARNING] Mesh has no assigned material." << std::endl;
        }
        scene->Meshes[std::to_string(mi)] = info;
    }

    std::cout << "[INFO] Importing objects" << std::endl;
    for (size_t ni = 0; ni < doc["nodes"].Size(); ni++) {
        const minijson::Value& node = doc["nodes"][ni];
        const std::string name = node["name"].GetString();
        std::cout << "[INFO] Trying to import object: " << name << std::endl;
        const long long meshIndex = node["mesh"].GetInt(-1);
        const std::string meshId = std::to_string(meshIndex);
        if (meshIndex >= 0 && !node.Has("matrix") && scene->Meshes.find(meshId) != scene->Meshes.end()) {
            VObjectInfo obj;
            obj.MeshID = meshId;
            const minijson::Value& t = node["translation"];
            const minijson::Value& s = node["scale"];
            const minijson::Value& r = node["rotation"];
            obj.Position = VVector(t[0].GetFloat(), t[1].GetFloat(), t[2].GetFloat()) * 100.f;
            obj.Scale = s.IsArray() ? VVector(s[0].GetFloat(1.f), s[1].GetFloat(1.f), s[2].GetFloat(1.f)) : VVector::ONE;
            obj.Rotation = r.IsArray() ? VQuat(r[0].GetFloat(), r[1].GetFloat(), r[2].GetFloat(), r[3].GetFloat(1.f)) : VQuat::IDENTITY;
            scene->Objects.push_back(obj);
        } else if (starts_with(name, "Light")) {
            scene->Lights.push_back(light_from_node(node));
        } else {
            std::cout << "[INFO] Skipping non geometry object." << std::endl;
        }
    }
    return scene;
}

bool VTextureLibraryImporter::Import(const std::string& jsonPath, VTextureLibrary& out) {
    try {
        const minijson::ValuePtr root = minijson::Parse(read_file(jsonPath, false));
        const minijson::Value& mats = (*root)["materials"];
        if (!mats.IsArray()) return false;
        for (size_t i = 0; i < mats.Size(); i++) {
            const minijson::Value& m = mats[i];
            if (!m["material"].IsString()) continue;
            VMaterialTextures t;
            t.TextureTiling = VVector2D(m["tiling-x"].GetFloat(100.f), m["tiling-y"].GetFloat(100.f));
            t.Albedo = m["albedo"].GetString();
            t.Normal = m["normal"].GetString();
            t.RM = m["rm"].GetString();
            out.Materials[m["material"].GetString()] = t;
        }
        return true;
    } catch (const std::exception& e) {
        std::cerr << "[ERROR] texture library: " << e.what() << std::endl;
        return false;
    }
}

}  // namespace Voxelizer
}  // namespace VolumeRaytracer

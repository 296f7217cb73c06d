/* host_c_api.cpp — C entry points of libvrt_host.so for the Python tests / bench (ctypes): the
 * Voxelizer on in-memory meshes and on .gltf files, and a .vox rewrite through the C++ reader and
 * writer.  Not part of the renderer boundary (that is include/vrt.h). */
#include <cstring>
#include <exception>
#include <string>

#include "../../../include/vrt.h"
#include "HostRenderer.h"
#include "HostSerialization.h"
#include "SceneConverter.h"
#include "VolumeConverter.h"

using namespace VolumeRaytracer;

namespace {
thread_local std::string g_error;
struct VolumeHandle {
    std::shared_ptr<Voxel::VVoxelVolume> volume;
};
}  // namespace

extern "C" {

const char* vrh_last_error(void) { return g_error.c_str(); }

/* ConvertMeshInfoToVoxelVolume on caller arrays.  positions: n_vertices*3 floats, already in the
 * importer's output space (x100, re-centred); bounds_extends: VMeshInfo::Bounds extents. */
void* vrh_convert_mesh(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_indices,
                       const float bounds_extends[3], const char* mesh_name) {
    try {
        Voxelizer::VMeshInfo info;
        info.MeshName = mesh_name ? mesh_name : "";
        info.Vertices.resize(n_vertices);
        for (size_t i = 0; i < n_vertices; i++) info.Vertices[i].Position = VVector(positions[i * 3], positions[i * 3 + 1], positions[i * 3 + 2]);
        info.Indices.assign(indices, indices + n_indices);
        info.Bounds = VAABB(VVector::ZERO, VVector(bounds_extends[0], bounds_extends[1], bounds_extends[2]));
        auto* h = new VolumeHandle;
        h->volume = Voxelizer::VVolumeConverter::ConvertMeshInfoToVoxelVolume(info, Voxelizer::VTextureLibrary());
        return h;
    } catch (const std::exception& e) {
        g_error = e.what();
        return nullptr;
    }
}

int vrh_volume_info(void* handle, int* resolution, int* size, float* extent, float* cell, float* density_scale, float* step_max) {
    if (!handle) return -1;
    const auto& v = *static_cast<VolumeHandle*>(handle)->volume;
    if (resolution) *resolution = v.GetResolution();
    if (size) *size = (int)v.GetSize();
    if (extent) *extent = v.GetVolumeExtends();
    if (cell) *cell = v.GetCellSize();
    if (density_scale) *density_scale = v.DensityScale;
    if (step_max) *step_max = v.StepMax;
    return 0;
}

int vrh_volume_copy_voxels(void* handle, vrt_voxel* out) {
    if (!handle || !out) return -1;
    const auto& vox = static_cast<VolumeHandle*>(handle)->volume->GetVoxels();
    for (size_t i = 0; i < vox.size(); i++) {
        memset(&out[i], 0, sizeof(vrt_voxel));
        out[i].material = vox[i].Material;
        out[i].density = vox[i].Density;
    }
    return 0;
}

void vrh_volume_free(void* handle) { delete static_cast<VolumeHandle*>(handle); }

/* The Voxelizer executable as a function; out_path receives the written file name. */
int vrh_voxelize_file(const char* gltf_path, const char* texlib_or_null, const char* out_or_null, char* out_path, size_t out_path_len) {
    try {
        const std::string out = Voxelizer::VoxelizeFile(gltf_path, texlib_or_null ? texlib_or_null : "", out_or_null ? out_or_null : "");
        if (out_path && out_path_len) {
            strncpy(out_path, out.c_str(), out_path_len - 1);
            out_path[out_path_len - 1] = '\0';
        }
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

/* Load a .vox scene with the C++ reader and write it back with the C++ writer. */
int vrh_vox_rewrite(const char* in_path, const char* out_path) {
    try {
        auto scene = VSerializationManager::LoadSceneFromFile(in_path);
        if (!scene) {
            g_error = std::string("cannot open ") + in_path;
            return -1;
        }
        if (!VSerializationManager::SaveToFile(*scene, out_path)) {
            g_error = std::string("cannot write ") + out_path;
            return -1;
        }
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

/* Decode a material texture file (PNG or binary PPM) the way VHipRenderer resolves VMaterial texture paths.
   Writes width/height; copies width*height*4 RGBA8 bytes when out is non-null and cap suffices.  0 / -1. */
int vrh_texture_load(const char* path, int* width, int* height, uint8_t* out, size_t cap) {
    VObjectPtr<VTexture2D> t = VTexture2D::LoadFromFile(path ? path : "");
    if (!t) {
        g_error = std::string("cannot decode ") + (path ? path : "(null)");
        return -1;
    }
    if (width) *width = (int)t->GetWidth();
    if (height) *height = (int)t->GetHeight();
    if (out) {
        if (cap < t->GetPixels().size()) {
            g_error = "buffer too small";
            return -1;
        }
        memcpy(out, t->GetPixels().data(), t->GetPixels().size());
    }
    return 0;
}

/* Sky box from a .dds cube map (VTextureFactory::LoadTextureCubeFromFile) or from a folder of six face images
   (XP/XM/YP/YM/ZP/ZM.png).  Writes the face size; copies 6*size*size*4 RGBA8 bytes (+X,-X,+Y,-Y,+Z,-Z) when out is
   non-null and cap suffices.  0 / -1. */
int vrh_cubemap_load(const char* dir, int* face_size, uint8_t* out, size_t cap) {
    const std::string where = dir ? dir : "";
    const bool dds = VTextureCube::IsDDSPath(where);
    VObjectPtr<VTextureCube> t = dds ? VTextureCube::LoadFromDDSFile(where) : VTextureCube::LoadFromFaceDirectory(where);
    if (!t) {
        g_error = (dds ? std::string("cannot read a cube map (uncompressed or BC1-BC5) from ") : std::string("cannot load six equal square faces from ")) + where;
        return -1;
    }
    if (face_size) *face_size = (int)t->GetWidth();
    if (out) {
        if (cap < t->GetPixels().size()) {
            g_error = "buffer too small";
            return -1;
        }
        memcpy(out, t->GetPixels().data(), t->GetPixels().size());
    }
    return 0;
}

/* The texture side of the drop-in boundary without a GPU: a renderer from VRendererFactory::NewRenderer() (not started), the five
   VTextureFactory functions (Renderer/Public/TextureFactory.h:32-41) through it, and VRenderer::InitializeTexture / UploadToGPU with
   base-typed arguments (Renderer.h:56-57).  dims_out[8]: 2D file w, h; cube face size; created 2D w, h; 3D depth; 3D-float depth; how many
   InitializeTexture calls the renderer saw.  0 / -1. */
namespace {
struct CountingRenderer : Renderer::VRenderer {
    int initialized = 0, uploaded = 0;
    void Render() override {}
    bool Start() override { return false; }
    void Stop() override {}
    bool IsActive() const override { return false; }
    void InitializeTexture(VObjectPtr<VTexture>) override { initialized++; }
    void UploadToGPU(VObjectPtr<VTexture>) override { uploaded++; }
    void ResizeRenderOutput(unsigned int, unsigned int) override {}
};
}  // namespace

int vrh_texture_factory_probe(const char* image_path, const char* cube_path, int* dims_out) {
    using Renderer::VTextureFactory;
    if (!dims_out) return -1;
    for (int i = 0; i < 8; i++) dims_out[i] = 0;
    auto counting = std::make_shared<CountingRenderer>();
    std::weak_ptr<Renderer::VRenderer> r = counting;
    const std::string ip = image_path ? image_path : "", cp = cube_path ? cube_path : "";
    if (!ip.empty()) {
        VObjectPtr<VTexture2D> t = VTextureFactory::LoadTexture2DFromFile(r, std::wstring(ip.begin(), ip.end()));
        if (!t) { g_error = "LoadTexture2DFromFile failed"; return -1; }
        dims_out[0] = (int)t->GetWidth();
        dims_out[1] = (int)t->GetHeight();
    }
    if (!cp.empty()) {
        VObjectPtr<VTextureCube> t = VTextureFactory::LoadTextureCubeFromFile(r, std::wstring(cp.begin(), cp.end()));
        if (!t) { g_error = "LoadTextureCubeFromFile failed"; return -1; }
        dims_out[2] = (int)t->GetWidth();
    }
    VObjectPtr<VTexture2D> c2 = VTextureFactory::CreateTexture2D(r, 3, 2, 1);
    uint8_t* px = nullptr;
    size_t n = 0;
    c2->GetPixels(0, px, &n);
    dims_out[3] = (int)c2->GetWidth();
    dims_out[4] = n == 3 * 2 * 4 && px ? (int)c2->GetHeight() : -1;
    dims_out[5] = (int)VTextureFactory::CreateTexture3D(r, 4, 5, 6, 1)->GetDepth();
    dims_out[6] = (int)VTextureFactory::CreateTexture3DFloat(r, 2, 3, 7, 1)->GetDepth();
    dims_out[7] = counting->initialized;
    /* the HIP renderer's own overrides take base-typed textures too (inactive: UploadToGPU warns and returns) */
    std::shared_ptr<Renderer::VRenderer> hip = Renderer::VRendererFactory::NewRenderer();
    hip->InitializeTexture(std::static_pointer_cast<VTexture>(c2));
    return 0;
}

}  // extern "C"

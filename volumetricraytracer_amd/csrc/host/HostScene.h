/*
 * HostScene.h — scene graph types the renderer consumes, restated without VObject / boost /
 * tick-manager machinery (out of scope: interactive editing).  What is kept:
 * the object/volume/light model, spawn semantics, the active camera / directional light, the
 * environment cube map, and change detection (volume dirty flags + a scene revision counter in
 * place of the per-frame add/remove sets, Scene.cpp:265-300).
 *
 * Reference (relative to /root/reference/VolumetricRaytracer/VolumetricRaytracer/Scene/):
 *   Public/LevelObject.h:28-52, Public/VoxelObject.h, Public/Camera.h:23-33, Public/Light.h:22-40,
 *   Public/PointLight.h:22-32, Public/SpotLight.h:22-35, Public/Scene.h:52-161,
 *   serialisation Private/Scene.cpp:314-544, Private/VoxelObject.cpp:37-71, Private/Light.cpp:17-57.
 */
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "HostCore.h"
#include "HostVoxel.h"

namespace VolumeRaytracer {

template <typename T> using VObjectPtr = std::shared_ptr<T>;

/* VTexture: the base every texture handed to VRenderer::InitializeTexture / UploadToGPU derives from
   (Core/Public/Textures/Texture.h:22-35; the renderer dispatches on the dynamic type, as VDXRenderer does with its VDXTexture* classes,
   DXRenderer.cpp:115-143). */
class VTexture {
public:
    virtual ~VTexture() = default;
    size_t GetMipCount() const { return MipCount; }
    virtual size_t GetPixelCount() = 0;
    virtual void Commit() {} /* Texture.h:31: the DX textures copy their CPU pixels into an upload heap here; host vectors need nothing */

protected:
    std::wstring AssetPath;
    size_t MipCount = 1;
};

/* VTextureCube reduced to what the miss shader reads: 6 faces (+X,-X,+Y,-Y,+Z,-Z) of RGBA8 (Core/Public/Textures/TextureCube.h:21-38). */
class VTextureCube : public VTexture {
public:
    VTextureCube(size_t faceSize, std::vector<uint8_t> rgba8) : Width(faceSize), Height(faceSize), Pixels(std::move(rgba8)) {}
    size_t GetPixelCount() override { return Width * Height * 6; }
    size_t GetWidth() const { return Width; }
    size_t GetHeight() const { return Height; }
    size_t GetArraySize() const { return 6; }
    const std::vector<uint8_t>& GetPixels() const { return Pixels; }
    /* Six square faces of equal size from <dir>/XP.png, XM.png, YP.png, YM.png, ZP.png, ZM.png (the layout of the
       reference's Resources/Skybox/ folder, the source of its Skybox.dds) in D3D cube-face order +X,-X,+Y,-Y,+Z,-Z;
       nullptr when a face is missing, undecodable, not square or of another size. */
    static std::shared_ptr<VTextureCube> LoadFromFaceDirectory(const std::string& dir);
    /* A cube map from a .dds file — what VTextureFactory::LoadTextureCubeFromFile reads (Renderer/Private/
       TextureFactory.cpp:28-67; the reference's Resources/Skybox/Skybox.dds): uncompressed 32-bit RGBA / BGRA / BGRX or
       24-bit RGB, or block-compressed BC1 / BC2 / BC3 (DXT1 / DXT3 / DXT5); legacy or DX10 header, six faces, top mip level only.
       nullptr for anything else (not a cube map, BC6H / BC7 / float formats, truncated, faces not square). */
    static std::shared_ptr<VTextureCube> LoadFromDDSFile(const std::string& path);
    /* Whether a sky box argument names a .dds file (suffix, any case) rather than a folder of face images: the one rule every
       caller uses. */
    static bool IsDDSPath(const std::string& path) {
        if (path.size() <= 4) return false;
        const char* e = path.c_str() + path.size() - 4;
        return e[0] == '.' && (e[1] | 0x20) == 'd' && (e[2] | 0x20) == 'd' && (e[3] | 0x20) == 's';
    }

private:
    size_t Width, Height;
    std::vector<uint8_t> Pixels;
};

/* VTexture2D reduced to what the closest-hit shader samples: one mip of R8G8B8A8_UNORM, row-major
   (Renderer/DX/Private/DXTexture2D.cpp:63-81).  The reference decodes image files with WIC/DDS
   (Renderer/Private/TextureFactory.cpp:58-125); here PNG (zlib), JPEG and binary PPM are read directly,
   anything else is the application's job (RegisterTexture). */
class VTexture2D : public VTexture {
public:
    VTexture2D(size_t width, size_t height, std::vector<uint8_t> rgba8) : Width(width), Height(height), Pixels(std::move(rgba8)) {}
    size_t GetPixelCount() override { return Width * Height; }
    /* Texture2D.h:26: the CPU pixel array of a mip level, for callers that fill a created texture in place */
    void GetPixels(const size_t& mipLevel, uint8_t*& outPixelArray, size_t* outArraySize) {
        outPixelArray = mipLevel == 0 ? Pixels.data() : nullptr;
        if (outArraySize) *outArraySize = mipLevel == 0 ? Pixels.size() : 0;
    }
    size_t GetWidth() const { return Width; }
    size_t GetHeight() const { return Height; }
    const std::vector<uint8_t>& GetPixels() const { return Pixels; }
    /* nullptr when the file is missing or not an 8-bit P6 image */
    static VObjectPtr<VTexture2D> LoadPPM(const std::string& path);
    /* 8-bit, non-interlaced PNG (grey, grey+alpha, RGB, RGBA, palette); nullptr otherwise */
    static VObjectPtr<VTexture2D> LoadPNG(const std::string& path);
    /* baseline / extended-sequential / progressive Huffman JPEG, 8-bit, grey or three components (JpegDecoder.cpp); nullptr otherwise */
    static VObjectPtr<VTexture2D> LoadJPEG(const std::string& path);
    /* by content: PNG, then JPEG, then PPM */
    static VObjectPtr<VTexture2D> LoadFromFile(const std::string& path);

private:
    size_t Width, Height;
    std::vector<uint8_t> Pixels;
};

/* VTexture3D / VTexture3DFloat (Core/Public/Textures/Texture3D.h, Texture3DFloat.h): host containers with the reference's surface.  The DX
   backend keeps its volume and octree-traversal textures in these (RDXVoxelVolume.cpp:156-213); this backend re-tiles volumes on the device
   itself (vrt_volume_upload*), so they exist for VTextureFactory's callers and carry no device state. */
class VTexture3D : public VTexture {
public:
    VTexture3D(size_t width, size_t height, size_t depth, size_t mipLevels) : Width(width), Height(height), Depth(depth), Pixels(width * height * depth * 4) { MipCount = mipLevels; }
    size_t GetPixelCount() override { return Width * Height * Depth; }
    void GetPixels(const size_t& mipLevel, uint8_t*& outPixelArray, size_t* outArraySize) {
        outPixelArray = mipLevel == 0 ? Pixels.data() : nullptr;
        if (outArraySize) *outArraySize = mipLevel == 0 ? Pixels.size() : 0;
    }
    size_t GetWidth() const { return Width; }
    size_t GetHeight() const { return Height; }
    size_t GetDepth() const { return Depth; }

private:
    size_t Width, Height, Depth;
    std::vector<uint8_t> Pixels;
};

class VTexture3DFloat : public VTexture {
public:
    VTexture3DFloat(size_t width, size_t height, size_t depth, size_t mipLevels) : Width(width), Height(height), Depth(depth), Pixels(width * height * depth) { MipCount = mipLevels; }
    size_t GetPixelCount() override { return Width * Height * Depth; }
    void GetPixels(const size_t& mipLevel, float*& outPixelArray, size_t* outArraySize) {
        outPixelArray = mipLevel == 0 ? Pixels.data() : nullptr;
        if (outArraySize) *outArraySize = mipLevel == 0 ? Pixels.size() : 0;
    }
    size_t GetWidth() const { return Width; }
    size_t GetHeight() const { return Height; }
    size_t GetDepth() const { return Depth; }

private:
    size_t Width, Height, Depth;
    std::vector<float> Pixels;
};

namespace Scene {

class VScene;

class VLevelObject {
public:
    virtual ~VLevelObject() = default;
    VVector Position = VVector::ZERO;
    VQuat Rotation = VQuat::IDENTITY;
    VVector Scale = VVector::ZERO; /* LevelObject.h:50 — SpawnObject always overwrites it */
};

class VVoxelObject : public VLevelObject {
public:
    void SetVoxelVolume(VObjectPtr<Voxel::VVoxelVolume> volume) { VoxelVolume = std::move(volume); }
    VObjectPtr<Voxel::VVoxelVolume> GetVoxelVolume() const { return VoxelVolume; }

private:
    VObjectPtr<Voxel::VVoxelVolume> VoxelVolume;
};

class VCamera : public VLevelObject {
public:
    float FOVAngle = 60.f;
    float NearClipPlane = 0.01f;
    float FarClipPlane = 125.f;
    float AspectRatio = 1.7777f;
};

class VLight : public VLevelObject {
public:
    float IlluminationStrength = 1.f;
    VColor Color = VColor::WHITE;
};

class VPointLight : public VLight {
public:
    float AttenuationLinear = 0.5f;
    float AttenuationExp = 0.005f;
};

class VSpotLight : public VLight {
public:
    float AttenuationLinear = 0.5f;
    float AttenuationExp = 0.005f;
    float FalloffAngle = 20.f;
    float Angle = 45.f;
};

class VScene : public std::enable_shared_from_this<VScene> {
public:
    template <typename T, class... Args>
    VObjectPtr<T> SpawnObject(const VVector& location, const VQuat& rotation, const VVector& scale, Args&&... args) {
        static_assert(std::is_base_of<VLevelObject, T>::value, "T must inherit from VLevelObject");
        VObjectPtr<T> obj = std::make_shared<T>(std::forward<Args>(args)...);
        obj->Position = location;
        obj->Rotation = rotation;
        obj->Scale = scale;
        PlacedObjects.push_back(obj);
        Revision++;
        return obj;
    }
    void DestroyObject(const VObjectPtr<VLevelObject>& obj) {
        for (size_t i = 0; i < PlacedObjects.size(); i++)
            if (PlacedObjects[i] == obj) {
                PlacedObjects.erase(PlacedObjects.begin() + (long)i);
                Revision++;
                return;
            }
    }
    void SetEnvironmentTexture(VObjectPtr<VTextureCube> texture) {
        EnvironmentTexture = std::move(texture);
        Revision++;
    }
    VObjectPtr<VTextureCube> GetEnvironmentTexture() const { return EnvironmentTexture; }
    void SetActiveSceneCamera(std::weak_ptr<VCamera> camera) { ActiveCamera = std::move(camera); }
    void SetActiveDirectionalLight(std::weak_ptr<VLight> light) { ActiveDirectionalLight = std::move(light); }
    VObjectPtr<VCamera> GetActiveCamera() const { return ActiveCamera.lock(); }
    VObjectPtr<VLight> GetActiveDirectionalLight() const { return ActiveDirectionalLight.lock(); }
    const std::vector<VObjectPtr<VLevelObject>>& GetAllPlacedObjects() const { return PlacedObjects; }
    /* distinct volumes in first-use order (the index is the renderer's volume slot) */
    std::vector<VObjectPtr<Voxel::VVoxelVolume>> GetAllRegisteredVolumes() const {
        std::vector<VObjectPtr<Voxel::VVoxelVolume>> out;
        for (const auto& o : PlacedObjects) {
            auto vo = std::dynamic_pointer_cast<VVoxelObject>(o);
            if (!vo || !vo->GetVoxelVolume()) continue;
            bool seen = false;
            for (const auto& v : out) seen = seen || v == vo->GetVoxelVolume();
            if (!seen) out.push_back(vo->GetVoxelVolume());
        }
        return out;
    }
    void PostRender() { /* Scene.cpp:568-576 clears the frame diffs; volumes clear their dirty flags */
        for (const auto& v : GetAllRegisteredVolumes()) v->PostRender();
    }
    uint64_t GetRevision() const { return Revision; }
    void Touch() { Revision++; } /* call after moving objects / lights between frames */

private:
    std::vector<VObjectPtr<VLevelObject>> PlacedObjects;
    std::weak_ptr<VCamera> ActiveCamera;
    std::weak_ptr<VLight> ActiveDirectionalLight;
    VObjectPtr<VTextureCube> EnvironmentTexture;
    uint64_t Revision = 0;
};

}  // namespace Scene
}  // namespace VolumeRaytracer

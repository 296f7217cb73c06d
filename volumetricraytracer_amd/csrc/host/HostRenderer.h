/*
 * HostRenderer.h — the reference's abstract renderer surface, verbatim in shape:
 * Renderer/Public/Renderer.h:32-66 (EVRenderMode, VRenderer — the same virtuals with the same parameter types, so that a class written
 * against this header overrides the reference's own), Renderer/Public/RendererFactory.h:24-28, Renderer/Public/TextureFactory.h:32-41.
 * The texture classes (VTexture and its four kinds) are HostScene.h's.
 */
#pragma once
#include <memory>
#include <string>
#include "HostScene.h"

namespace VolumeRaytracer {
namespace Renderer {

enum class EVRenderMode {
    Interp = 0,
    Interp_Unlit = 1,
    Interp_NoTex = 2,
    Interp_NoTex_Unlit = 3,
    Cube = 4,
    Cube_Unlit = 5,
    Cube_NoTex = 6,
    Cube_NoTex_Unlit = 7
};

class VRenderer : public std::enable_shared_from_this<VRenderer> {
public:
    virtual ~VRenderer() = default;
    virtual void Render() = 0;
    virtual bool Start() = 0;
    virtual void Stop() = 0;
    virtual bool IsActive() const = 0;
    virtual void SetSceneToRender(VObjectPtr<Scene::VScene> scene) { SceneRef = scene; }
    virtual void InitializeTexture(VObjectPtr<VTexture> texture) = 0;
    virtual void UploadToGPU(VObjectPtr<VTexture> texture) = 0;
    virtual void ResizeRenderOutput(unsigned int width, unsigned int height) = 0;
    void SetRendererMode(const EVRenderMode& renderMode) { RenderMode = renderMode; }

protected:
    std::weak_ptr<Scene::VScene> SceneRef;
    EVRenderMode RenderMode = EVRenderMode::Interp;
};

/* Renderer/Public/TextureFactory.h:32-41, the five static signatures (std::wstring paths).  The reference's bodies are DirectXTex / WIC
   (Renderer/Private/TextureFactory.cpp:28-147); these go through the build's own readers (DDS cube maps or a folder of six face images;
   PNG / JPEG / PPM) and hand the texture to renderer->InitializeTexture like theirs (TextureFactory.cpp:58,114,125,134,143).  nullptr
   (after logging) when the file cannot be read. */
class VTextureFactory {
public:
    static VObjectPtr<VTextureCube> LoadTextureCubeFromFile(std::weak_ptr<VRenderer> renderer, const std::wstring& path);
    static VObjectPtr<VTexture2D> LoadTexture2DFromFile(std::weak_ptr<VRenderer> renderer, const std::wstring& path);

    static VObjectPtr<VTexture3D> CreateTexture3D(std::weak_ptr<VRenderer> renderer, const size_t& width, const size_t& height, const size_t& depth, const size_t& mipLevels);
    static VObjectPtr<VTexture2D> CreateTexture2D(std::weak_ptr<VRenderer> renderer, const size_t& width, const size_t& height, const size_t& mipLevels);
    static VObjectPtr<VTexture3DFloat> CreateTexture3DFloat(std::weak_ptr<VRenderer> renderer, const size_t& width, const size_t& height, const size_t& depth, const size_t& mipLevels);
};

class VRendererFactory {
public:
    /* RendererFactory.cpp:23-26 returns the only backend of the platform: here the HIP one. */
    static std::shared_ptr<VRenderer> NewRenderer();
};

}  // namespace Renderer
}  // namespace VolumeRaytracer

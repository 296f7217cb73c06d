/*
 * HostRenderer.h — the reference's abstract renderer surface, verbatim in shape:
 * Renderer/Public/Renderer.h:32-66 (EVRenderMode, VRenderer), Renderer/Public/RendererFactory.h:24-28.
 * VTexture is reduced to VTextureCube (sky) and VTexture2D (material textures), HostScene.h.
 */
#pragma once
#include <memory>
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
    virtual void InitializeTexture(VObjectPtr<VTextureCube> texture) = 0;
    virtual void UploadToGPU(VObjectPtr<VTextureCube> texture) = 0;
    virtual void InitializeTexture(VObjectPtr<VTexture2D> texture) = 0;
    virtual void UploadToGPU(VObjectPtr<VTexture2D> texture) = 0;
    virtual void ResizeRenderOutput(unsigned int width, unsigned int height) = 0;
    void SetRendererMode(const EVRenderMode& renderMode) { RenderMode = renderMode; }

protected:
    std::weak_ptr<Scene::VScene> SceneRef;
    EVRenderMode RenderMode = EVRenderMode::Interp;
};

class VRendererFactory {
public:
    /* RendererFactory.cpp:23-26 returns the only backend of the platform: here the HIP one. */
    static std::shared_ptr<VRenderer> NewRenderer();
};

}  // namespace Renderer
}  // namespace VolumeRaytracer

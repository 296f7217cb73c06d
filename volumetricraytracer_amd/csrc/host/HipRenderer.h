/*
 * HipRenderer.h — `VRenderer` for MI355X: the sibling of the reference's Renderer/DX backend
 * (VDXRenderer, Renderer/DX/Public/DXRenderer.h:120-257).  It owns a vrt_ctx (include/vrt.h) and
 * does on the host what VRDXScene did: mirror the VScene into the C-ABI's flat structs every
 * frame (volumes only when dirty), then render.  There is no swap chain on a headless node: the
 * frame lands in a host float RGBA buffer the caller can read (GetFrame) — the backend-specific
 * part, like VDXRenderer::SetWindowHandle was.
 *
 * Error convention of the reference: Start() returns bool, everything else logs and returns.
 */
#pragma once
#include <map>
#include <string>
#include <vector>
#include "../../../include/vrt.h"
#include "HostRenderer.h"

namespace VolumeRaytracer {
namespace Renderer {
namespace Hip {

class VHipRenderer : public VRenderer {
public:
    VHipRenderer();
    ~VHipRenderer() override;

    void Render() override;
    bool Start() override;
    void Stop() override;
    bool IsActive() const override { return Ctx != nullptr; }
    void InitializeTexture(VObjectPtr<VTextureCube> texture) override;
    void UploadToGPU(VObjectPtr<VTextureCube> texture) override;
    void InitializeTexture(VObjectPtr<VTexture2D> texture) override;
    void UploadToGPU(VObjectPtr<VTexture2D> texture) override;
    void ResizeRenderOutput(unsigned int width, unsigned int height) override;

    /* Material textures are looked up by the path a VMaterial names (the reference's path-keyed table,
       RDXScene.cpp:771-800, 905-925).  An application that decodes its own images registers them here; paths
       nobody registered are tried once as PNG / binary PPM files and otherwise stay unbound (logged). */
    void RegisterTexture(const std::string& path, VObjectPtr<VTexture2D> texture);

    /* backend-specific */
    void SetDevices(const std::vector<int>& hipOrdinals) { Devices = hipOrdinals; } /* before Start(); default {0} */
    /* The newest finished frame: Width*Height float RGBA (null before the first one).  With frames in flight it points
       into the slot's pinned buffer and stays valid for FramesInFlight - 1 further Render() calls. */
    const float* GetFramePixels() const { return FramePixels; }
    size_t GetFramePixelCount() const { return FramePixelCount; }
    const std::vector<float>& GetFrame() const { return Frame; }                     /* FramesInFlight == 1 only */
    unsigned GetWidth() const { return Width; }
    unsigned GetHeight() const { return Height; }
    bool GetLastTiming(vrt_timing& out) const;
    /* march contract knobs; defaults follow the smallest cell of the scene */
    int MaxSteps = 255;       /* Raytracing.hlsl:229: the budget at the reference's largest resolution, 8; doubled per
                                 resolution step beyond it (cells half the size need twice the positions) */
    /* Device format of the volumes: VRT_FORMAT_F32, or VRT_FORMAT_TEXEL16 = the reference's own 16-bit volume texel
       (VDXVoxelVolume::EncodeVoxel, RDXVoxelVolume.cpp:399-421): the march then sees exactly the DXR backend's field */
    int VolumeFormat = VRT_FORMAT_F32;
    float Relaxation = 1.7f;  /* vrt_params::k_relax: over-relaxed sphere-trace with the sphere-overlap fallback; 1 = plain */
    bool Shadows = true;      /* the reference always casts the directional shadow ray */
    int MaxBounces = 2;       /* MAX_RAY_RECURSION_DEPTH 3 = primary + 2 mirror bounces (RaytracingHlsl.h:32) */
    int DataPath = VRT_PATH_AUTO;
    /* 1: Render() returns with the finished frame (vrt_render).  2..3: frames in flight like the reference's swap chain
       (FrameCount 3, DXConstants.cpp:23): Render() enqueues (vrt_render_begin) and GetFrame() lags by FramesInFlight - 1
       frames until Flush() collects what is still in flight.  Single-device only. */
    int FramesInFlight = 1;
    void Flush();

private:
    bool SyncWithScene(Scene::VScene& scene);
    vrt_ctx* Ctx = nullptr;
    std::vector<int> Devices{0};
    unsigned Width = 1024, Height = 576; /* Win32Window.cpp:218-219 */
    std::vector<float> Frame;
    const float* FramePixels = nullptr;
    size_t FramePixelCount = 0;
    std::vector<const Voxel::VVoxelVolume*> Uploaded; /* per slot */
    const VTextureCube* UploadedEnv = nullptr;
    struct TextureEntry {
        VObjectPtr<VTexture2D> Texture; /* null: lookup failed, do not retry */
        int Id = -1;                    /* index in the device table once uploaded */
    };
    std::map<std::string, TextureEntry> Textures;
    std::map<const VTexture2D*, int> TextureIds;
    int ResolveTexture(const std::string& path);
    float MinCell = 1.f;
    int MaxResolution = 0;
    int UploadedFormat = -1;
    void Collect(int slot);
    unsigned long long FrameIndex = 0;
    bool SlotBusy[VRT_FRAMES_IN_FLIGHT] = {};
    size_t SlotPixels[VRT_FRAMES_IN_FLIGHT] = {};
};

}  // namespace Hip
}  // namespace Renderer
}  // namespace VolumeRaytracer

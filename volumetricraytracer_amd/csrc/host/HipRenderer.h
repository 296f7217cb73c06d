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
#include <functional>
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
    /* Renderer.h:56-57: dispatch on the texture's dynamic type (cube map -> the sky, 2D -> the material-texture table; the 3D kinds
       carry no device state in this backend).  The typed overloads below are what they call. */
    void InitializeTexture(VObjectPtr<VTexture> texture) override;
    void UploadToGPU(VObjectPtr<VTexture> texture) override;
    void UploadToGPU(VObjectPtr<VTextureCube> texture);
    void UploadToGPU(VObjectPtr<VTexture2D> texture);
    void ResizeRenderOutput(unsigned int width, unsigned int height) override;

    /* Material textures are looked up by the path a VMaterial names (the reference's path-keyed table,
       RDXScene.cpp:771-800, 905-925).  An application that decodes its own images registers them here; paths
       nobody registered are tried once as PNG / binary PPM files and otherwise stay unbound (logged). */
    void RegisterTexture(const std::string& path, VObjectPtr<VTexture2D> texture);

    /* backend-specific */
    void SetDevices(const std::vector<int>& hipOrdinals) { Devices = hipOrdinals; } /* before Start(); default {0} */
    /* Pixel format of the frames handed to the host.  BGRA8 is the reference's own back buffer (DXGI_FORMAT_B8G8R8A8_UNORM,
       DXConstants.cpp:21, DXRenderer.cpp:1322) and the default; RGBA8 the same bytes with R first; Float4 keeps the float
       channels (16 B per pixel: four times the bytes over PCIe) — what the parity tests compare. */
    enum class EFrameFormat { Float4, RGBA8, BGRA8 };
    EFrameFormat FrameFormat = EFrameFormat::BGRA8;
    size_t BytesPerPixel() const { return FrameFormat == EFrameFormat::Float4 ? 16 : 4; }
    /* The newest finished frame: Width*Height pixels of FrameFormat (null before the first one).  With frames in flight it
       points into the slot's pinned buffer and stays valid for FramesInFlight - 1 further Render() calls. */
    const void* GetFrameData() const { return FramePixels; }
    size_t GetFrameByteCount() const { return FrameBytes; }
    const float* GetFramePixels() const { return FrameFormat == EFrameFormat::Float4 ? static_cast<const float*>(FramePixels) : nullptr; }
    unsigned GetWidth() const { return Width; }
    unsigned GetHeight() const { return Height; }
    bool GetLastTiming(vrt_timing& out) const;
    /* march contract knobs; defaults follow the smallest cell of the scene */
    int MaxSteps = 255;       /* Raytracing.hlsl:229: the budget at the reference's largest resolution, 8; doubled per
                                 resolution step beyond it (cells half the size need twice the positions) */
    /* Device format of the volumes.  Default VRT_FORMAT_TEXEL16 = the reference's own 16-bit volume texel (sign + 15-bit |d| * 100,
       VDXVoxelVolume::EncodeVoxel, RDXVoxelVolume.cpp:399-421): the march then sees exactly the field the DXR backend's GPU sees —
       what a drop-in for that backend should render (frames over the unquantised floats differ from the reference's in 1.6 % of a
       surface's pixels by more than one 8-bit step: the texel's 0.01 quantum).  VRT_FORMAT_F32 keeps the caller's
       floats (3 % faster, bench.py's format). */
    int VolumeFormat = VRT_FORMAT_TEXEL16;
    /* The reference binds 1x1 DEFAULT textures to every material slot that names no image (VRDXScene::AllocateDefaultTextures,
       RDXScene.cpp:241-260): albedo white and RM (1, 1, 0) are exact identities, but the default normal texel, VColor(0.5, 0.5, 1)
       stored as 8 bits, is (127, 127, 255) and decodes to (-0.0039, -0.0039, 1): in the textured render modes (Interp — the
       reference's DEFAULT mode —, Interp_Unlit, Cube, Cube_Unlit) it tilts every normal by 0.3 degrees, which moves 2-10 % of a surface's
       pixels by more than one 8-bit step.  true (default): materials without a normal map get that texel, so that
       the frames are the reference's; false: unbound slots are exact identities (and a scene without textures keeps the lean kernel). */
    bool ReferenceDefaultTextures = true;
    /* Two more of the reference's artefacts, on by default for the same reason (measured against the literal restatement of its shaders,
       DESIGN.md section 5.0): it never normalises its camera direction, so its closest-hit shader evaluates the BRDF with a view vector of
       length 1 ... 1.55 and backs the camera ray's secondary rays off by 0.1 times that (VRT_FLAG_REFERENCE_VIEW_VECTOR: 9.5 % of a mirror
       scene's surface pixels by more than one 8-bit step); and its normal's taps beyond the volume texture read 0
       (VRT_FLAG_REFERENCE_BOUNDARY_TEXELS: surfaces within a cell of their volume's box).  false: unit view vector / clamped cell. */
    bool ReferenceViewVector = true;
    bool ReferenceBoundaryTexels = true;
    float Relaxation = 1.7f;  /* vrt_params::k_relax: over-relaxed sphere-trace with the sphere-overlap fallback; 1 = plain */
    bool Shadows = true;      /* the reference always casts the directional shadow ray */
    int MaxBounces = 2;       /* MAX_RAY_RECURSION_DEPTH 3 = primary + 2 mirror bounces (RaytracingHlsl.h:32) */
    int DataPath = VRT_PATH_AUTO;
    /* 2..3 (default 3, the reference's swap chain: FrameCount 3, DXConstants.cpp:23, fence pacing DXRenderer.cpp:974-989):
       Render() enqueues the frame (vrt_render_begin) and returns; GetFrameData() lags by FramesInFlight - 1 frames until
       Flush() collects what is still in flight.  1: Render() returns with the finished frame (vrt_render).  Contexts over
       several devices always render synchronously. */
    int FramesInFlight = 3;
    void Flush();
    /* n_frames frames of the application's animation with ONE march launch: tick(f) moves objects / lights / camera for frame
       f (may be empty: a standing scene), every frame's scene state is mirrored, the block is marched and copied to pinned host
       memory.  GetBlockFrame(f): frame f in FrameFormat, valid until the next RenderBlock / Stop.  n_frames <= 256. */
    bool RenderBlock(int n_frames, const std::function<void(int)>& tick);
    const void* GetBlockFrame(int f) const { return (BlockFrames && f >= 0 && f < BlockFrameCount) ? BlockFrames + (size_t)f * BlockFrameBytes : nullptr; }

private:
    bool SyncWithScene(Scene::VScene& scene);
    bool FillSceneStruct(Scene::VScene& scene, vrt_scene& out);
    vrt_params MakeParams(Scene::VScene& scene) const;
    std::vector<vrt_scene> BlockScenes;
    const unsigned char* BlockFrames = nullptr;
    int BlockFrameCount = 0;
    size_t BlockFrameBytes = 0;
    vrt_ctx* Ctx = nullptr;
    std::vector<int> Devices{0};
    unsigned Width = 1024, Height = 576; /* Win32Window.cpp:218-219 */
    std::vector<float> Frame; /* synchronous frames (FramesInFlight 1, or several devices) */
    const void* FramePixels = nullptr;
    size_t FrameBytes = 0;
    std::vector<const Voxel::VVoxelVolume*> Uploaded; /* per slot */
    const VTextureCube* UploadedEnv = nullptr;
    struct TextureEntry {
        VObjectPtr<VTexture2D> Texture; /* null: lookup failed, do not retry */
        int Id = -1;                    /* index in the device table once uploaded */
    };
    std::map<std::string, TextureEntry> Textures;
    std::map<const VTexture2D*, int> TextureIds;
    int ResolveTexture(const std::string& path);
    VObjectPtr<VTexture2D> DefaultNormalTexture; /* the reference's 1x1 default normal texel (127, 127, 255, 255) */
    float MinCell = 1.f;
    int MaxResolution = 0;
    int UploadedFormat = -1;
    void Collect(int slot);
    unsigned long long FrameIndex = 0;
    bool SlotBusy[VRT_FRAMES_IN_FLIGHT] = {};
    size_t SlotBytes[VRT_FRAMES_IN_FLIGHT] = {};
};

}  // namespace Hip
}  // namespace Renderer
}  // namespace VolumeRaytracer

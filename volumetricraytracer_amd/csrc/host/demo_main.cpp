/*
 * vrt_demo — headless stand-in for the reference application: builds the demo scene of
 * App/Private/RendererEngineInstance.cpp:232-316 (camera (300,0,100) yaw 180°, directional light
 * yaw 45° / pitch −30° strength 6, two 64^3 SDF spheres r=40 / r=20 orbiting as in
 * OnEngineUpdate :76-109; Monkey.vox and Skybox.dds are not in the checkout, so an optional .vox
 * scene can be given and the sky is procedural) and drives it through the VRenderer interface
 * exactly like VEngine::EngineLoop does (Engine/Private/Engine.cpp:201-232):
 * tick → Renderer->Render() → post-render.  Writes the last frame as a PPM.
 *
 *   vrt_demo [--frames N] [--size WxH] [--scene file.vox] [--out frame.ppm] [--mode 0..7 (EVRenderMode)] [--in-flight 1..3] [--skybox dir-with-XP..ZM.png | cube.dds]
 *            [--format bgra8|rgba8|float]   frame format handed to the host; default bgra8, the reference's back buffer (DXConstants.cpp:21)
 *            [--volumes texel16|f32]        device format of the volumes; default texel16, the reference's own 16-bit volume texel
 *            [--identity-defaults]          none of the reference's artefacts: unbound material slots are exact identities instead of its 1x1 default texels (its normal texel tilts by 0.3 degrees), unit view vector, clamped normal taps
 *            [--block N]                    N frames of the animation per RenderBlock call (ONE march launch per N frames) instead of one Render() per frame
 */
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "HipRenderer.h"
#include "HostSerialization.h"

using namespace VolumeRaytracer;

static VObjectPtr<Scene::VVoxelObject> InitSphere(Scene::VScene& scene, float radius, const VMaterial& material) {
    auto volume = std::make_shared<Voxel::VVoxelVolume>(6, 100.f);
    const int n = (int)volume->GetSize();
    for (int x = 0; x < n; x++)
        for (int y = 0; y < n; y++)
            for (int z = 0; z < n; z++) {
                const VIntVector idx(x, y, z);
                const float density = volume->VoxelIndexToRelativePosition(idx).Length() - radius; /* VSphere, DensityGenerator.cpp:33-36 */
                Voxel::VVoxel v;
                v.Material = density <= 0 ? 1 : 0;
                v.Density = density;
                volume->SetVoxel(idx, v);
            }
    volume->SetMaterial(material);
    auto obj = scene.SpawnObject<Scene::VVoxelObject>(VVector::ZERO, VQuat::IDENTITY, VVector::ONE);
    obj->SetVoxelVolume(volume);
    return obj;
}

static VObjectPtr<VTextureCube> ProceduralSky(size_t S) {
    std::vector<uint8_t> px(6 * S * S * 4);
    const float tint[6][3] = {{1.f, .85f, .8f}, {.8f, .85f, 1.f}, {.85f, 1.f, .8f}, {1.f, .8f, 1.f}, {.6f, .75f, 1.f}, {.55f, .5f, .45f}};
    for (size_t f = 0; f < 6; f++)
        for (size_t y = 0; y < S; y++)
            for (size_t x = 0; x < S; x++) {
                const float v = ((float)y + 0.5f) / (float)S;
                const float g = 0.35f + 0.6f * (1.f - v);
                uint8_t* p = &px[((f * S + y) * S + x) * 4];
                for (int c = 0; c < 3; c++) p[c] = (uint8_t)std::fmin(255.f, g * tint[f][c] * 255.f + 0.5f);
                p[3] = 255;
            }
    return std::make_shared<VTextureCube>(S, px);
}

int main(int argc, char** argv) {
    int frames = 60;
    unsigned W = 1024, H = 576;
    std::string scenePath, skyboxDir, outPath = "vrt_demo.ppm";
    bool identityDefaults = false;
    int mode = 0, inFlight = 3, block = 0; /* three frames in flight: the reference's swap chain (FrameCount, DXConstants.cpp:23) */
    std::string format = "bgra8", volumes = "texel16";
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--size") && i + 1 < argc) sscanf(argv[++i], "%ux%u", &W, &H);
        else if (!strcmp(argv[i], "--scene") && i + 1 < argc) scenePath = argv[++i];
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) outPath = argv[++i];
        else if (!strcmp(argv[i], "--mode") && i + 1 < argc) mode = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--in-flight") && i + 1 < argc) inFlight = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--skybox") && i + 1 < argc) skyboxDir = argv[++i];
        else if (!strcmp(argv[i], "--format") && i + 1 < argc) format = argv[++i];
        else if (!strcmp(argv[i], "--block") && i + 1 < argc) block = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--volumes") && i + 1 < argc) volumes = argv[++i];
        else if (!strcmp(argv[i], "--identity-defaults")) identityDefaults = true;
    }

    std::shared_ptr<Renderer::VRenderer> renderer = Renderer::VRendererFactory::NewRenderer();
    if (!renderer->Start()) {
        fprintf(stderr, "No suitable GPU found in this system!\n"); /* Engine.cpp:56 */
        return 1;
    }
    renderer->ResizeRenderOutput(W, H);
    if (mode >= 0 && mode <= 7) renderer->SetRendererMode((Renderer::EVRenderMode)mode); /* the reference's F1-F8 switch */

    VObjectPtr<Scene::VScene> scene;
    if (!scenePath.empty()) {
        scene = VSerializationManager::LoadSceneFromFile(scenePath);
        if (!scene) {
            fprintf(stderr, "cannot load %s\n", scenePath.c_str());
            return 1;
        }
    } else {
        scene = std::make_shared<Scene::VScene>();
    }
    auto camera = scene->SpawnObject<Scene::VCamera>(VVector(300.f, 0.f, 100.f), VQuat::FromAxisAngle(VVector::UP, 3.14159265f), VVector::ONE);
    if (!scene->GetActiveDirectionalLight()) {
        auto light = scene->SpawnObject<Scene::VLight>(
            VVector::ZERO, VQuat::FromAxisAngle(VVector::UP, 45.f * 3.14159265f / 180.f) * VQuat::FromAxisAngle(VVector::RIGHT, -30.f * 3.14159265f / 180.f), VVector::ONE);
        light->IlluminationStrength = 6.f;
        scene->SetActiveDirectionalLight(light);
    }
    const bool skyIsDDS = VTextureCube::IsDDSPath(skyboxDir); /* the reference's Skybox.dds */
    VObjectPtr<VTextureCube> sky = skyboxDir.empty() ? nullptr : (skyIsDDS ? VTextureCube::LoadFromDDSFile(skyboxDir) : VTextureCube::LoadFromFaceDirectory(skyboxDir));
    if (!skyboxDir.empty() && !sky) fprintf(stderr, "cannot load a sky box from %s; using the procedural one\n", skyboxDir.c_str());
    scene->SetEnvironmentTexture(sky ? sky : ProceduralSky(256));
    scene->SetActiveSceneCamera(camera);
    VMaterial material;
    material.AlbedoColor = VColor::RED;
    material.Roughness = 0.1f;
    material.Metallic = 0.6f;
    auto sphere1 = InitSphere(*scene, 40.f, material);
    material.AlbedoColor = VColor::BLUE;
    auto sphere2 = InitSphere(*scene, 20.f, material);
    const VVector rel1(200.f, 0.f, 100.f), rel2(100.f, 0.f, 200.f);
    renderer->SetSceneToRender(scene);

    auto* hip = dynamic_cast<Renderer::Hip::VHipRenderer*>(renderer.get());
    using Fmt = Renderer::Hip::VHipRenderer::EFrameFormat;
    if (hip && inFlight >= 1 && inFlight <= 3) hip->FramesInFlight = inFlight;
    if (hip) hip->FrameFormat = format == "float" ? Fmt::Float4 : (format == "rgba8" ? Fmt::RGBA8 : Fmt::BGRA8);
    if (hip) hip->VolumeFormat = volumes == "f32" ? VRT_FORMAT_F32 : VRT_FORMAT_TEXEL16;
    if (hip) hip->ReferenceDefaultTextures = hip->ReferenceViewVector = hip->ReferenceBoundaryTexels = !identityDefaults;
    double kernel_ms = 0.0;
    auto tick = [&](int f) {                                                     /* TickEngineInstance */
        const float dt = 1.f / 60.f, angle = (float)f * dt * 0.5f;
        sphere1->Position = VQuat::FromAxisAngle(VVector::UP, angle) * rel1;
        sphere2->Position = VQuat::FromAxisAngle(VVector::RIGHT, angle) * rel2;
        scene->Touch();
    };
    /* one untimed frame / block first: the pinned frame buffers and the device buffers of this size are allocated by the first call */
    if (hip && block > 0) {
        if (!hip->RenderBlock(frames < block ? frames : block, tick)) return 1;
    } else {
        tick(0);
        renderer->Render();
        if (hip) hip->Flush();
    }
    scene->PostRender();
    const auto t0 = std::chrono::steady_clock::now();
    if (hip && block > 0) {
        /* the same animation, `block` frames per call: per-frame scene state, ONE march launch per block */
        for (int f0 = 0; f0 < frames; f0 += block) {
            const int n = frames - f0 < block ? frames - f0 : block;
            if (!hip->RenderBlock(n, [&](int f) { tick(f0 + f); })) return 1;
            scene->PostRender();
            vrt_timing tm;
            if (hip->GetLastTiming(tm)) kernel_ms += tm.kernel_ms;
        }
    } else {
        for (int f = 0; f < frames; f++) {
            tick(f);
            renderer->Render();                                                     /* Engine.cpp:212 */
            scene->PostRender();                                                    /* :214 */
            vrt_timing tm;
            if (hip && inFlight <= 1 && hip->GetLastTiming(tm)) kernel_ms += tm.kernel_ms; /* (asking a frame in flight for its time would wait for it) */
        }
        if (hip) hip->Flush(); /* collect the frames still in flight: GetFrameData() is the last frame again */
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d frames %ux%u %s%s: %.3f ms/frame wall (%.0f frames/s)", frames, W, H, format.c_str(),
           block > 0 ? (", RenderBlock of " + std::to_string(block)).c_str() : (", " + std::to_string(inFlight) + " in flight").c_str(), wall / frames * 1e3, frames / wall);
    if (kernel_ms > 0.0) printf(", march kernel %.3f ms/%s", kernel_ms / (block > 0 ? (frames + block - 1) / block : frames), block > 0 ? "block" : "frame");
    printf("\n");

    if (hip && hip->GetFrameData()) {
        FILE* fp = fopen(outPath.c_str(), "wb");
        if (fp) {
            fprintf(fp, "P6 %u %u 255\n", W, H);
            const float* fr = hip->GetFramePixels();
            const unsigned char* by = static_cast<const unsigned char*>(hip->GetFrameData());
            const bool bgra = hip->FrameFormat == Fmt::BGRA8;
            for (size_t i = 0; i < (size_t)W * H; i++) {
                unsigned char rgb[3];
                for (int c = 0; c < 3; c++)
                    rgb[c] = fr ? (unsigned char)(std::fmin(std::fmax(fr[i * 4 + c], 0.f), 1.f) * 255.f + 0.5f) : by[i * 4 + (bgra ? 2 - c : c)];
                fwrite(rgb, 1, 3, fp);
            }
            fclose(fp);
            printf("wrote %s\n", outPath.c_str());
        }
    }
    renderer->Stop();
    return 0;
}

#include "HipRenderer.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define V_LOG_ERROR(msg) fprintf(stderr, "[VHipRenderer][error] %s\n", (msg))
#define V_LOG_WARNING(msg) fprintf(stderr, "[VHipRenderer][warning] %s\n", (msg))

namespace VolumeRaytracer {

VObjectPtr<VTexture2D> VTexture2D::LoadPPM(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return nullptr;
    auto token = [&](char* buf, size_t n) { /* whitespace-separated header token, '#' comments skipped */
        int c = fgetc(f);
        while (c != EOF) {
            if (c == '#') {
                while (c != EOF && c != '\n') c = fgetc(f);
            } else if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
                c = fgetc(f);
            } else {
                break;
            }
        }
        size_t i = 0;
        while (c != EOF && c != ' ' && c != '\t' && c != '\n' && c != '\r' && i + 1 < n) {
            buf[i++] = (char)c;
            c = fgetc(f);
        }
        buf[i] = 0;
        return i > 0;
    };
    char magic[8], ws[16], hs[16], ms[16];
    VObjectPtr<VTexture2D> out;
    if (token(magic, sizeof magic) && !strcmp(magic, "P6") && token(ws, sizeof ws) && token(hs, sizeof hs) && token(ms, sizeof ms)) {
        const long w = atol(ws), h = atol(hs), maxv = atol(ms);
        if (w > 0 && h > 0 && w <= 16384 && h <= 16384 && maxv == 255) {
            std::vector<uint8_t> rgb((size_t)w * h * 3), rgba((size_t)w * h * 4);
            if (fread(rgb.data(), 1, rgb.size(), f) == rgb.size()) {
                for (size_t i = 0; i < (size_t)w * h; i++) {
                    rgba[i * 4 + 0] = rgb[i * 3 + 0];
                    rgba[i * 4 + 1] = rgb[i * 3 + 1];
                    rgba[i * 4 + 2] = rgb[i * 3 + 2];
                    rgba[i * 4 + 3] = 255;
                }
                out = std::make_shared<VTexture2D>((size_t)w, (size_t)h, std::move(rgba));
            }
        }
    }
    fclose(f);
    return out;
}

namespace Renderer {

std::shared_ptr<VRenderer> VRendererFactory::NewRenderer() { return std::make_shared<Hip::VHipRenderer>(); }

namespace Hip {

namespace {
bool ok(int rc, const char* what) {
    if (rc == VRT_OK) return true;
    fprintf(stderr, "[VHipRenderer][error] %s: %s (%d)\n", what, vrt_strerror(rc), rc);
    return false;
}
}  // namespace

VHipRenderer::VHipRenderer() = default;
VHipRenderer::~VHipRenderer() { Stop(); }

bool VHipRenderer::Start() {
    if (Ctx) return true;
    if (!ok(vrt_create(&Ctx, (int)Devices.size(), Devices.data()), "vrt_create")) {
        Ctx = nullptr;
        return false; /* the engine shows its "no suitable GPU" message and shuts down (Engine.cpp:54-58) */
    }
    return true;
}

void VHipRenderer::Stop() {
    if (!Ctx) return;
    vrt_destroy(Ctx);
    Ctx = nullptr;
    Uploaded.clear();
    UploadedEnv = nullptr;
    TextureIds.clear();
    for (auto& kv : Textures) kv.second.Id = -1;
}

void VHipRenderer::ResizeRenderOutput(unsigned int width, unsigned int height) {
    if (width == 0 || height == 0) return;
    Width = width;
    Height = height;
}

void VHipRenderer::InitializeTexture(VObjectPtr<VTextureCube>) { /* nothing to create ahead of the upload */ }

void VHipRenderer::UploadToGPU(VObjectPtr<VTextureCube> texture) {
    if (!Ctx) {
        V_LOG_WARNING("UploadToGPU on an inactive renderer");
        return;
    }
    if (!texture) {
        if (ok(vrt_env_upload(Ctx, 0, nullptr), "vrt_env_upload")) UploadedEnv = nullptr;
        return;
    }
    if (texture->GetPixels().size() != texture->GetWidth() * texture->GetWidth() * 6 * 4) {
        V_LOG_ERROR("cube map must hold 6 square RGBA8 faces");
        return;
    }
    if (ok(vrt_env_upload(Ctx, (int)texture->GetWidth(), texture->GetPixels().data()), "vrt_env_upload")) UploadedEnv = texture.get();
}

void VHipRenderer::InitializeTexture(VObjectPtr<VTexture2D>) { /* nothing to create ahead of the upload */ }

void VHipRenderer::UploadToGPU(VObjectPtr<VTexture2D> texture) {
    if (!Ctx) {
        V_LOG_WARNING("UploadToGPU on an inactive renderer");
        return;
    }
    if (!texture || TextureIds.count(texture.get())) return;
    if (texture->GetPixels().size() != texture->GetWidth() * texture->GetHeight() * 4 || texture->GetWidth() == 0) {
        V_LOG_ERROR("2D texture must hold width*height RGBA8 texels");
        return;
    }
    const int id = (int)TextureIds.size();
    if (id >= VRT_MAX_TEXTURES) {
        V_LOG_ERROR("too many material textures");
        return;
    }
    if (ok(vrt_texture_upload(Ctx, id, (int)texture->GetWidth(), (int)texture->GetHeight(), texture->GetPixels().data()), "vrt_texture_upload"))
        TextureIds[texture.get()] = id;
}

void VHipRenderer::RegisterTexture(const std::string& path, VObjectPtr<VTexture2D> texture) {
    TextureEntry e;
    e.Texture = std::move(texture);
    Textures[path] = e;
}

/* Device texture id of the image a material path names, -1 = unbound. */
int VHipRenderer::ResolveTexture(const std::string& path) {
    if (path.empty()) return -1;
    auto it = Textures.find(path);
    if (it == Textures.end()) {
        TextureEntry e;
        e.Texture = VTexture2D::LoadPPM(path);
        if (!e.Texture) fprintf(stderr, "[VHipRenderer][warning] material texture '%s' is neither registered nor a binary PPM; left unbound\n", path.c_str());
        it = Textures.emplace(path, e).first;
    }
    if (!it->second.Texture) return -1;
    UploadToGPU(it->second.Texture);
    auto id = TextureIds.find(it->second.Texture.get());
    return id == TextureIds.end() ? -1 : id->second;
}

bool VHipRenderer::SyncWithScene(Scene::VScene& scene) {
    /* geometry: upload new / dirty volumes, free vanished slots (VRDXScene::UpdateSceneGeometry) */
    const auto volumes = scene.GetAllRegisteredVolumes();
    if (volumes.size() > VRT_MAX_VOLUMES) {
        V_LOG_ERROR("more than 20 distinct volumes (MaxAllowedObjectData)");
        return false;
    }
    if (Uploaded.size() < volumes.size()) Uploaded.resize(volumes.size(), nullptr);
    MinCell = 0.f;
    for (size_t slot = 0; slot < volumes.size(); slot++) {
        const Voxel::VVoxelVolume& v = *volumes[slot];
        if (Uploaded[slot] != &v || v.IsDirty()) {
            static_assert(sizeof(Voxel::VVoxel) == sizeof(vrt_voxel), "VVoxel must match the wire record");
            if (!ok(vrt_volume_upload_voxels(Ctx, (int)slot, v.GetResolution(), v.GetVolumeExtends(),
                                             reinterpret_cast<const vrt_voxel*>(v.GetVoxels().data())), "vrt_volume_upload_voxels"))
                return false;
            Uploaded[slot] = &v;
        }
        const VMaterial& m = v.GetMaterial();
        vrt_material mat = {{m.AlbedoColor.R, m.AlbedoColor.G, m.AlbedoColor.B, m.AlbedoColor.A}, m.Roughness, m.Metallic};
        if (!ok(vrt_volume_set_material(Ctx, (int)slot, &mat), "vrt_volume_set_material")) return false;
        if (!ok(vrt_volume_set_metric(Ctx, (int)slot, v.DensityScale, v.StepMax), "vrt_volume_set_metric")) return false;
        const int ta = ResolveTexture(m.AlbedoTexturePath), tn = ResolveTexture(m.NormalTexturePath), tr = ResolveTexture(m.RMTexturePath);
        const float su = m.TextureScale.X != 0.f ? m.TextureScale.X : 100.f, sv = m.TextureScale.Y != 0.f ? m.TextureScale.Y : 100.f;
        if (!ok(vrt_volume_set_textures(Ctx, (int)slot, ta, tn, tr, su, sv), "vrt_volume_set_textures")) return false;
        MinCell = (MinCell == 0.f || v.GetCellSize() < MinCell) ? v.GetCellSize() : MinCell;
    }
    for (size_t slot = volumes.size(); slot < Uploaded.size(); slot++)
        if (Uploaded[slot]) {
            ok(vrt_volume_free(Ctx, (int)slot), "vrt_volume_free");
            Uploaded[slot] = nullptr;
        }
    if (MinCell == 0.f) MinCell = 1.f;

    const VTextureCube* env = scene.GetEnvironmentTexture().get();
    if (env != UploadedEnv) UploadToGPU(scene.GetEnvironmentTexture());

    /* scene constants + instance list (UpdateSceneConstantBuffer, BuildTopLevelAccelerationStructures) */
    vrt_scene s;
    memset(&s, 0, sizeof s);
    const VObjectPtr<Scene::VCamera> cam = scene.GetActiveCamera();
    if (!cam) {
        V_LOG_ERROR("scene has no active camera");
        return false;
    }
    s.cam_position[0] = cam->Position.X; s.cam_position[1] = cam->Position.Y; s.cam_position[2] = cam->Position.Z;
    s.cam_rotation[0] = cam->Rotation.x; s.cam_rotation[1] = cam->Rotation.y; s.cam_rotation[2] = cam->Rotation.z; s.cam_rotation[3] = cam->Rotation.w;
    s.cam_fov_deg = cam->FOVAngle;
    s.cam_near = cam->NearClipPlane;
    s.cam_far = cam->FarClipPlane;
    if (const VObjectPtr<Scene::VLight> dl = scene.GetActiveDirectionalLight()) {
        const VVector f = dl->Rotation.GetForwardVector(); /* RDXScene.cpp:720 */
        s.light_dir[0] = f.X; s.light_dir[1] = f.Y; s.light_dir[2] = f.Z;
        s.light_strength = dl->IlluminationStrength;
    } else {
        s.light_dir[0] = 1.f; /* no directional light: zero strength */
    }
    for (const auto& o : scene.GetAllPlacedObjects()) {
        if (auto pl = std::dynamic_pointer_cast<Scene::VPointLight>(o)) {
            if (s.n_point_lights >= VRT_MAX_POINT_LIGHTS) continue;
            vrt_point_light& L = s.point_lights[s.n_point_lights++];
            L.position[0] = pl->Position.X; L.position[1] = pl->Position.Y; L.position[2] = pl->Position.Z;
            L.color[0] = pl->Color.R; L.color[1] = pl->Color.G; L.color[2] = pl->Color.B;
            L.intensity = pl->IlluminationStrength;
            L.att_linear = pl->AttenuationLinear;
            L.att_exp = pl->AttenuationExp;
        } else if (auto sl = std::dynamic_pointer_cast<Scene::VSpotLight>(o)) {
            if (s.n_spot_lights >= VRT_MAX_SPOT_LIGHTS) continue;
            vrt_spot_light& L = s.spot_lights[s.n_spot_lights++];
            const VVector f = sl->Rotation.GetForwardVector();
            L.position[0] = sl->Position.X; L.position[1] = sl->Position.Y; L.position[2] = sl->Position.Z;
            L.forward[0] = f.X; L.forward[1] = f.Y; L.forward[2] = f.Z;
            L.color[0] = sl->Color.R; L.color[1] = sl->Color.G; L.color[2] = sl->Color.B;
            L.intensity = sl->IlluminationStrength;
            L.att_linear = sl->AttenuationLinear;
            L.att_exp = sl->AttenuationExp;
            L.cos_angle = std::cos(VMathHelpers::ToRadians(sl->Angle * 0.5f));           /* DXLightFactory.cpp:46 */
            L.cos_falloff_angle = std::cos(VMathHelpers::ToRadians(sl->FalloffAngle * 0.5f)); /* :47 */
        } else if (auto vo = std::dynamic_pointer_cast<Scene::VVoxelObject>(o)) {
            if (!vo->GetVoxelVolume()) continue;
            if (s.n_instances >= VRT_MAX_INSTANCES) {
                V_LOG_WARNING("more placed objects than the renderer supports; extra objects are not drawn");
                continue;
            }
            vrt_instance& I = s.instances[s.n_instances++];
            for (size_t slot = 0; slot < volumes.size(); slot++)
                if (volumes[slot] == vo->GetVoxelVolume()) I.volume_slot = (int)slot;
            I.position[0] = vo->Position.X; I.position[1] = vo->Position.Y; I.position[2] = vo->Position.Z;
            I.rotation[0] = vo->Rotation.x; I.rotation[1] = vo->Rotation.y; I.rotation[2] = vo->Rotation.z; I.rotation[3] = vo->Rotation.w;
            I.scale[0] = vo->Scale.X; I.scale[1] = vo->Scale.Y; I.scale[2] = vo->Scale.Z;
        }
    }
    return ok(vrt_scene_set(Ctx, &s), "vrt_scene_set");
}

void VHipRenderer::Render() {
    if (!IsActive()) {
        V_LOG_WARNING("Render() on an inactive renderer"); /* DXRenderer.cpp:62-65 */
        return;
    }
    const VObjectPtr<Scene::VScene> scene = SceneRef.lock();
    if (!scene) return;
    if (const VObjectPtr<Scene::VCamera> cam = scene->GetActiveCamera()) cam->AspectRatio = (float)Width / (float)Height; /* :47 */
    if (!SyncWithScene(*scene)) return;

    vrt_params p;
    memset(&p, 0, sizeof p);
    p.width = (int)Width;
    p.height = (int)Height;
    p.max_steps = MaxSteps;
    p.shadow = Shadows ? 1 : 0;
    p.mode = (int)RenderMode;
    p.path = DataPath;
    p.max_bounces = MaxBounces;
    p.eps_hit = 0.004f * MinCell;
    p.eps_in = 0.01f; /* Raytracing.hlsl:178 */
    p.step_min = 0.004f * MinCell;
    p.k_relax = 1.0f;
    const VObjectPtr<Scene::VCamera> cam = scene->GetActiveCamera();
    p.cone_eps = std::tan(cam->FOVAngle * (3.14159265358979323846f / 180.0f) * 0.5f) / (float)Height;
    Frame.resize((size_t)Width * Height * 4);
    ok(vrt_render(Ctx, &p, Frame.data()), "vrt_render");
}

bool VHipRenderer::GetLastTiming(vrt_timing& out) const { return Ctx && vrt_last_timing(Ctx, &out) == VRT_OK; }

}  // namespace Hip
}  // namespace Renderer
}  // namespace VolumeRaytracer

"""Deterministic workloads: the BASELINE.json configs as scenes (no RNG besides the seeded fuzz scenes, no files).
bench.py, __graft_entry__.smoke() and the tests build their scenes here; sizes are arguments."""
from __future__ import annotations

import math

import numpy as np

import volumetricraytracer_amd as v
from volumetricraytracer_amd import _abi

# Device volume format bench.py marches by default
BENCH_VOLUME_FORMAT = _abi.FORMAT_F32


def config2_sphere(resolution: int = 6, env: int = 64) -> v.VScene:
    """BASELINE config 2: one r=6 sphere volume (radius 40, extent 100), camera (300,0,0) looking -X,
    demo directional light (RendererEngineInstance.cpp:232-316)."""
    vol = v.sphere_volume(resolution, 100.0, 40.0, v.VMaterial((1.0, 0.0, 0.0, 1.0), 0.8, 0.0))
    return v.VScene(Camera=v.look_minus_x_camera(300.0), DirectionalLight=v.demo_light(),
                    Objects=[v.VVoxelObject(Volume=vol)], EnvironmentMap=v.procedural_skybox(env))


def config3_torus(resolution: int = 8, env: int = 256, distance: float = 260.0) -> v.VScene:
    """BASELINE config 3 (analytic variant): exact torus SDF on a 2^r grid, close camera so the
    object fills the frame, shadow ray meaningful (torus shadows itself)."""
    vol = v.torus_volume(resolution, 100.0, 55.0, 22.0, v.VMaterial((0.8, 0.6, 0.2, 1.0), 0.8, 0.0))
    cam = v.VCamera(Position=(distance * math.cos(math.radians(35.0)), 0.0, distance * math.sin(math.radians(35.0))),
                    Rotation=tuple(v.quat_mul(v.quat_from_axis_angle(v.UP, math.pi),
                                              v.quat_from_axis_angle(v.RIGHT, math.radians(35.0)))),
                    FOVAngle=60.0)
    return v.VScene(Camera=cam, DirectionalLight=v.demo_light(), Objects=[v.VVoxelObject(Volume=vol)],
                    EnvironmentMap=v.procedural_skybox(env))


def config5_instances(resolution: int = 7, env: int = 64, distinct_volumes: bool = False) -> v.VScene:
    """BASELINE config 5: 8 instances of an r=7 CSG volume on a 2x2x2 lattice, varied yaw/scale."""
    mats = [v.VMaterial((0.9, 0.3, 0.3, 1), 0.8, 0.0), v.VMaterial((0.3, 0.9, 0.3, 1), 0.6, 0.2)]
    base = v.csg_volume(resolution, 100.0, mats[0])
    vols = [base]
    if distinct_volumes:
        vols = [v.csg_volume(resolution, 100.0, mats[i % 2]) for i in range(8)]
    objs = []
    k = 0
    for ix in (-1, 1):
        for iy in (-1, 1):
            for iz in (-1, 1):
                s = (0.75, 1.0, 1.25)[k % 3]
                yaw = math.radians(20.0 * k)
                objs.append(v.VVoxelObject(Position=(ix * 150.0, iy * 150.0, iz * 150.0),
                                           Rotation=tuple(v.quat_from_axis_angle(v.UP, yaw)), Scale=(s, s, s),
                                           Volume=vols[k % len(vols)]))
                k += 1
    cam = v.look_minus_x_camera(900.0, 0.0)
    return v.VScene(Camera=cam, DirectionalLight=v.demo_light(), Objects=objs, EnvironmentMap=v.procedural_skybox(env))


def min_cell(scene: v.VScene) -> float:
    return min(vol.GetCellSize() for vol in scene.volumes())


def voxelized_torus(resolution: int = 8, material=None) -> v.VVoxelVolume:
    """A procedural UV-torus triangle mesh (128 x 64 quads, seedless) run through the build's C++
    Voxelizer at `_<resolution>` — the stand-in for the reference's missing Monkey.vox.
    The result is the Voxelizer's unsigned shell field (density_scale = thr, step_max = thr/2)."""
    from volumetricraytracer_amd import voxelizer as vx

    pos, nrm, idx = vx.torus_mesh(0.55, 0.22, 128, 64)
    p, be = vx.importer_space(pos)
    return vx.convert_mesh(p, idx, be, f"torus_{resolution}", material or v.VMaterial((0.8, 0.6, 0.2, 1.0), 0.8, 0.0))


def config3_voxelized(resolution: int = 8, env: int = 256, distance: float = 195.0, device_format: int = _abi.FORMAT_F32) -> v.VScene:
    """BASELINE config 3: voxelized mesh at 2^resolution cells, demo light, shadow ray meaningful."""
    vol = voxelized_torus(resolution).set_device_format(device_format)
    cam = v.VCamera(Position=(distance * math.cos(math.radians(35.0)), 0.0, distance * math.sin(math.radians(35.0))),
                    Rotation=tuple(v.quat_mul(v.quat_from_axis_angle(v.UP, math.pi),
                                              v.quat_from_axis_angle(v.RIGHT, math.radians(35.0)))),
                    FOVAngle=60.0)
    return v.VScene(Camera=cam, DirectionalLight=v.demo_light(), Objects=[v.VVoxelObject(Volume=vol)],
                    EnvironmentMap=v.procedural_skybox(env))


_bench_c3 = None


def bench_config3() -> v.VScene:
    """bench.py workload for BASELINE config 3 (1080p, 256^3 voxelized mesh, shadow ray on)."""
    global _bench_c3
    if _bench_c3 is None:
        _bench_c3 = config3_voxelized(8, 256)
    return _bench_c3


def orbit_cameras(scene: v.VScene, n: int, step_deg: float = 0.25):
    """n cameras on a short orbit about the world's up axis through the scene's own camera (frame n/2 is the scene's
    view): the batch of views one bench step renders — consecutive frames of a moving camera, not n copies of one frame.
    Returns (position, rotation, fov) triples as vrt_render_block takes them."""
    cam = scene.Camera
    out = []
    for f in range(n):
        a = math.radians((f - n // 2) * step_deg)
        c, s_ = math.cos(a), math.sin(a)
        pos = (cam.Position[0] * c - cam.Position[1] * s_, cam.Position[0] * s_ + cam.Position[1] * c, cam.Position[2])
        out.append((pos, tuple(v.quat_mul(v.quat_from_axis_angle(v.UP, a), cam.Rotation)), float(cam.FOVAngle)))
    return out


def demo_scene(mesh_resolution: int = 7, env: int = 64, mirror: bool = True) -> v.VScene:
    """The reference demo's scene (App/Private/RendererEngineInstance.cpp:232-263): the model of Resources/Model/Monkey.vox
    (absent from the checkout: the build's voxelized torus stands in), camera (300, 0, 100) yawed 180 degrees, the demo's
    directional light, and two 64^3 SDF spheres — radius 40, red, and radius 20, blue, roughness 0.1 / metallic 0.6 (they
    mirror: roughness < 0.3) — at their relative positions (200, 0, 100) and (100, 0, 200).  mirror=False gives them
    roughness 0.8 (no bounce rays: the directional-light kernel)."""
    rough = 0.1 if mirror else 0.8
    model = voxelized_torus(mesh_resolution)
    s1 = v.sphere_volume(6, 100.0, 40.0, v.VMaterial((1.0, 0.0, 0.0, 1.0), rough, 0.6))
    s2 = v.sphere_volume(6, 100.0, 20.0, v.VMaterial((0.0, 0.0, 1.0, 1.0), rough, 0.6))
    objs = [v.VVoxelObject(Volume=model), v.VVoxelObject(Position=(200.0, 0.0, 100.0), Volume=s1),
            v.VVoxelObject(Position=(100.0, 0.0, 200.0), Volume=s2)]
    return v.VScene(Camera=v.look_minus_x_camera(300.0, 100.0), DirectionalLight=v.demo_light(), Objects=objs,
                    EnvironmentMap=v.procedural_skybox(env))


def demo_frames(scene: v.VScene, n: int, dt: float = 1.0 / 60.0, t0: float = 0.0):
    """n consecutive frames of the demo's animation (RendererEngineInstance.cpp:111-130): sphere 1 orbits the up axis at
    +10 degrees per second, sphere 2 at -50, from their relative positions; everything else stands still.  Returns n scenes
    that share the volumes (and the sky box) of `scene` — what VRDXScene::SyncWithScene sees frame after frame."""
    import copy

    rel1, rel2 = (200.0, 0.0, 100.0), (100.0, 0.0, 200.0)
    out = []
    for f in range(n):
        t = t0 + (f + 1) * dt
        a1, a2 = math.radians((10.0 * t) % 360.0), math.radians((-50.0 * t) % 360.0)
        sc = copy.copy(scene)
        objs = [copy.copy(o) for o in scene.Objects]
        objs[1].Position = tuple(float(x) for x in v.quat_rotate(v.quat_from_axis_angle(v.UP, a1), rel1))
        objs[2].Position = tuple(float(x) for x in v.quat_rotate(v.quat_from_axis_angle(v.UP, a2), rel2))
        sc.Objects = objs
        out.append(sc)
    return out


def moving_instances(scene: v.VScene, n: int, step_deg: float = 0.25):
    """n consecutive frames of `scene` with EVERY placed object moving (each spins about the up axis at its own rate and bobs along
    it) under the camera of orbit_cameras: the per-frame scene state of bench.py's dynamic_scene leg — BASELINE config 5 as the
    reference would animate it (objects move every frame, TLAS rebuilt every frame: RendererEngineInstance.cpp:111-130,
    DXRenderer.cpp:809-825).  The scenes share `scene`'s volumes."""
    import copy

    cams = orbit_cameras(scene, n, step_deg)
    out = []
    for f in range(n):
        sc = copy.copy(scene)
        objs = []
        for k, o in enumerate(scene.Objects):
            q = copy.copy(o)
            q.Rotation = tuple(v.quat_mul(v.quat_from_axis_angle(v.UP, math.radians(0.5 * (k % 3 + 1) * (f - n // 2))), o.Rotation))
            q.Position = (o.Position[0], o.Position[1], o.Position[2] + 6.0 * math.sin(0.1 * (f - n // 2) + k))
            objs.append(q)
        sc.Objects = objs
        sc.Camera = v.VCamera(Position=cams[f][0], Rotation=cams[f][1], FOVAngle=cams[f][2])
        out.append(sc)
    return out


def full_closest_hit_scene(resolution: int = 6, env: int = 32) -> v.VScene:
    """Exercises the whole closest-hit shader: smooth metallic spheres that mirror each
    other (roughness 0.1 < 0.3 → bounce, as in the reference's demo materials,
    RendererEngineInstance.cpp:251-263), a rough one, one point and one spot light next to the
    directional light."""
    smooth_red = v.sphere_volume(resolution, 100.0, 60.0, v.VMaterial((1.0, 0.1, 0.1, 1.0), 0.1, 0.6))
    smooth_blue = v.sphere_volume(resolution, 100.0, 45.0, v.VMaterial((0.1, 0.1, 1.0, 1.0), 0.2, 0.3))
    rough = v.csg_volume(resolution, 100.0, v.VMaterial((0.8, 0.8, 0.3, 1.0), 0.7, 0.0))
    objs = [v.VVoxelObject(Position=(0.0, -110.0, 0.0), Volume=smooth_red),
            v.VVoxelObject(Position=(0.0, 80.0, 30.0), Volume=smooth_blue),
            v.VVoxelObject(Position=(-60.0, 0.0, -150.0), Rotation=tuple(v.quat_from_axis_angle(v.UP, 0.6)), Scale=(1.6, 1.6, 0.5), Volume=rough)]
    point = v.VPointLight(Position=(150.0, 0.0, 120.0), IlluminationStrength=400.0, Color=(1.0, 0.9, 0.7, 1.0),
                          AttenuationLinear=0.05, AttenuationExp=0.002)
    spot_dir = v.quat_mul(v.quat_from_axis_angle(v.UP, math.radians(150.0)), v.quat_from_axis_angle(v.RIGHT, math.radians(40.0)))
    spot = v.VSpotLight(Position=(200.0, -150.0, 200.0), Rotation=tuple(spot_dir), IlluminationStrength=900.0,
                        Color=(0.6, 1.0, 0.6, 1.0), AttenuationLinear=0.02, AttenuationExp=0.001, FalloffAngle=25.0, Angle=60.0)
    return v.VScene(Camera=v.look_minus_x_camera(420.0, 40.0), DirectionalLight=v.demo_light(), Objects=objs,
                    PointLights=[point], SpotLights=[spot], EnvironmentMap=v.procedural_skybox(env))


def boundary_box_scene(resolution: int = 4, env: int = 16, inset_cells: float = 0.6) -> v.VScene:
    """A box whose faces lie `inset_cells` cells inside its volume's own boundary (16^3 cells by default): every hit's cell is the first
    or the last of its axis, so the normal's central difference reaches for a neighbour cell BEYOND the grid — where the reference's
    texture Load returns 0 (Voxel.hlsli:607-617) and this build's default repeats the boundary cell (VRT_FLAG_REFERENCE_BOUNDARY_TEXELS
    selects the reference's rule).  Three faces in view, the object rotated and scaled anisotropically."""
    vol = v.VVoxelVolume(resolution, 100.0)
    gen = v.VDensityGenerator()
    half = 100.0 - inset_cells * vol.GetCellSize()
    gen.GetRootShape().AddChild(v.VBox((half, half, half)))
    vol.fill(gen.Evaluate)
    vol.Material = v.VMaterial((0.7, 0.75, 0.9, 1.0), 0.5, 0.1)
    q = v.quat_mul(v.quat_from_axis_angle(v.UP, math.radians(33.0)), v.quat_from_axis_angle(v.RIGHT, math.radians(-21.0)))
    obj = v.VVoxelObject(Rotation=tuple(q), Scale=(1.0, 0.8, 0.6), Volume=vol)
    return v.VScene(Camera=v.look_minus_x_camera(380.0, 60.0), DirectionalLight=v.demo_light(), Objects=[obj],
                    EnvironmentMap=v.procedural_skybox(env))


def reference_default_normal_texel() -> np.ndarray:
    """The 1x1 normal texture the reference binds to every material without a normal map: VColor(0.5, 0.5, 1) stored as 8 bits
    (VRDXScene::AllocateDefaultTextures, RDXScene.cpp:241-260) = (127, 127, 255) — a 0.3-degree tilt of every normal in its
    default render mode.  What the C++ adaptor binds by default (VHipRenderer::ReferenceDefaultTextures)."""
    return np.array([[[127, 127, 255, 255]]], np.uint8)


def with_reference_default_textures(scene: v.VScene) -> v.VScene:
    """`scene` with the reference's default normal texel on every material that has no normal map (in place; returns it)."""
    tex = reference_default_normal_texel()
    for vol in scene.volumes():
        if vol.Material.NormalTexture is None:
            vol.Material.NormalTexture = tex
    return scene


def procedural_textures(seed: int = 5):
    """Seeded stand-ins for the reference's material texture files (none ship with it): an albedo checker with
    per-texel jitter (13x8), a bumpy normal map (16x16, z-dominant) and a roughness/metal map (4x6)."""
    rng = np.random.default_rng(seed)
    alb = np.zeros((8, 13, 4), np.uint8)
    yy, xx = np.mgrid[0:8, 0:13]
    alb[..., :3] = np.where(((xx + yy) & 1)[..., None] == 0, 235, 90) + rng.integers(-20, 20, size=(8, 13, 3))
    alb[..., 3] = 255
    nrm = np.zeros((16, 16, 4), np.uint8)
    nrm[..., 0] = 128 + rng.integers(-70, 70, size=(16, 16))
    nrm[..., 1] = 128 + rng.integers(-70, 70, size=(16, 16))
    nrm[..., 2] = 230
    nrm[..., 3] = 255
    rm = np.zeros((6, 4, 4), np.uint8)
    rm[..., 0] = rng.integers(40, 256, size=(6, 4))
    rm[..., 1] = rng.integers(0, 256, size=(6, 4))
    rm[..., 3] = 255
    return alb, nrm, rm


def textured_scene(resolution: int = 6, env: int = 32) -> v.VScene:
    """full_closest_hit_scene with material textures: all three on the red mirror sphere, albedo only on the blue one,
    normal + RM on the scaled/rotated CSG object (exercises the object-space projection), shared images."""
    sc = full_closest_hit_scene(resolution, env)
    alb, nrm, rm = procedural_textures()
    vols = sc.volumes()
    m = vols[0].Material
    m.AlbedoTexture, m.NormalTexture, m.RMTexture, m.TextureScale = alb, nrm, rm, (37.0, 23.0)
    vols[1].Material.AlbedoTexture = alb
    vols[1].Material.TextureScale = (100.0, 100.0)
    m = vols[2].Material
    m.NormalTexture, m.RMTexture, m.TextureScale = nrm, rm, (61.0, 44.0)
    return sc


def random_scene(seed: int):
    """A seeded random scene + march parameters for the fuzz parity test: 1-6 instances of 1-3 volumes (sphere / torus /
    CSG / voxelized shell at resolutions 3-6) with arbitrary rotation, anisotropic (also mirrored) scale and material,
    a camera somewhere around (sometimes inside a volume), optional point / spot lights, textures, sky, any render
    mode, shadow on/off, 0-2 bounces, small or large step budgets, cone termination on/off."""
    rng = np.random.default_rng(1000 + seed)

    def material():
        m = v.VMaterial(tuple(rng.uniform(0.05, 1.0, 3)) + (1.0,), float(rng.uniform(0.02, 1.0)), float(rng.uniform(0.0, 1.0)))
        if rng.random() < 0.35:
            alb, nrm, rm = procedural_textures(int(rng.integers(0, 1000)))
            if rng.random() < 0.7:
                m.AlbedoTexture = alb
            if rng.random() < 0.5:
                m.NormalTexture = nrm
            if rng.random() < 0.5:
                m.RMTexture = rm
            m.TextureScale = (float(rng.uniform(5, 120)), float(rng.uniform(5, 120)))
        return m

    vols = []
    for _ in range(int(rng.integers(1, 4))):
        kind, res = int(rng.integers(0, 4)), int(rng.integers(3, 7))
        if kind == 0:
            vol = v.sphere_volume(res, 100.0, float(rng.uniform(25, 80)), material())
        elif kind == 1:
            vol = v.torus_volume(res, 100.0, float(rng.uniform(35, 60)), float(rng.uniform(12, 30)), material())
        elif kind == 2:
            vol = v.csg_volume(res, 100.0, material())
        else:
            vol = voxelized_torus(min(res, 5), material())
        vols.append(vol)
    objs = []
    for _ in range(int(rng.integers(1, 7))):
        axis = rng.normal(size=3)
        q = v.quat_from_axis_angle(tuple(axis / np.linalg.norm(axis)), float(rng.uniform(0, 2 * math.pi)))
        sc = rng.uniform(0.4, 1.8, 3) * (rng.choice([1.0, 1.0, 1.0, -1.0], 3))
        objs.append(v.VVoxelObject(Position=tuple(rng.uniform(-160, 160, 3)), Rotation=tuple(q), Scale=tuple(float(x) for x in sc),
                                   Volume=vols[int(rng.integers(0, len(vols)))]))
    used = {id(o.Volume) for o in objs}
    cam_pos = rng.uniform(-1, 1, 3)
    cam_pos = cam_pos / np.linalg.norm(cam_pos) * rng.uniform(60, 650)
    yaw = math.atan2(-cam_pos[1], -cam_pos[0])
    pitch = math.atan2(cam_pos[2], math.hypot(cam_pos[0], cam_pos[1]))  # looks at the origin (pitch about +Y tilts +X towards -Z... up)
    cam_q = v.quat_mul(v.quat_from_axis_angle(v.UP, yaw), v.quat_from_axis_angle(v.RIGHT, pitch))
    cam = v.VCamera(Position=tuple(float(x) for x in cam_pos), Rotation=tuple(cam_q), FOVAngle=float(rng.uniform(30, 95)))
    light_axis = rng.normal(size=3)
    light = v.VLight(Rotation=tuple(v.quat_from_axis_angle(tuple(light_axis / np.linalg.norm(light_axis)), float(rng.uniform(0, 6.28)))),
                     IlluminationStrength=float(rng.uniform(0, 8)))
    points = [v.VPointLight(Position=tuple(rng.uniform(-300, 300, 3)), IlluminationStrength=float(rng.uniform(50, 900)),
                            Color=tuple(rng.uniform(0.2, 1, 3)) + (1.0,), AttenuationLinear=float(rng.uniform(0.005, 0.1)),
                            AttenuationExp=float(rng.uniform(0.0002, 0.01))) for _ in range(int(rng.integers(0, 3)))]
    spots = []
    for _ in range(int(rng.integers(0, 3))):
        a = rng.normal(size=3)
        spots.append(v.VSpotLight(Position=tuple(rng.uniform(-300, 300, 3)),
                                  Rotation=tuple(v.quat_from_axis_angle(tuple(a / np.linalg.norm(a)), float(rng.uniform(0, 6.28)))),
                                  IlluminationStrength=float(rng.uniform(100, 1500)), Color=tuple(rng.uniform(0.2, 1, 3)) + (1.0,),
                                  AttenuationLinear=float(rng.uniform(0.005, 0.05)), AttenuationExp=float(rng.uniform(0.0002, 0.005)),
                                  FalloffAngle=float(rng.uniform(10, 40)), Angle=float(rng.uniform(45, 120))))
    env = v.procedural_skybox(int(rng.choice([4, 16, 33]))) if rng.random() < 0.8 else None
    scene = v.VScene(Camera=cam, DirectionalLight=light, Objects=objs, PointLights=points, SpotLights=spots, EnvironmentMap=env)
    w, h = int(rng.integers(40, 130)), int(rng.integers(30, 80))
    p = v.default_params(w, h, min(vol.GetCellSize() for vol in vols if id(vol) in used), int(rng.choice([3, 40, 255])),
                         shadow=bool(rng.random() < 0.7), mode=int(rng.integers(0, 8)), cone=bool(rng.random() < 0.7))
    p.max_bounces = int(rng.integers(0, 3))
    return scene, p

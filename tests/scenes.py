"""Deterministic scenes shared by the tests, smoke() and bench.py (no RNG, no files).
They are the BASELINE.json configs at test-friendly sizes; sizes are arguments."""
from __future__ import annotations

import math

import numpy as np

import volumetricraytracer_amd as v


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


def config3_voxelized(resolution: int = 8, env: int = 256, distance: float = 195.0) -> v.VScene:
    """BASELINE config 3: voxelized mesh at 2^resolution cells, demo light, shadow ray meaningful."""
    vol = voxelized_torus(resolution)
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

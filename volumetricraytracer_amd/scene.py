"""Host-side scene model: a dependency-free mirror of the reference's Voxel/Scene/Core types
that feed the renderer (numpy instead of Eigen).  Names and semantics follow the reference so
that tests read like code written against it.

Reference (paths relative to /root/reference/VolumetricRaytracer/VolumetricRaytracer/):
  VVoxelVolume        Voxel/Public/VoxelVolume.h:37-100, Voxel/Private/VoxelVolume.cpp:19-27,139-146
  VMaterial           Core/Public/Material.h:22-42
  VQuat helpers       Core/Private/Quat.cpp:28-63, axes Core/Private/Vector.cpp:42-46
  VCamera             Scene/Public/Camera.h:23-33
  VLight & co         Scene/Public/Light.h, PointLight.h:26-27, SpotLight.h:26-29
  VDensityGenerator   Scene/Private/DensityGenerator.cpp:18-108
  demo scene          App/Private/RendererEngineInstance.cpp:232-316
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _abi

UP = np.array([0.0, 0.0, 1.0], dtype=np.float32)       # VVector::UP
RIGHT = np.array([0.0, 1.0, 0.0], dtype=np.float32)    # VVector::RIGHT
FORWARD = np.array([1.0, 0.0, 0.0], dtype=np.float32)  # VVector::FORWARD
IDENTITY = np.array([0.0, 0.0, 0.0, 1.0], dtype=np.float32)  # quaternion x,y,z,w


# ---- quaternions (x, y, z, w), Eigen conventions -------------------------------------------

def quat_from_axis_angle(axis: Sequence[float], angle_rad: float) -> np.ndarray:
    """VQuat::FromAxisAngle (Quat.cpp:28-33): Eigen::AngleAxisf → Quaternionf."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    s = math.sin(angle_rad * 0.5)
    return np.array([a[0] * s, a[1] * s, a[2] * s, math.cos(angle_rad * 0.5)], dtype=np.float32)


def quat_mul(a: Sequence[float], b: Sequence[float]) -> np.ndarray:
    """Hamilton product a*b (VQuat::operator*, Quat.cpp:118-121): apply b first, then a."""
    ax, ay, az, aw = [float(v) for v in a]
    bx, by, bz, bw = [float(v) for v in b]
    return np.array(
        [
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
            aw * bw - ax * bx - ay * by - az * bz,
        ],
        dtype=np.float32,
    )


def quat_rotate(q: Sequence[float], v: Sequence[float]) -> np.ndarray:
    """VQuat::operator*(VVector) (Quat.cpp:91-95)."""
    x, y, z, w = [float(c) for c in q]
    qv = np.array([x, y, z])
    v = np.asarray(v, dtype=np.float64)
    uv = 2.0 * np.cross(qv, v)
    return (v + w * uv + np.cross(qv, uv)).astype(np.float32)


def quat_inverse(q: Sequence[float]) -> np.ndarray:
    x, y, z, w = [float(c) for c in q]
    n = x * x + y * y + z * z + w * w
    return np.array([-x / n, -y / n, -z / n, w / n], dtype=np.float32)


def quat_from_euler_deg(roll: float, yaw: float, pitch: float) -> np.ndarray:
    """VQuat::FromEulerAnglesDegrees (Quat.cpp:57-66)."""
    r = math.radians
    return quat_mul(
        quat_mul(quat_from_axis_angle(RIGHT, r(pitch)), quat_from_axis_angle(UP, r(yaw))),
        quat_from_axis_angle(FORWARD, r(roll)),
    )


# ---- material / volume ---------------------------------------------------------------------

@dataclass
class VMaterial:
    """VMaterial, Core/Public/Material.h:21-45.  The texture *paths* travel in .vox files; the decoded images
    (uint8 [H, W, 4], R8G8B8A8) are attached by the caller — the reference decodes them with WIC/DDS loaders
    (Renderer/Private/TextureFactory.cpp:58-125), which this build leaves to the application."""
    AlbedoColor: Sequence[float] = (0.8, 0.8, 0.8, 1.0)
    Roughness: float = 0.8
    Metallic: float = 0.0
    AlbedoTexturePath: str = ""
    NormalTexturePath: str = ""
    RMTexturePath: str = ""
    TextureScale: Sequence[float] = (100.0, 100.0)
    AlbedoTexture: Optional[np.ndarray] = None
    NormalTexture: Optional[np.ndarray] = None
    RMTexture: Optional[np.ndarray] = None

    def textures(self):
        """(albedo, normal, rm) images or None, validated."""
        out = []
        for t in (self.AlbedoTexture, self.NormalTexture, self.RMTexture):
            if t is not None:
                t = np.ascontiguousarray(t, dtype=np.uint8)
                if t.ndim != 3 or t.shape[2] != 4 or t.shape[0] < 1 or t.shape[1] < 1:
                    raise ValueError("material textures must be uint8 [H, W, 4]")
            out.append(t)
        return out

    def to_abi(self) -> _abi.vrt_material:
        m = _abi.vrt_material()
        for i in range(4):
            m.tint[i] = float(self.AlbedoColor[i])
        m.roughness = float(self.Roughness)
        m.metallic = float(self.Metallic)
        return m


class VVoxelVolume:
    """Dense voxel grid.  N = 2^resolution + 1 voxels per axis, cube [-extent, extent]^3.

    `density` is a C-contiguous float32 array of shape (N, N, N) whose axes are (x, z, y), i.e.
    flat index x*N*N + z*N + y — VMathHelpers::Index3DTo1D (Core/Private/MathHelpers (2).cpp:43-46).
    """

    def __init__(self, resolution: int, extent: float):
        if not (0 <= resolution <= 9):
            raise ValueError("resolution must be within 0..9 (VRT_MAX_RESOLUTION)")
        self.Resolution = int(resolution)
        self.VolumeExtends = float(extent)
        self.N = (1 << self.Resolution) + 1                      # VoxelVolume.cpp:23
        self.CellSize = np.float32(self.VolumeExtends * 2) / np.float32(self.N - 1)  # :24
        self.density = np.full((self.N,) * 3, 30.0, dtype=np.float32)  # VVoxel::Density default, Voxel.h:29
        self.material_id = np.zeros((self.N,) * 3, dtype=np.uint8)
        self.Material = VMaterial()
        # metric of the density field: object-space length of one density unit, and the largest
        # safe step (<= 0: unbounded).  1 / unbounded for analytic SDFs.
        self.density_scale = 1.0
        self.step_max = 0.0
        # how the renderer keeps the samples on the device: _abi.FORMAT_F32, or _abi.FORMAT_TEXEL16 = the reference's
        # own 16-bit volume texel (RDXVoxelVolume.cpp:399-421)
        self.device_format = _abi.FORMAT_F32
        self.dirty = True

    def GetSize(self) -> int:
        return self.N

    def GetVolumeExtends(self) -> float:
        return self.VolumeExtends

    def GetCellSize(self) -> float:
        return float(self.CellSize)

    def axis_positions(self) -> np.ndarray:
        """VoxelIndexToRelativePosition along one axis (VoxelVolume.cpp:139-146), fp32."""
        idx = np.arange(self.N, dtype=np.float32)
        return idx * np.float32(self.CellSize) + np.float32(-self.VolumeExtends)

    def set_device_format(self, fmt: int) -> "VVoxelVolume":
        if fmt not in (_abi.FORMAT_F32, _abi.FORMAT_TEXEL16):
            raise ValueError("unknown device format")
        if fmt != self.device_format:
            self.device_format = int(fmt)
            self.dirty = True
        return self

    def reference_texels(self) -> np.ndarray:
        """The reference's volume texture for this volume: uint8 [N, N, N, 4] indexed [z, y, x] with
        R = sign<<7 | q>>8, G = q & 0xff, B = A = material, q = trunc(|d| * 100) & 0x7fff
        (VDXVoxelVolume::UpdateVolumeTexture / EncodeVoxel, Renderer/DX/Private/RDXVoxelVolume.cpp:294-327, 399-421)."""
        d = np.asarray(self.density, dtype=np.float32)               # [x, z, y]
        q = ((np.abs(d) * np.float32(100.0)).astype(np.int64) & 0x7FFF).astype(np.uint16)
        tex = np.zeros((self.N, self.N, self.N, 4), dtype=np.uint8)  # [z, y, x, rgba]
        r = ((q >> 8).astype(np.uint8) | np.where(d < 0, 0x80, 0).astype(np.uint8))
        for ch, a in ((0, r), (1, (q & 0xFF).astype(np.uint8)), (2, self.material_id), (3, self.material_id)):
            tex[..., ch] = np.transpose(a, (1, 2, 0))                # [x, z, y] -> [z, y, x]
        return tex

    def quantize_like_reference_texels(self) -> "VVoxelVolume":
        """Rounds the densities the way the reference's GPU texture does: sign bit + 15-bit trunc(|d| * 100)
        (VDXVoxelVolume::EncodeVoxel, Renderer/DX/Private/RDXVoxelVolume.cpp:399-421; DecodeDensity,
        Shaders/Include/Voxel.hlsli:254-266).  This build keeps fp32 on the device; apply this before the
        upload to march exactly the field the DXR backend sees (0.01 quantum, magnitudes wrap at 327.68)."""
        d = np.asarray(self.density, dtype=np.float32)
        q = ((np.abs(d) * np.float32(100.0)).astype(np.int64) & 0xFFFF & 0x7FFF).astype(np.float32) * np.float32(0.01)
        self.density = np.where(d < 0, -q, q).astype(np.float32)
        self.dirty = True
        return self

    def fill(self, fn: Callable[[np.ndarray, np.ndarray, np.ndarray], np.ndarray]) -> "VVoxelVolume":
        """density[x,z,y] = fn(X, Y, Z); material = (density <= 0) like InitSphere
        (RendererEngineInstance.cpp:286-308)."""
        p = self.axis_positions()
        X = p[:, None, None]
        Z = p[None, :, None]
        Y = p[None, None, :]
        d = np.asarray(fn(X, Y, Z), dtype=np.float32)
        self.density = np.ascontiguousarray(np.broadcast_to(d, (self.N,) * 3), dtype=np.float32)
        self.material_id = (self.density <= 0).astype(np.uint8)
        self.dirty = True
        return self

    def voxel_records(self) -> np.ndarray:
        """The std::vector<VVoxel> image: N^3 records {u8 Material, pad[3], f32 Density} (Voxel.h:23-30)."""
        rec = np.zeros(self.N ** 3, dtype=np.dtype([("material", "u1"), ("pad", "u1", 3), ("density", "<f4")]))
        rec["material"] = self.material_id.reshape(-1)
        rec["density"] = self.density.reshape(-1)
        return rec


# ---- analytic density shapes (VDensityGenerator) ---------------------------------------------

class VDensityShape:
    def __init__(self, position=(0, 0, 0), rotation=IDENTITY):
        self.Position = np.asarray(position, dtype=np.float32)
        self.Rotation = np.asarray(rotation, dtype=np.float32)

    def _local(self, X, Y, Z):
        X = X - self.Position[0]
        Y = Y - self.Position[1]
        Z = Z - self.Position[2]
        qi = quat_inverse(self.Rotation)
        if np.allclose(qi, IDENTITY):
            return X, Y, Z
        # rotate by the inverse quaternion: columns of R(qi)
        ex, ey, ez = (quat_rotate(qi, e) for e in (FORWARD, RIGHT, UP))
        return (ex[0] * X + ey[0] * Y + ez[0] * Z, ex[1] * X + ey[1] * Y + ez[1] * Z, ex[2] * X + ey[2] * Y + ez[2] * Z)

    def Evaluate(self, X, Y, Z):
        return self.EvaluateInternal(*self._local(X, Y, Z))

    def EvaluateInternal(self, X, Y, Z):  # pragma: no cover - abstract
        raise NotImplementedError


class VSphere(VDensityShape):
    def __init__(self, radius: float, **kw):
        super().__init__(**kw)
        self.Radius = np.float32(radius)

    def EvaluateInternal(self, X, Y, Z):  # DensityGenerator.cpp:33-36
        return np.sqrt(X * X + Y * Y + Z * Z) - self.Radius


class VBox(VDensityShape):
    def __init__(self, extends: Sequence[float], **kw):
        super().__init__(**kw)
        self.Extends = np.asarray(extends, dtype=np.float32)

    def EvaluateInternal(self, X, Y, Z):  # :27-31
        qx, qy, qz = np.abs(X) - self.Extends[0], np.abs(Y) - self.Extends[1], np.abs(Z) - self.Extends[2]
        out = np.sqrt(np.maximum(qx, 0) ** 2 + np.maximum(qy, 0) ** 2 + np.maximum(qz, 0) ** 2)
        return out + np.minimum(np.maximum(qx, np.maximum(qy, qz)), 0)


class VCylinder(VDensityShape):
    def __init__(self, radius: float, height: float, **kw):
        super().__init__(**kw)
        self.Radius = np.float32(radius)
        self.Height = np.float32(height)

    def EvaluateInternal(self, X, Y, Z):  # :38-42
        dx = np.abs(np.sqrt(X * X + Z * Z)) - self.Radius
        dy = np.abs(Y) - self.Height
        return np.minimum(np.maximum(dx, dy), 0) + np.sqrt(np.maximum(dx, 0) ** 2 + np.maximum(dy, 0) ** 2)


ADD, SUBTRACT = 0, 1


class VDensityShapeContainer:
    """CSG node, DensityGenerator.cpp:45-96."""

    def __init__(self, shape: Optional[VDensityShape] = None, combination: int = ADD):
        self.Shape = shape
        self.CombinationType = combination
        self.Children: List["VDensityShapeContainer"] = []

    def AddChild(self, shape: VDensityShape, combination: int = ADD) -> "VDensityShapeContainer":
        c = VDensityShapeContainer(shape, combination)
        self.Children.append(c)
        return c

    def Evaluate(self, X, Y, Z):
        sx, sy, sz = X, Y, Z
        if self.Shape is not None:
            d = self.Shape.Evaluate(X, Y, Z)
            sx, sy, sz = self.Shape._local(X, Y, Z)
        elif self.Children:
            d = self.Children[0].Evaluate(X, Y, Z)
        else:
            d = np.zeros(np.broadcast(X, Y, Z).shape, dtype=np.float32)
        for child in self.Children:
            cd = child.Evaluate(sx, sy, sz)
            if child.CombinationType == ADD:
                d = np.minimum(d, cd)
            elif child.CombinationType == SUBTRACT:
                d = np.maximum(d, -cd)
        return d


class VDensityGenerator:
    def __init__(self):
        self.Root = VDensityShapeContainer()

    def GetRootShape(self) -> VDensityShapeContainer:
        return self.Root

    def Evaluate(self, X, Y, Z):  # :97-103 (generator itself at the origin, identity)
        return self.Root.Evaluate(X, Y, Z)


def sphere_volume(resolution: int = 6, extent: float = 100.0, radius: float = 40.0,
                  material: Optional[VMaterial] = None) -> VVoxelVolume:
    """InitSphere (RendererEngineInstance.cpp:266-316): density = |p| - radius."""
    vol = VVoxelVolume(resolution, extent)
    gen = VDensityGenerator()
    gen.GetRootShape().AddChild(VSphere(radius))
    vol.fill(gen.Evaluate)
    if material is not None:
        vol.Material = material
    return vol


def torus_volume(resolution: int = 8, extent: float = 100.0, major: float = 55.0, minor: float = 22.0,
                 material: Optional[VMaterial] = None) -> VVoxelVolume:
    """Exact torus SDF around the Z axis sampled on the grid (analytic variant of bench config 3)."""
    vol = VVoxelVolume(resolution, extent)
    R, r = np.float32(major), np.float32(minor)

    def f(X, Y, Z):
        q = np.sqrt(X * X + Y * Y) - R
        return np.sqrt(q * q + Z * Z) - r

    vol.fill(f)
    if material is not None:
        vol.Material = material
    return vol


def csg_volume(resolution: int = 7, extent: float = 100.0, material: Optional[VMaterial] = None) -> VVoxelVolume:
    """Sphere minus box, via the VDensityGenerator rules (bench config 5)."""
    vol = VVoxelVolume(resolution, extent)
    gen = VDensityGenerator()
    node = gen.GetRootShape().AddChild(VSphere(70.0))
    node.AddChild(VBox((80.0, 30.0, 30.0)), SUBTRACT)
    vol.fill(gen.Evaluate)
    if material is not None:
        vol.Material = material
    return vol


# ---- placed objects / scene ----------------------------------------------------------------

@dataclass
class VLevelObject:
    Position: Sequence[float] = (0.0, 0.0, 0.0)
    Rotation: Sequence[float] = tuple(IDENTITY)
    Scale: Sequence[float] = (1.0, 1.0, 1.0)


@dataclass
class VVoxelObject(VLevelObject):
    Volume: Optional[VVoxelVolume] = None


@dataclass
class VCamera(VLevelObject):
    FOVAngle: float = 60.0
    NearClipPlane: float = 0.01
    FarClipPlane: float = 125.0


@dataclass
class VLight(VLevelObject):
    IlluminationStrength: float = 1.0
    Color: Sequence[float] = (1.0, 1.0, 1.0, 1.0)


@dataclass
class VPointLight(VLight):
    AttenuationLinear: float = 0.5    # PointLight.h:26-27
    AttenuationExp: float = 0.005


@dataclass
class VSpotLight(VLight):
    AttenuationLinear: float = 0.5    # SpotLight.h:26-29
    AttenuationExp: float = 0.005
    FalloffAngle: float = 20.0
    Angle: float = 45.0


@dataclass
class VScene:
    Camera: VCamera = field(default_factory=VCamera)
    DirectionalLight: VLight = field(default_factory=VLight)
    Objects: List[VVoxelObject] = field(default_factory=list)
    PointLights: List[VPointLight] = field(default_factory=list)
    SpotLights: List[VSpotLight] = field(default_factory=list)
    EnvironmentMap: Optional[np.ndarray] = None  # uint8 [6, S, S, 4]

    def volumes(self) -> List[VVoxelVolume]:
        """Distinct volumes in first-use order; the index is the volume slot."""
        seen: List[VVoxelVolume] = []
        for o in self.Objects:
            if o.Volume is not None and not any(o.Volume is v for v in seen):
                seen.append(o.Volume)
        if len(seen) > _abi.VRT_MAX_VOLUMES:
            raise ValueError(f"at most {_abi.VRT_MAX_VOLUMES} distinct volumes")
        return seen

    def to_abi(self) -> _abi.vrt_scene:
        s = _abi.vrt_scene()
        cam = self.Camera
        for i in range(3):
            s.cam_position[i] = float(cam.Position[i])
        for i in range(4):
            s.cam_rotation[i] = float(cam.Rotation[i])
        s.cam_fov_deg = float(cam.FOVAngle)
        s.cam_near = float(cam.NearClipPlane)
        s.cam_far = float(cam.FarClipPlane)
        ld = quat_rotate(self.DirectionalLight.Rotation, FORWARD)  # RDXScene.cpp:720
        for i in range(3):
            s.light_dir[i] = float(ld[i])
        s.light_strength = float(self.DirectionalLight.IlluminationStrength)
        vols = self.volumes()
        if len(self.Objects) > _abi.VRT_MAX_INSTANCES:
            raise ValueError(f"at most {_abi.VRT_MAX_INSTANCES} placed objects")
        s.n_instances = len(self.Objects)
        for k, o in enumerate(self.Objects):
            inst = s.instances[k]
            inst.volume_slot = next(i for i, v in enumerate(vols) if v is o.Volume)
            for i in range(3):
                inst.position[i] = float(o.Position[i])
                inst.scale[i] = float(o.Scale[i])
            for i in range(4):
                inst.rotation[i] = float(o.Rotation[i])
        s.n_point_lights = min(len(self.PointLights), _abi.VRT_MAX_POINT_LIGHTS)
        for k in range(s.n_point_lights):
            L, o = self.PointLights[k], s.point_lights[k]
            for i in range(3):
                o.position[i] = float(L.Position[i])
                o.color[i] = float(L.Color[i])
            o.intensity, o.att_linear, o.att_exp = float(L.IlluminationStrength), float(L.AttenuationLinear), float(L.AttenuationExp)
        s.n_spot_lights = min(len(self.SpotLights), _abi.VRT_MAX_SPOT_LIGHTS)
        for k in range(s.n_spot_lights):
            L, o = self.SpotLights[k], s.spot_lights[k]
            fwd = quat_rotate(L.Rotation, FORWARD)
            for i in range(3):
                o.position[i] = float(L.Position[i])
                o.color[i] = float(L.Color[i])
                o.forward[i] = float(fwd[i])
            o.intensity, o.att_linear, o.att_exp = float(L.IlluminationStrength), float(L.AttenuationLinear), float(L.AttenuationExp)
            # DXLightFactory.cpp:46-47, VMathHelpers::ToRadians uses 3.141592f
            o.cos_angle = math.cos(L.Angle * 0.5 * (3.141592 / 180.0))
            o.cos_falloff_angle = math.cos(L.FalloffAngle * 0.5 * (3.141592 / 180.0))
        return s


def demo_light() -> VLight:
    """Directional light of the demo scene: yaw 45°, pitch -30°, strength 6
    (RendererEngineInstance.cpp:239-241)."""
    q = quat_mul(quat_from_axis_angle(UP, math.radians(45.0)), quat_from_axis_angle(RIGHT, math.radians(-30.0)))
    return VLight(Rotation=tuple(q), IlluminationStrength=6.0)


def look_minus_x_camera(distance: float, height: float = 0.0, fov: float = 60.0) -> VCamera:
    """Camera on +X looking at the origin side: yaw 180° about +Z (RendererEngineInstance.cpp:237)."""
    return VCamera(Position=(distance, 0.0, height), Rotation=tuple(quat_from_axis_angle(UP, math.pi)), FOVAngle=fov)


def procedural_skybox(face_size: int = 256) -> np.ndarray:
    """Deterministic 6 x S x S RGBA8 cube map: vertical gradient + per-face tint (SURVEY §8d;
    the reference's Skybox.dds is not in the checkout)."""
    S = face_size
    v = (np.arange(S, dtype=np.float32) + 0.5) / S
    u = v
    tints = np.array(
        [[1.0, 0.85, 0.8], [0.8, 0.85, 1.0], [0.85, 1.0, 0.8], [1.0, 0.8, 1.0], [0.6, 0.75, 1.0], [0.55, 0.5, 0.45]],
        dtype=np.float32,
    )
    out = np.zeros((6, S, S, 4), dtype=np.uint8)
    for f in range(6):
        g = 0.35 + 0.6 * (1.0 - v)[:, None] * np.ones((1, S), dtype=np.float32)
        checker = (((np.floor(u * 8)[None, :] + np.floor(v * 8)[:, None]) % 2) * 0.08).astype(np.float32)
        rgb = np.clip((g + checker)[:, :, None] * tints[f][None, None, :], 0.0, 1.0)
        out[f, :, :, :3] = np.floor(rgb * 255.0 + 0.5).astype(np.uint8)
        out[f, :, :, 3] = 255
    return out


def march_budget(resolution: int, base: int = 255) -> int:
    """Positions a ray may visit per instance: the reference's 255 (Raytracing.hlsl:229) up to its largest resolution, 8;
    doubled per resolution step beyond it (cells half the size need twice the positions for the same path)."""
    return int(base) << max(0, int(resolution) - 8)


def default_params(width: int, height: int, cell: float, max_steps: int = 128, shadow: bool = False,
                   mode: int = _abi.MODE_INTERP_NOTEX, path: int = _abi.PATH_AUTO, fov_deg: float = 60.0,
                   cone: bool = True, k_relax: float = 1.7) -> _abi.vrt_params:
    """March contract defaults: hit threshold and minimum step are 0.4 % of a cell at
    the ray origin — at most 0.02, a fifth of the 0.1 the reference backs its secondary rays off the hit
    (Raytracing.hlsl:52): in a coarse volume (resolution <= 4: cells of 12 units and more) 0.4 % of a cell is that
    very offset, and every shadow ray would "hit" the surface it starts on — and grow with the pixel footprint
    (angular pixel radius tan(fov/2)/height)."""
    p = _abi.vrt_params()
    p.width, p.height = int(width), int(height)
    p.max_steps = int(max_steps)
    p.shadow = 1 if shadow else 0
    p.mode = int(mode)
    p.path = int(path)
    p.max_bounces = 0
    p.flags = 0
    p.eps_hit = float(np.float32(min(0.004 * cell, 0.02)))
    p.eps_in = 0.01  # Raytracing.hlsl:178
    p.step_min = float(np.float32(min(0.004 * cell, 0.02)))
    p.k_relax = float(k_relax)  # > 1: over-relaxed sphere-trace with the sphere-overlap fallback
    p.cone_eps = float(np.float32(math.tan(math.radians(fov_deg) * 0.5) / height)) if cone else 0.0
    return p

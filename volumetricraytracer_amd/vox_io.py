"""Reader / writer of the reference's `.vox` scene files (pure Python, independent of the C++
implementation in csrc/host/HostSerialization.cpp — the two are cross-checked in tests).

archive := u64 BufferSize, bytes, u64 numProps, numProps x {u64 nameLen (incl. NUL), name, archive}
(Core/Private/SerializationManager.cpp:24-100; property order is hash-map order, so the reader is
order-agnostic; this writer sorts names).  Scene / volume / object / light property names:
Scene/Private/Scene.cpp:392-458, Voxel/Private/VoxelVolume.cpp:178-198, Core/Private/Material.cpp:19-70,
Scene/Private/VoxelObject.cpp:37-71, Light.cpp:17-57, PointLight.cpp:17-33, SpotLight.cpp:17-37.
"""
from __future__ import annotations

import os
import struct
from typing import BinaryIO, Dict, Tuple

import numpy as np

from .scene import VLight, VMaterial, VPointLight, VScene, VSpotLight, VVoxelObject, VVoxelVolume

VOXEL_DTYPE = np.dtype([("material", "u1"), ("pad", "u1", 3), ("density", "<f4")])


class Archive:
    def __init__(self, buffer: bytes = b""):
        self.buffer = bytes(buffer)
        self.props: Dict[str, "Archive"] = {}

    def __getitem__(self, k: str) -> "Archive":
        return self.props[k]

    def __contains__(self, k: str) -> bool:
        return k in self.props

    def unpack(self, fmt: str):
        n = struct.calcsize(fmt)
        if len(self.buffer) < n:
            raise ValueError("archive buffer too small")
        v = struct.unpack(fmt, self.buffer[:n])
        return v[0] if len(v) == 1 else v

    def cstr(self) -> str:
        return self.buffer.split(b"\0", 1)[0].decode("utf-8", "replace")


def _read(f: BinaryIO, depth: int = 0) -> Archive:
    if depth > 64:
        raise ValueError("archive nesting too deep")
    head = f.read(8)
    if len(head) != 8:
        raise ValueError("truncated archive")
    (size,) = struct.unpack("<Q", head)
    a = Archive(f.read(size))
    if len(a.buffer) != size:
        raise ValueError("truncated archive buffer")
    (n,) = struct.unpack("<Q", f.read(8))
    for _ in range(n):
        (ln,) = struct.unpack("<Q", f.read(8))
        if not 0 < ln <= 4096:
            raise ValueError("bad property name length")
        name = f.read(ln).split(b"\0", 1)[0].decode()
        a.props[name] = _read(f, depth + 1)
    return a


def _write(a: Archive, f: BinaryIO) -> None:
    f.write(struct.pack("<Q", len(a.buffer)))
    f.write(a.buffer)
    f.write(struct.pack("<Q", len(a.props)))
    for name in sorted(a.props):
        raw = name.encode() + b"\0"
        f.write(struct.pack("<Q", len(raw)))
        f.write(raw)
        _write(a.props[name], f)


def read_archive(path: str) -> Archive:
    with open(path, "rb") as f:
        return _read(f)


def write_archive(a: Archive, path: str) -> None:
    with open(path, "wb") as f:
        _write(a, f)


def _cstring(s: str) -> Archive:
    return Archive(s.encode() + b"\0")


def _material_archive(m: VMaterial) -> Archive:
    a = Archive()
    a.props["Color"] = Archive(struct.pack("<4f", *[float(c) for c in m.AlbedoColor]))
    a.props["Roughness"] = Archive(struct.pack("<f", m.Roughness))
    a.props["Metallic"] = Archive(struct.pack("<f", m.Metallic))
    a.props["TextureScale"] = Archive(struct.pack("<2f", float(m.TextureScale[0]), float(m.TextureScale[1])))
    a.props["AlbedoTexture"] = _cstring(m.AlbedoTexturePath)
    a.props["NormalTexture"] = _cstring(m.NormalTexturePath)
    a.props["RMTexture"] = _cstring(m.RMTexturePath)
    return a


def _read_cstring(a: Archive) -> str:
    return bytes(a.buffer).split(b"\0", 1)[0].decode(errors="replace")


def volume_archive(v: VVoxelVolume) -> Archive:
    a = Archive(v.voxel_records().tobytes())
    a.props["Resolution"] = Archive(struct.pack("<B", v.Resolution))
    a.props["Extends"] = Archive(struct.pack("<f", v.VolumeExtends))
    a.props["Material"] = _material_archive(v.Material)
    return a


def volume_from_archive(a: Archive, source_path: str = "") -> VVoxelVolume:
    """source_path: the .vox file the archive came from — material texture paths that are not absolute are resolved against
    its folder (VMaterial::Deserialize, Core/Private/Material.cpp:72-100); empty: paths stay as stored."""
    res = a["Resolution"].unpack("<B")
    ext = a["Extends"].unpack("<f")
    v = VVoxelVolume(res, ext)
    rec = np.frombuffer(a.buffer, dtype=VOXEL_DTYPE, count=v.N ** 3)
    v.density = np.ascontiguousarray(rec["density"].reshape(v.N, v.N, v.N))
    v.material_id = np.ascontiguousarray(rec["material"].reshape(v.N, v.N, v.N))
    m = a["Material"]
    v.Material = VMaterial(tuple(m["Color"].unpack("<4f")), m["Roughness"].unpack("<f"),
                           m["Metallic"].unpack("<f") if "Metallic" in m else 0.0)
    if "TextureScale" in m:
        v.Material.TextureScale = tuple(m["TextureScale"].unpack("<2f"))
    for key, attr in (("AlbedoTexture", "AlbedoTexturePath"), ("NormalTexture", "NormalTexturePath"), ("RMTexture", "RMTexturePath")):
        if key in m:
            path = _read_cstring(m[key])
            if path and source_path and not os.path.isabs(path):
                path = os.path.join(os.path.dirname(os.path.abspath(source_path)), path)
            setattr(v.Material, attr, path)
    return v


def _level_object(o) -> Archive:
    a = Archive()
    a.props["Position"] = Archive(struct.pack("<3f", *[float(x) for x in o.Position]))
    a.props["Scale"] = Archive(struct.pack("<3f", *[float(x) for x in o.Scale]))
    a.props["Rotation"] = Archive(struct.pack("<4f", *[float(x) for x in o.Rotation]))
    return a


def _read_level_object(a: Archive) -> Tuple[tuple, tuple, tuple]:
    return a["Position"].unpack("<3f"), a["Rotation"].unpack("<4f"), a["Scale"].unpack("<3f")


def _light(l: VLight) -> Archive:
    a = _level_object(l)
    a.props["Color"] = Archive(struct.pack("<4f", *[float(c) for c in l.Color]))
    a.props["Strength"] = Archive(struct.pack("<f", l.IlluminationStrength))
    return a


def save_scene(scene: VScene, path: str) -> None:
    root = Archive()
    vols = scene.volumes()
    root.props["VCount"] = Archive(struct.pack("<Q", len(vols)))
    for i, v in enumerate(vols):
        root.props[f"V_{i}"] = volume_archive(v)
    root.props["OCount"] = Archive(struct.pack("<Q", len(scene.Objects)))
    for i, o in enumerate(scene.Objects):
        root.props[f"OI_{i}"] = Archive(struct.pack("<Q", next(k for k, v in enumerate(vols) if v is o.Volume)))
        root.props[f"O_{i}"] = _level_object(o)
    root.props["LDCount"] = Archive(struct.pack("<Q", 1))
    root.props["LD_0"] = _light(scene.DirectionalLight)
    root.props["LPCount"] = Archive(struct.pack("<Q", len(scene.PointLights)))
    for i, l in enumerate(scene.PointLights):
        a = _light(l)
        a.props["AttL"] = Archive(struct.pack("<f", l.AttenuationLinear))
        a.props["AttExp"] = Archive(struct.pack("<f", l.AttenuationExp))
        root.props[f"LP_{i}"] = a
    root.props["LSCount"] = Archive(struct.pack("<Q", len(scene.SpotLights)))
    for i, l in enumerate(scene.SpotLights):
        a = _light(l)
        a.props["AttL"] = Archive(struct.pack("<f", l.AttenuationLinear))
        a.props["AttExp"] = Archive(struct.pack("<f", l.AttenuationExp))
        a.props["AngleF"] = Archive(struct.pack("<f", l.FalloffAngle))
        a.props["Angle"] = Archive(struct.pack("<f", l.Angle))
        root.props[f"LS_{i}"] = a
    write_archive(root, path)


def load_scene(path: str) -> VScene:
    root = read_archive(path)
    vols = [volume_from_archive(root[f"V_{i}"], path) for i in range(root["VCount"].unpack("<Q"))]
    sc = VScene()
    for i in range(root["OCount"].unpack("<Q")):
        pos, rot, scale = _read_level_object(root[f"O_{i}"])
        sc.Objects.append(VVoxelObject(Position=pos, Rotation=rot, Scale=scale, Volume=vols[root[f"OI_{i}"].unpack("<Q")]))

    def light(a: Archive, cls, **extra):
        pos, rot, scale = _read_level_object(a)
        return cls(Position=pos, Rotation=rot, Scale=scale, IlluminationStrength=a["Strength"].unpack("<f"),
                   Color=a["Color"].unpack("<4f"), **extra)

    n = root["LDCount"].unpack("<Q") if "LDCount" in root else 0
    for i in range(n):
        sc.DirectionalLight = light(root[f"LD_{i}"], VLight)  # the last one becomes active (Scene.cpp:516-520)
    for i in range(root["LPCount"].unpack("<Q") if "LPCount" in root else 0):
        a = root[f"LP_{i}"]
        sc.PointLights.append(light(a, VPointLight, AttenuationLinear=a["AttL"].unpack("<f"), AttenuationExp=a["AttExp"].unpack("<f")))
    for i in range(root["LSCount"].unpack("<Q") if "LSCount" in root else 0):
        a = root[f"LS_{i}"]
        sc.SpotLights.append(light(a, VSpotLight, AttenuationLinear=a["AttL"].unpack("<f"), AttenuationExp=a["AttExp"].unpack("<f"),
                                   FalloffAngle=a["AngleF"].unpack("<f"), Angle=a["Angle"].unpack("<f")))
    return sc

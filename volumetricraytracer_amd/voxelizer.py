"""ctypes access to the C++ Voxelizer (csrc/host, libvrt_host.so) plus small mesh / glTF helpers
used by tests and bench.py.  The conversion itself is the C++ restatement of the reference's
Voxelizer (VolumeConverter.cpp); nothing is voxelized in Python."""
from __future__ import annotations

import base64
import ctypes as C
import json
import math
import os
import struct
from typing import Tuple

import numpy as np

from . import _abi
from .scene import VMaterial, VVoxelVolume

HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvrt_host.so")
_host = None


def load_host() -> C.CDLL:
    global _host
    if _host is None:
        path = os.environ.get("VRT_HOST_LIB") or HOST_LIB_PATH  # VRT_HOST_LIB: the sanitizer build (make -C csrc/host asan)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build()")
        _abi.load()  # libvrt_host.so links libvrt_hip.so: settle the HIP runtime first
        lib = C.CDLL(path)
        lib.vrh_last_error.restype = C.c_char_p
        lib.vrh_convert_mesh.restype = C.c_void_p
        lib.vrh_convert_mesh.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.c_char_p]
        lib.vrh_volume_info.restype = C.c_int
        lib.vrh_volume_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 2 + [C.POINTER(C.c_float)] * 4
        lib.vrh_volume_copy_voxels.restype = C.c_int
        lib.vrh_volume_copy_voxels.argtypes = [C.c_void_p, C.c_void_p]
        lib.vrh_volume_free.restype = None
        lib.vrh_volume_free.argtypes = [C.c_void_p]
        lib.vrh_voxelize_file.restype = C.c_int
        lib.vrh_voxelize_file.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        lib.vrh_vox_rewrite.restype = C.c_int
        lib.vrh_vox_rewrite.argtypes = [C.c_char_p, C.c_char_p]
        _host = lib
    return _host


def convert_mesh(positions: np.ndarray, indices: np.ndarray, bounds_extends, mesh_name: str,
                 material: VMaterial | None = None) -> VVoxelVolume:
    """VVolumeConverter::ConvertMeshInfoToVoxelVolume on arrays that are already in the importer's
    output space (positions x100 and re-centred, bounds = half size + 5)."""
    lib = load_host()
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    be = (C.c_float * 3)(*[float(x) for x in bounds_extends])
    h = lib.vrh_convert_mesh(pos.ctypes.data, len(pos), idx.ctypes.data, len(idx), be, mesh_name.encode())
    if not h:
        raise RuntimeError("vrh_convert_mesh: " + lib.vrh_last_error().decode(errors="replace"))
    try:
        res, size = C.c_int(), C.c_int()
        ext, cell, scale, smax = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        lib.vrh_volume_info(h, C.byref(res), C.byref(size), C.byref(ext), C.byref(cell), C.byref(scale), C.byref(smax))
        vol = VVoxelVolume(res.value, ext.value)
        rec = np.zeros(vol.N ** 3, dtype=np.dtype([("material", "u1"), ("pad", "u1", 3), ("density", "<f4")]))
        lib.vrh_volume_copy_voxels(h, rec.ctypes.data)
        vol.density = np.ascontiguousarray(rec["density"].reshape(vol.N, vol.N, vol.N))
        vol.material_id = np.ascontiguousarray(rec["material"].reshape(vol.N, vol.N, vol.N))
        vol.density_scale = scale.value
        vol.step_max = smax.value
        if material is not None:
            vol.Material = material
        return vol
    finally:
        lib.vrh_volume_free(h)


def voxelize_file(gltf_path: str, out_path: str | None = None, texlib: str | None = None) -> str:
    lib = load_host()
    buf = C.create_string_buffer(4096)
    rc = lib.vrh_voxelize_file(gltf_path.encode(), texlib.encode() if texlib else None, out_path.encode() if out_path else None, buf, 4096)
    if rc != 0:
        raise RuntimeError("vrh_voxelize_file: " + lib.vrh_last_error().decode(errors="replace"))
    return buf.value.decode()


def vox_rewrite(in_path: str, out_path: str) -> None:
    lib = load_host()
    if lib.vrh_vox_rewrite(in_path.encode(), out_path.encode()) != 0:
        raise RuntimeError("vrh_vox_rewrite: " + lib.vrh_last_error().decode(errors="replace"))


# ---- procedural meshes --------------------------------------------------------------------------

def cube_mesh(half: float = 0.5) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """24 vertices (4 per face, with normals), 12 triangles; positions in glTF units."""
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)),
             ((0, -1, 0), (1, 0, 0), (0, 0, 1)), ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0), (1, 0, 0))]
    pos, nrm, idx = [], [], []
    for n, u, w in faces:
        n, u, w = np.array(n, float), np.array(u, float), np.array(w, float)
        base = len(pos)
        for su, sw in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            pos.append((n + su * u + sw * w) * half)
            nrm.append(n)
        idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return np.array(pos, np.float32), np.array(nrm, np.float32), np.array(idx, np.uint32)


def torus_mesh(major: float = 0.55, minor: float = 0.22, nu: int = 128, nv: int = 64) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """UV torus around the Z axis (seedless), positions in glTF units."""
    u = np.arange(nu) * (2 * math.pi / nu)
    v = np.arange(nv) * (2 * math.pi / nv)
    U, V = np.meshgrid(u, v, indexing="ij")
    x = (major + minor * np.cos(V)) * np.cos(U)
    y = (major + minor * np.cos(V)) * np.sin(U)
    z = minor * np.sin(V)
    pos = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    nrm = np.stack([np.cos(V) * np.cos(U), np.cos(V) * np.sin(U), np.sin(V)], -1).reshape(-1, 3).astype(np.float32)
    idx = []
    for i in range(nu):
        for j in range(nv):
            a = i * nv + j
            b = ((i + 1) % nu) * nv + j
            c = ((i + 1) % nu) * nv + (j + 1) % nv
            d = i * nv + (j + 1) % nv
            idx += [a, b, c, a, c, d]
    return pos, nrm, np.array(idx, np.uint32)


def importer_space(positions: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """What VGLTFImporter does to POSITION data (GLTFImporter.cpp:52-64,111-121): scale by 100,
    re-centre on the accessor's min/max midpoint; bounds extents = half size + 5."""
    p = positions.astype(np.float32)
    mn = p.min(0) * np.float32(100.0)
    mx = p.max(0) * np.float32(100.0)
    ext = (mx - mn) * np.float32(0.5)
    off = mx - ext
    return p * np.float32(100.0) - off, ext + np.float32(5.0)


def write_gltf(path: str, meshes, nodes, materials=None, embed: bool = False) -> None:
    """Minimal .gltf (+ .bin) writer.  meshes: list of (name, positions, normals, indices, material
    index or None); nodes: list of dicts (name, mesh, translation, rotation, scale, extras)."""
    blob = bytearray()
    views, accessors, gm = [], [], []

    def add(data: np.ndarray, target: int, comp: int, typ: str, minmax: bool):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": data.nbytes, "target": target})
        blob.extend(data.tobytes())
        acc = {"bufferView": len(views) - 1, "componentType": comp, "count": int(data.shape[0]), "type": typ}
        if minmax:
            acc["min"] = [float(x) for x in data.min(0)]
            acc["max"] = [float(x) for x in data.max(0)]
        accessors.append(acc)
        return len(accessors) - 1

    for name, pos, nrm, idx, mat in meshes:
        use16 = len(pos) < 65536
        ia = add(idx.astype(np.uint16 if use16 else np.uint32).reshape(-1), 34963, 5123 if use16 else 5125, "SCALAR", False)
        pa = add(pos.astype(np.float32), 34962, 5126, "VEC3", True)
        na = add(nrm.astype(np.float32), 34962, 5126, "VEC3", False)
        prim = {"attributes": {"POSITION": pa, "NORMAL": na}, "indices": ia}
        if mat is not None:
            prim["material"] = mat
        gm.append({"name": name, "primitives": [prim]})
    bin_name = os.path.splitext(os.path.basename(path))[0] + ".bin"
    buf = {"byteLength": len(blob)}
    buf["uri"] = ("data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()) if embed else bin_name
    doc = {"asset": {"version": "2.0", "generator": "volumetricraytracer_amd tests"}, "buffers": [buf], "bufferViews": views,
           "accessors": accessors, "meshes": gm, "nodes": nodes, "scenes": [{"nodes": list(range(len(nodes)))}], "scene": 0}
    if materials:
        doc["materials"] = materials
    with open(path, "w") as f:
        json.dump(doc, f)
    if not embed:
        with open(os.path.join(os.path.dirname(path), bin_name), "wb") as f:
            f.write(bytes(blob))


def gltf_to_glb(gltf_path: str, glb_path: str) -> None:
    """Packs a .gltf with ONE external .bin buffer into a binary .glb container (glTF 2.0 spec §4.4)."""
    import json
    import struct

    doc = json.load(open(gltf_path))
    if len(doc.get("buffers", [])) != 1 or doc["buffers"][0].get("uri", "").startswith("data:"):
        raise ValueError("expects exactly one external buffer")
    blob = open(os.path.join(os.path.dirname(gltf_path), doc["buffers"][0].pop("uri")), "rb").read()
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * (-len(js) % 4)
    blob += b"\0" * (-len(blob) % 4)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(glb_path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(blob), 0x004E4942) + blob)


def load_texture(path: str) -> np.ndarray:
    """Decodes a material texture file (PNG or binary PPM) with the C++ host loader VHipRenderer uses for
    VMaterial texture paths; returns uint8 [H, W, 4]."""
    lib = load_host()
    lib.vrh_texture_load.restype = C.c_int
    lib.vrh_texture_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
    w, h = C.c_int(), C.c_int()
    if lib.vrh_texture_load(path.encode(), C.byref(w), C.byref(h), None, 0) != 0:
        raise RuntimeError("vrh_texture_load: " + lib.vrh_last_error().decode(errors="replace"))
    out = np.zeros((h.value, w.value, 4), np.uint8)
    if lib.vrh_texture_load(path.encode(), None, None, out.ctypes.data, out.nbytes) != 0:
        raise RuntimeError("vrh_texture_load: " + lib.vrh_last_error().decode(errors="replace"))
    return out


def load_skybox_faces(directory: str) -> np.ndarray:
    """Sky box from <directory>/XP.png ... ZM.png (the reference's Resources/Skybox layout), or from a .dds cube map (what
    VTextureFactory::LoadTextureCubeFromFile reads: uncompressed RGBA / BGRA / BGRX / RGB, top mip), through the C++ host
    loader; returns uint8 [6, S, S, 4] in the order vrt_env_upload / VScene.EnvironmentMap expect (+X,-X,+Y,-Y,+Z,-Z)."""
    lib = load_host()
    lib.vrh_cubemap_load.restype = C.c_int
    lib.vrh_cubemap_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
    size = C.c_int()
    if lib.vrh_cubemap_load(directory.encode(), C.byref(size), None, 0) != 0:
        raise RuntimeError("vrh_cubemap_load: " + lib.vrh_last_error().decode(errors="replace"))
    out = np.zeros((6, size.value, size.value, 4), np.uint8)
    if lib.vrh_cubemap_load(directory.encode(), None, out.ctypes.data, out.nbytes) != 0:
        raise RuntimeError("vrh_cubemap_load: " + lib.vrh_last_error().decode(errors="replace"))
    return out

"""ctypes binding of the CPU oracle (oracle/_build/libvrt_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never from volumetricraytracer_amd/.  PARITY UNPINNED by reference tests
(the reference has none); see vrt_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

from volumetricraytracer_amd import _abi
from volumetricraytracer_amd.scene import VScene

_HERE = os.path.dirname(os.path.abspath(__file__))
# VRT_ORACLE_LIB: another build of the same oracle (the sanitizer build of `make -C oracle asan`)
LIB_PATH = os.environ.get("VRT_ORACLE_LIB") or os.path.join(_HERE, "_build", "libvrt_oracle.so")


class vrto_texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class vrto_volume(C.Structure):
    _fields_ = [
        ("density", C.c_void_p),
        ("resolution", C.c_int32),
        ("extent", C.c_float),
        ("density_scale", C.c_float),
        ("step_max", C.c_float),
        ("material", _abi.vrt_material),
        ("albedo_tex", vrto_texture),
        ("normal_tex", vrto_texture),
        ("rm_tex", vrto_texture),
        ("tex_scale", C.c_float * 2),
        ("format", C.c_int32),
    ]


class vrto_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("primary_rays", "shadow_rays", "bounce_rays", "primary_steps", "shadow_steps", "hits", "exhausted_rays")]


class vrto_literal_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays", "iterations", "solid_start_hits", "entry_hits", "root_hits", "tail_hits", "red_hits", "rejected_reports")]


class vrto_octree_info(C.Structure):
    _fields_ = [("nodes", C.c_uint64), ("leaves_at_depth", C.c_uint64 * 9), ("texture_edge", C.c_int32), ("pointer_overflow", C.c_int32)]


LIT_NORMALISED_CAMERA = 1

_lib = None


def build() -> None:
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.vrto_render.restype = C.c_int
        lib.vrto_render.argtypes = [C.POINTER(_abi.vrt_scene), C.POINTER(vrto_volume), C.c_void_p, C.c_int,
                                    C.POINTER(_abi.vrt_params), C.c_int, C.c_int, C.c_void_p,
                                    C.POINTER(vrto_stats), C.c_int]
        lib.vrto_trace.restype = C.c_int
        lib.vrto_trace.argtypes = [C.POINTER(_abi.vrt_scene), C.POINTER(vrto_volume), C.POINTER(_abi.vrt_params),
                                   C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.vrto_camera_ray.restype = None
        lib.vrto_camera_ray.argtypes = [C.POINTER(_abi.vrt_scene), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.vrto_sample.restype = C.c_float
        lib.vrto_sample.argtypes = [C.POINTER(vrto_volume), C.POINTER(C.c_float)]
        lib.vrto_ref_hit_t.restype = C.c_int
        lib.vrto_ref_hit_t.argtypes = [C.POINTER(vrto_volume), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                       C.POINTER(C.c_double)]
        lib.vrto_trace_batch.restype = C.c_int
        lib.vrto_trace_batch.argtypes = [C.POINTER(_abi.vrt_scene), C.POINTER(vrto_volume), C.POINTER(_abi.vrt_params), C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.vrto_ref_hit_batch.restype = C.c_int
        lib.vrto_ref_hit_batch.argtypes = [C.POINTER(vrto_volume), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.vrto_ref_render.restype = C.c_int
        lib.vrto_ref_render.argtypes = [C.POINTER(_abi.vrt_scene), C.POINTER(vrto_volume), C.c_void_p, C.c_int,
                                        C.POINTER(_abi.vrt_params), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.vrto_ref_literal_render.restype = C.c_int
        lib.vrto_ref_literal_render.argtypes = [C.POINTER(_abi.vrt_scene), C.POINTER(vrto_volume), C.c_void_p, C.c_int,
                                                C.POINTER(_abi.vrt_params), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint,
                                                C.POINTER(vrto_literal_stats), C.c_int]
        lib.vrto_literal_octree_info.restype = C.c_int
        lib.vrto_literal_octree_info.argtypes = [C.POINTER(vrto_volume), C.POINTER(vrto_octree_info)]
        lib.vrto_debug_tables.restype = C.c_int
        lib.vrto_debug_tables.argtypes = [C.POINTER(vrto_volume), C.c_void_p, C.c_void_p, C.c_void_p]
        lib.vrto_env_lookup.restype = None
        lib.vrto_env_lookup.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = lib
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class OracleScene:
    """Holds the ctypes image of a VScene (+ keeps the numpy buffers alive)."""

    def __init__(self, scene: VScene):
        self.scene = scene
        self.abi = scene.to_abi()
        self.vols = (vrto_volume * _abi.VRT_MAX_VOLUMES)()
        self._keep = []
        for slot, v in enumerate(scene.volumes()):
            d = np.ascontiguousarray(v.density, dtype=np.float32)
            self._keep.append(d)
            o = self.vols[slot]
            o.density = d.ctypes.data
            o.resolution = v.Resolution
            o.extent = v.VolumeExtends
            o.density_scale = v.density_scale
            o.step_max = v.step_max
            o.format = int(v.device_format)
            o.material = v.Material.to_abi()
            for name, img in zip(("albedo_tex", "normal_tex", "rm_tex"), v.Material.textures()):
                if img is not None:
                    self._keep.append(img)
                    t = getattr(o, name)
                    t.rgba8 = img.ctypes.data
                    t.width, t.height = img.shape[1], img.shape[0]
            o.tex_scale[0] = float(v.Material.TextureScale[0])
            o.tex_scale[1] = float(v.Material.TextureScale[1])
        self.env = None
        self.env_size = 0
        if scene.EnvironmentMap is not None:
            self.env = np.ascontiguousarray(scene.EnvironmentMap, dtype=np.uint8)
            self.env_size = int(self.env.shape[1])

    def render(self, params: _abi.vrt_params, row0: int = 0, rows: Optional[int] = None,
               threads: int = 1) -> Tuple[np.ndarray, dict]:
        lib = load()
        rows = params.height - row0 if rows is None else rows
        out = np.empty((rows, params.width, 4), dtype=np.float32)
        st = vrto_stats()
        rc = lib.vrto_render(C.byref(self.abi), self.vols, self.env.ctypes.data if self.env is not None else None,
                             self.env_size, C.byref(params), row0, rows, out.ctypes.data, C.byref(st), threads)
        if rc != 0:
            raise RuntimeError(f"vrto_render failed: {rc}")
        return out, {n: getattr(st, n) for n, _ in st._fields_}

    def ref_render(self, params: _abi.vrt_params, row0: int = 0, rows: Optional[int] = None, threads: int = 8):
        """vrto_ref_render: the frame the REFERENCE's intersection (exact cubic root, normal at the root) would shade.
        Returns (image [rows, W, 4] float32, t [rows, W] float32 with -1 for camera rays that miss)."""
        rows = params.height - row0 if rows is None else rows
        out = np.empty((rows, params.width, 4), dtype=np.float32)
        t = np.empty((rows, params.width), dtype=np.float32)
        rc = load().vrto_ref_render(C.byref(self.abi), self.vols, self.env.ctypes.data if self.env is not None else None,
                                    self.env_size, C.byref(params), row0, rows, out.ctypes.data, t.ctypes.data, threads)
        if rc != 0:
            raise RuntimeError(f"vrto_ref_render failed: {rc}")
        return out, t

    def ref_literal_render(self, params: _abi.vrt_params, row0: int = 0, rows: Optional[int] = None, threads: int = 8, options: int = 0):
        """vrto_ref_literal_render: the frame the reference's shaders compute LITERALLY (fp32, un-normalised camera direction, nudges,
        octree leaves, three secant steps, abs()-weighted normal, budget).  Returns (image, t in world units with -1 for misses, stats)."""
        rows = params.height - row0 if rows is None else rows
        out = np.empty((rows, params.width, 4), dtype=np.float32)
        t = np.empty((rows, params.width), dtype=np.float32)
        st = vrto_literal_stats()
        rc = load().vrto_ref_literal_render(C.byref(self.abi), self.vols, self.env.ctypes.data if self.env is not None else None,
                                            self.env_size, C.byref(params), row0, rows, out.ctypes.data, t.ctypes.data, options,
                                            C.byref(st), threads)
        if rc != 0:
            raise RuntimeError(f"vrto_ref_literal_render failed: {rc}")
        return out, t, {n: getattr(st, n) for n, _ in st._fields_}

    def octree_info(self, slot: int) -> dict:
        """The collapsed octree the reference would build for the volume in `slot` (VCellOctree)."""
        info = vrto_octree_info()
        rc = load().vrto_literal_octree_info(C.byref(self.vols[slot]), C.byref(info))
        if rc != 0:
            raise RuntimeError(f"vrto_literal_octree_info failed: {rc}")
        return {"nodes": info.nodes, "leaves_at_depth": list(info.leaves_at_depth), "texture_edge": info.texture_edge,
                "pointer_overflow": bool(info.pointer_overflow)}

    def trace(self, params: _abi.vrt_params, origin, direction, t_max: float = 10000.0):
        lib = load()
        t = C.c_float()
        n = (C.c_float * 3)()
        inst = C.c_int(-1)
        steps = C.c_int()
        hit = lib.vrto_trace(C.byref(self.abi), self.vols, C.byref(params), _f3(origin), _f3(direction), t_max,
                             C.byref(t), n, C.byref(inst), C.byref(steps))
        if hit < 0:
            raise RuntimeError(f"vrto_trace failed: {hit}")
        return bool(hit), t.value, np.array(list(n), dtype=np.float32), inst.value, steps.value

    def trace_batch(self, params: _abi.vrt_params, origins, directions, t_max: float = 10000.0, threads: int = 8):
        """vrto_trace for many world-space rays with the scene set up once: (hit [n] bool, t [n] float32, normal [n, 3])."""
        org = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        dr = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = len(org)
        hit, t, nrm = np.zeros(n, np.uint8), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
        rc = load().vrto_trace_batch(C.byref(self.abi), self.vols, C.byref(params), n, org.ctypes.data, dr.ctypes.data, t_max,
                                     hit.ctypes.data, t.ctypes.data, nrm.ctypes.data, threads)
        if rc != 0:
            raise RuntimeError(f"vrto_trace_batch failed: {rc}")
        return hit.astype(bool), t, nrm

    def ref_hit_batch(self, slot: int, origins, directions, threads: int = 8):
        """vrto_ref_hit_t (the reference's DDA + per-cell cubic, double precision) for many OBJECT-space rays of one volume."""
        org = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        dr = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = len(org)
        hit, t = np.zeros(n, np.uint8), np.zeros(n, np.float64)
        rc = load().vrto_ref_hit_batch(C.byref(self.vols[slot]), n, org.ctypes.data, dr.ctypes.data, hit.ctypes.data, t.ctypes.data, threads)
        if rc != 0:
            raise RuntimeError(f"vrto_ref_hit_batch failed: {rc}")
        return hit.astype(bool), t

    def camera_rays(self, width: int, height: int, pixels):
        """Camera rays of many (px, py) pixels: origins [n, 3], directions [n, 3]."""
        o = np.zeros((len(pixels), 3), np.float32)
        d = np.zeros((len(pixels), 3), np.float32)
        for i, (px, py) in enumerate(pixels):
            o[i], d[i] = self.camera_ray(width, height, int(px), int(py))
        return o, d

    def camera_ray(self, width: int, height: int, px: int, py: int):
        lib = load()
        o, d = (C.c_float * 3)(), (C.c_float * 3)()
        lib.vrto_camera_ray(C.byref(self.abi), width, height, px, py, o, d)
        return np.array(list(o), dtype=np.float32), np.array(list(d), dtype=np.float32)

    def tables(self, slot: int):
        """(skip [nb,nb,nb] uint8 Chebyshev brick distances, nib [nb,nb,nb] uint32 sub-block nibbles, field [N,N,N]) of a
        bounded-step volume, indexed [x, z, y] like the grid."""
        vol = self.scene.volumes()[slot]
        nb = (vol.N - 1 + 3) // 4
        skip = np.zeros((nb, nb, nb), np.uint8)
        nib = np.zeros((nb, nb, nb), np.uint32)
        field = np.zeros((vol.N,) * 3, np.float32)
        rc = load().vrto_debug_tables(C.byref(self.vols[slot]), skip.ctypes.data, nib.ctypes.data, field.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"vrto_debug_tables failed: {rc}")
        return skip, nib, field

    def sample(self, slot: int, p) -> float:
        return float(load().vrto_sample(C.byref(self.vols[slot]), _f3(p)))

    def ref_hit_t(self, slot: int, origin, direction):
        t = C.c_double()
        hit = load().vrto_ref_hit_t(C.byref(self.vols[slot]), _f3(origin), _f3(direction), C.byref(t))
        return bool(hit), t.value


def env_lookup(env: np.ndarray, direction) -> np.ndarray:
    e = np.ascontiguousarray(env, dtype=np.uint8)
    out = (C.c_float * 3)()
    load().vrto_env_lookup(e.ctypes.data, int(e.shape[1]), _f3(direction), out)
    return np.array(list(out), dtype=np.float32)

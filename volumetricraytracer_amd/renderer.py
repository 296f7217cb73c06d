"""Python host wrapper over the C-ABI: the `VRenderer` surface of the reference
(Renderer/Public/Renderer.h:44-66 — Start/Stop/IsActive/SetSceneToRender/Render/
ResizeRenderOutput/SetRendererMode) for the HIP backend.  Used by tests and bench.py; the C++
adaptor with the same shape is csrc/host/HipRenderer.h.

Everything here calls into libvrt_hip.so.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _abi
from .scene import VScene, VVoxelVolume, default_params, march_budget


class VHipRenderer:
    def __init__(self, devices: Sequence[int] = (0,)):
        self._lib = _abi.load()
        self._devices = list(devices)
        self._ctx = C.c_void_p()
        self._scene: Optional[VScene] = None
        self._slots: List[VVoxelVolume] = []
        self._uploaded: Dict[int, int] = {}  # slot -> id(volume) currently on the device
        self.RenderMode = _abi.MODE_INTERP
        self.Width, self.Height = 1024, 576  # default window, UI/Win/Private/Win32Window.cpp:218-219
        self.params_override: Optional[_abi.vrt_params] = None
        self.DataPath = _abi.PATH_AUTO
        self.Shadows = True
        self.MaxSteps = 255  # Raytracing.hlsl:229
        self.MaxBounces = 2  # MAX_RAY_RECURSION_DEPTH 3 = primary + 2 mirror bounces (RaytracingHlsl.h:32)
        # the reference's artefacts the C++ adaptor reproduces by default (VHipRenderer::ReferenceViewVector / ReferenceBoundaryTexels); this
        # mirror, like the C-ABI, leaves them off unless asked
        self.ReferenceViewVector = False
        self.ReferenceBoundaryTexels = False
        self._env_id = None
        self._tex_ids: Dict[int, int] = {}   # id(image array) -> texture id on the device
        self._tex_keep: Dict[int, np.ndarray] = {}  # keeps the keyed arrays alive so id() stays unique
        self._bound_tex: Dict[int, tuple] = {}  # slot -> the (albedo, normal, rm) texture ids bound to it

    # -- VRenderer surface -------------------------------------------------------------------
    def Start(self) -> bool:
        """VDXRenderer::Start: bring the device up; False (after logging) on failure."""
        if self._ctx:
            return True
        arr = (C.c_int * len(self._devices))(*self._devices)
        rc = self._lib.vrt_create(C.byref(self._ctx), len(self._devices), arr)
        if rc != _abi.VRT_OK:
            print(f"[VHipRenderer] Start failed: {self._lib.vrt_strerror(rc).decode()}")
            self._ctx = C.c_void_p()
            return False
        return True

    def Stop(self) -> None:
        if self._ctx:
            self._lib.vrt_destroy(self._ctx)
            self._ctx = C.c_void_p()
            self._uploaded.clear()
            self._env_id = None
            self._tex_ids.clear()
            self._tex_keep.clear()
            self._bound_tex.clear()

    def IsActive(self) -> bool:
        return bool(self._ctx)

    def SetSceneToRender(self, scene: VScene) -> None:
        self._scene = scene

    def SetRendererMode(self, mode: int) -> None:
        self.RenderMode = int(mode)

    def ResizeRenderOutput(self, width: int, height: int) -> None:
        self.Width, self.Height = int(width), int(height)

    def Render(self) -> np.ndarray:
        """One frame: sync scene → device, march, return the RGBA image [H, W, 4] (float32, or uint8 when
        params_override sets FLAG_OUTPUT_RGBA8)."""
        self._require()
        self.SyncWithScene()
        p = self.make_params()
        out = np.empty((p.height, p.width, 4), dtype=np.uint8 if p.flags & _abi.FLAG_OUTPUT_RGBA8 else np.float32)
        _abi.check(self._lib.vrt_render(self._ctx, C.byref(p), out.ctypes.data_as(C.c_void_p)), "vrt_render")
        return out

    # -- scene mirroring (VRDXScene::SyncWithScene) --------------------------------------------
    def SyncWithScene(self) -> None:
        self._require()
        sc = self._scene
        if sc is None:
            raise RuntimeError("SetSceneToRender was not called")
        vols = sc.volumes()
        # Material textures no volume of the NEW scene names leave the device first — after every slot that still points at one of
        # them has been un-bound (a slot must never name an id the next upload reuses) — so that a scene swap never needs the old
        # and the new texture sets resident together (VRT_MAX_TEXTURES ids in all).
        live = {id(t) for vol in vols for t in vol.Material.textures() if t is not None}
        dead = {self._tex_ids[k] for k in self._tex_ids if k not in live}
        if dead:
            for slot, bound in list(self._bound_tex.items()):
                if any(i in dead for i in bound):
                    _abi.check(self._lib.vrt_volume_set_textures(self._ctx, slot, -1, -1, -1, 100.0, 100.0), "vrt_volume_set_textures")
                    self._bound_tex[slot] = (-1, -1, -1)
            for key in [k for k in self._tex_ids if k not in live]:
                _abi.check(self._lib.vrt_texture_free(self._ctx, self._tex_ids[key]), "vrt_texture_free")
                del self._tex_ids[key]
                del self._tex_keep[key]
        for slot, vol in enumerate(vols):
            if self._uploaded.get(slot) != id(vol) or vol.dirty:
                self.upload_volume(slot, vol)
            else:
                # the volume's voxels are unchanged, but its material may name other images or scalars than at the last sync
                # (VMaterial is edited in place; VDXVoxelVolume::UpdateGeometryConstantBuffer, RDXVoxelVolume.cpp:368-397, runs
                # every frame): re-send them — the C-ABI returns at once when nothing differs
                self._bind_material(slot, vol)
        for slot in [s for s in self._uploaded if s >= len(vols)]:
            _abi.check(self._lib.vrt_volume_free(self._ctx, slot), "vrt_volume_free")
            del self._uploaded[slot]
            self._bound_tex.pop(slot, None)
        env = sc.EnvironmentMap
        if env is None:
            if self._env_id is not None:
                _abi.check(self._lib.vrt_env_upload(self._ctx, 0, None), "vrt_env_upload")
                self._env_id = None
        elif self._env_id != id(env):
            e = np.ascontiguousarray(env, dtype=np.uint8)
            if e.ndim != 4 or e.shape[0] != 6 or e.shape[1] != e.shape[2] or e.shape[3] != 4:
                raise ValueError("EnvironmentMap must be uint8 [6, S, S, 4]")
            _abi.check(self._lib.vrt_env_upload(self._ctx, e.shape[1], e.ctypes.data_as(C.c_void_p)), "vrt_env_upload")
            self._env_id = id(env)
        abi_scene = sc.to_abi()
        _abi.check(self._lib.vrt_scene_set(self._ctx, C.byref(abi_scene)), "vrt_scene_set")

    def upload_volume(self, slot: int, vol: VVoxelVolume, as_voxels: bool = False, as_texels: bool = False) -> None:
        """vrt_volume_upload (or _upload_voxels: VVoxel records; or _upload_texels: the reference's own RGBA8 volume
        texture, always the 16-bit format) in the volume's device_format."""
        self._require()
        _abi.check(self._lib.vrt_set_volume_format(self._ctx, int(vol.device_format)), "vrt_set_volume_format")
        if as_texels:
            tex = vol.reference_texels()
            rc = self._lib.vrt_volume_upload_texels(self._ctx, slot, vol.Resolution, vol.VolumeExtends, tex.ctypes.data_as(C.c_void_p))
        elif as_voxels:
            rec = vol.voxel_records()
            rc = self._lib.vrt_volume_upload_voxels(self._ctx, slot, vol.Resolution, vol.VolumeExtends,
                                                    rec.ctypes.data_as(C.c_void_p))
        else:
            d = np.ascontiguousarray(vol.density, dtype=np.float32)
            m = np.ascontiguousarray(vol.material_id, dtype=np.uint8)
            rc = self._lib.vrt_volume_upload(self._ctx, slot, vol.Resolution, vol.VolumeExtends,
                                             d.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p))
        _abi.check(rc, "vrt_volume_upload")
        _abi.check(self._lib.vrt_volume_set_metric(self._ctx, slot, float(vol.density_scale), float(vol.step_max)),
                   "vrt_volume_set_metric")
        self._bind_material(slot, vol)
        self._uploaded[slot] = id(vol)
        vol.dirty = False

    def _bind_material(self, slot: int, vol: VVoxelVolume) -> None:
        """The slot's material scalars and texture bindings as the volume's VMaterial holds them NOW (images uploaded on first
        sight).  Both C-ABI calls return without touching the device when nothing changed."""
        mat = vol.Material.to_abi()
        _abi.check(self._lib.vrt_volume_set_material(self._ctx, slot, C.byref(mat)), "vrt_volume_set_material")
        ids = [self.upload_texture(t) if t is not None else -1 for t in vol.Material.textures()]
        _abi.check(self._lib.vrt_volume_set_textures(self._ctx, slot, ids[0], ids[1], ids[2], float(vol.Material.TextureScale[0]),
                                                     float(vol.Material.TextureScale[1])), "vrt_volume_set_textures")
        self._bound_tex[slot] = tuple(ids)

    def voxelize_mesh(self, slot: int, positions: np.ndarray, indices: np.ndarray, resolution: int, extent: float) -> int:
        """The Voxelizer's hot loop on the device (vrt_voxelize_mesh): fills `slot` with the shell field of a
        triangle mesh given in volume space; returns the number of skipped (degenerate) triangles.  The slot is
        not tracked by SyncWithScene: pair it with download_volume() or instance it through a raw vrt_scene."""
        self._require()
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        skipped = C.c_size_t(0)
        _abi.check(self._lib.vrt_voxelize_mesh(self._ctx, slot, resolution, float(extent), pos.ctypes.data_as(C.c_void_p), len(pos),
                                               idx.ctypes.data_as(C.c_void_p), len(idx), C.byref(skipped)), "vrt_voxelize_mesh")
        self._uploaded.pop(slot, None)
        return int(skipped.value)

    def download_volume(self, slot: int, resolution: int, extent: float) -> VVoxelVolume:
        """vrt_volume_download: the slot as a VVoxelVolume (densities + materials)."""
        self._require()
        vol = VVoxelVolume(resolution, extent)
        rec = np.zeros(vol.N ** 3, dtype=np.dtype([("material", "u1"), ("pad", "u1", 3), ("density", "<f4")]))
        _abi.check(self._lib.vrt_volume_download(self._ctx, slot, rec.ctypes.data_as(C.c_void_p)), "vrt_volume_download")
        vol.density = np.ascontiguousarray(rec["density"].reshape(vol.N, vol.N, vol.N))
        vol.material_id = np.ascontiguousarray(rec["material"].reshape(vol.N, vol.N, vol.N))
        return vol

    def upload_texture(self, image: np.ndarray) -> int:
        """VRenderer::InitializeTexture + UploadToGPU for a 2D material texture (uint8 [H, W, 4]); images are
        shared between volumes by identity, like the reference's path-keyed texture table (RDXScene.cpp:905-925)."""
        self._require()
        key = id(image)
        if key in self._tex_ids:
            return self._tex_ids[key]
        used = set(self._tex_ids.values())
        tid = next((i for i in range(_abi.VRT_MAX_TEXTURES) if i not in used), None)
        if tid is None:
            raise RuntimeError("too many material textures")
        img = np.ascontiguousarray(image, dtype=np.uint8)
        _abi.check(self._lib.vrt_texture_upload(self._ctx, tid, img.shape[1], img.shape[0], img.ctypes.data_as(C.c_void_p)),
                   "vrt_texture_upload")
        self._tex_ids[key] = tid
        self._tex_keep[key] = image
        return tid

    # -- parameters / raw launches ---------------------------------------------------------------
    def make_params(self) -> _abi.vrt_params:
        if self.params_override is not None:
            p = _abi.vrt_params.from_buffer_copy(self.params_override)
        else:
            cell = min((v.GetCellSize() for v in self._scene.volumes()), default=1.0) if self._scene else 1.0
            res = max((v.Resolution for v in self._scene.volumes()), default=0) if self._scene else 0
            p = default_params(self.Width, self.Height, cell, max_steps=march_budget(res, self.MaxSteps), shadow=self.Shadows)
            p.max_bounces = self.MaxBounces
            if self.ReferenceViewVector:
                p.flags |= _abi.FLAG_REFERENCE_VIEW_VECTOR
            if self.ReferenceBoundaryTexels:
                p.flags |= _abi.FLAG_REFERENCE_BOUNDARY_TEXELS
        p.width, p.height = self.Width, self.Height
        p.mode = self.RenderMode
        if self.params_override is None:
            p.path = self.DataPath
        return p

    def render_rows(self, params: _abi.vrt_params, row0: int, rows: int, device_ptr: int, stream: int = 0) -> None:
        """Asynchronous tile render into caller-owned device memory (vrt_render_rows)."""
        self._require()
        _abi.check(self._lib.vrt_render_rows(self._ctx, C.byref(params), row0, rows, C.c_void_p(device_ptr),
                                             C.c_void_p(stream)), "vrt_render_rows")

    def render_strips(self, params: _abi.vrt_params, strip_rows: int, first_strip: int, strip_stride: int, n_strips: int,
                      device_ptr: int, stream: int = 0) -> None:
        """Asynchronous render of interleaved strips into a compact device tile (vrt_render_strips)."""
        self._require()
        _abi.check(self._lib.vrt_render_strips(self._ctx, C.byref(params), strip_rows, first_strip, strip_stride, n_strips,
                                               C.c_void_p(device_ptr), C.c_void_p(stream)), "vrt_render_strips")

    def render_block(self, params: _abi.vrt_params, n_frames: int, device_ptr: int, frame_stride_bytes: int, stream: int = 0,
                     strips=None, rows=None, cameras=None, scenes=None) -> None:
        """vrt_render_block: n_frames frames of the current scene in flight with one call, frame f into
        device_ptr + f*frame_stride_bytes; to the caller one asynchronous operation on `stream`.  strips = (strip_rows,
        first_strip, strip_stride, n_strips) or rows = (row0, rows) (default: the whole frame); cameras: n_frames
        (position[3], rotation[4], fov_deg) triples overriding the scene's camera per frame; scenes: per-frame scene state —
        (a ctypes vrt_scene array from scene_array(), index of the first scene): frame f renders scenes[start + f] (objects,
        lights and camera that move from frame to frame) over the volumes of the scene this renderer is synced with."""
        self._require()
        b = _abi.vrt_block()
        b.n_frames = int(n_frames)
        if strips is not None:
            b.strip_rows, b.first_strip, b.strip_stride, b.n_strips = (int(x) for x in strips)
        else:
            b.row0, b.rows = (int(x) for x in rows) if rows is not None else (0, int(params.height))
        if cameras is not None:
            # a list of (position, rotation, fov) triples, or (a ctypes array from camera_array(), index of the first camera)
            arr, start = cameras if isinstance(cameras, tuple) and isinstance(cameras[0], C.Array) else (self.camera_array(cameras), 0)
            if start < 0 or start + b.n_frames > len(arr):
                raise ValueError("camera range outside the array")
            b.cameras = C.cast(C.byref(arr, start * C.sizeof(_abi.vrt_camera)), C.POINTER(_abi.vrt_camera))
        if scenes is not None:
            sarr, sstart = scenes
            if sstart < 0 or sstart + b.n_frames > len(sarr):
                raise ValueError("scene range outside the array")
            b.scenes = C.cast(C.byref(sarr, sstart * C.sizeof(_abi.vrt_scene)), C.POINTER(_abi.vrt_scene))
        b.frame_stride_bytes = int(frame_stride_bytes)
        _abi.check(self._lib.vrt_render_block(self._ctx, C.byref(params), C.byref(b), C.c_void_p(device_ptr), C.c_void_p(stream)),
                   "vrt_render_block")

    def scene_array(self, scenes):
        """A ctypes vrt_scene array from VScene objects that place (a subset of) the SAME volumes as the scene this renderer
        is synced with (SetSceneToRender + SyncWithScene): their instances name the synced scene's volume slots.  For
        render_block(scenes=(array, start)): objects, lights and camera per frame."""
        if self._scene is None:
            raise RuntimeError("SetSceneToRender was not called")
        slots = {id(vol): i for i, vol in enumerate(self._scene.volumes())}
        arr = (_abi.vrt_scene * len(scenes))()
        for k, sc in enumerate(scenes):
            a = sc.to_abi()
            own = sc.volumes()
            for i in range(a.n_instances):
                vol = own[a.instances[i].volume_slot]
                if id(vol) not in slots:
                    raise ValueError("a per-frame scene places a volume the synced scene does not hold")
                a.instances[i].volume_slot = slots[id(vol)]
            arr[k] = a
        return arr

    @staticmethod
    def camera_array(cameras):
        """A ctypes vrt_camera array from (position[3], rotation[4], fov_deg) triples, to be passed (with a start index) to
        render_block many times without rebuilding it."""
        arr = (_abi.vrt_camera * len(cameras))()
        for c, (pos, rot, fov) in zip(arr, cameras):
            c.position[:] = [float(x) for x in pos]
            c.rotation[:] = [float(x) for x in rot]
            c.fov_deg = float(fov)
        return arr

    def render_begin(self, slot: int, params: Optional[_abi.vrt_params] = None) -> _abi.vrt_params:
        """vrt_render_begin: sync the scene, snapshot it and enqueue the whole frame on frame slot `slot`
        (0..VRT_FRAMES_IN_FLIGHT-1); returns at once.  Collect with render_end(slot, params)."""
        self._require()
        self.SyncWithScene()
        p = params if params is not None else self.make_params()
        _abi.check(self._lib.vrt_render_begin(self._ctx, C.byref(p), slot), "vrt_render_begin")
        return p

    def render_end(self, slot: int, params: _abi.vrt_params, copy: bool = True) -> Optional[np.ndarray]:
        """vrt_render_end: wait for the frame begun on `slot` and return a copy of its pixels (copy=False: just wait;
        the pixels stay in the slot's pinned host frame)."""
        self._require()
        ptr = C.c_void_p()
        _abi.check(self._lib.vrt_render_end(self._ctx, slot, C.byref(ptr)), "vrt_render_end")
        if not copy:
            return None
        if params.flags & _abi.FLAG_OUTPUT_RGBA8:
            buf = (C.c_uint8 * (params.width * params.height * 4)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=np.uint8).reshape(params.height, params.width, 4).copy()
        buf = (C.c_float * (params.width * params.height * 4)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.float32).reshape(params.height, params.width, 4).copy()

    # -- multi-GPU exchange, one process per GPU (vrt_comm_* / vrt_gather_tiles: RCCL ncclGather) --------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: the 128-byte communicator id to hand to every rank (vrt_comm_unique_id)."""
        buf = (C.c_uint8 * _abi.VRT_COMM_ID_BYTES)()
        _abi.check(_abi.load().vrt_comm_unique_id(buf), "vrt_comm_unique_id")
        return bytes(buf)

    def comm_init(self, world: int, rank: int, unique_id: bytes) -> None:
        """Every rank (collective): join the communicator of `unique_id` on this renderer's device."""
        self._require()
        if len(unique_id) != _abi.VRT_COMM_ID_BYTES:
            raise ValueError("unique_id must be VRT_COMM_ID_BYTES long")
        buf = (C.c_uint8 * _abi.VRT_COMM_ID_BYTES).from_buffer_copy(unique_id)
        _abi.check(self._lib.vrt_comm_init(self._ctx, int(world), int(rank), buf), "vrt_comm_init")

    def comm_expect_sizes(self, gather_tile_bytes: int, exchange_chunk_bytes: int) -> None:
        """Every rank (collective, synchronous): agree on the byte counts gather_tiles / exchange_tiles are going to be called with;
        a rank that disagrees makes the call fail on every rank instead of hanging the collective later."""
        self._require()
        _abi.check(self._lib.vrt_comm_expect_sizes(self._ctx, int(gather_tile_bytes), int(exchange_chunk_bytes)), "vrt_comm_expect_sizes")

    def gather_tiles(self, tile_ptr: int, frame_ptr: int, tile_bytes: int, root: int = 0, stream: int = 0) -> None:
        """Asynchronous ncclGather of this rank's device tile into rank `root`'s device frame (rank-major)."""
        self._require()
        _abi.check(self._lib.vrt_gather_tiles(self._ctx, C.c_void_p(tile_ptr), C.c_void_p(frame_ptr) if frame_ptr else None, int(tile_bytes),
                                              int(root), C.c_void_p(stream)), "vrt_gather_tiles")

    def exchange_tiles(self, tiles_ptr: int, recv_ptr: int, chunk_bytes: int, stream: int = 0) -> None:
        """Asynchronous all-to-all of equal chunks (vrt_exchange_tiles): chunk d of `tiles_ptr` goes to rank d, chunk s of
        `recv_ptr` comes from rank s."""
        self._require()
        _abi.check(self._lib.vrt_exchange_tiles(self._ctx, C.c_void_p(tiles_ptr), C.c_void_p(recv_ptr), int(chunk_bytes), C.c_void_p(stream)),
                   "vrt_exchange_tiles")

    def last_timing(self) -> dict:
        self._require()
        t = _abi.vrt_timing()
        _abi.check(self._lib.vrt_last_timing(self._ctx, C.byref(t)), "vrt_last_timing")
        return {name: getattr(t, name) for name, _ in t._fields_}

    def timing_history(self, n: int) -> List[float]:
        self._require()
        buf = (C.c_float * max(n, 1))()
        m = self._lib.vrt_timing_history(self._ctx, n, buf)
        if m < 0:
            _abi.check(m, "vrt_timing_history")
        return [buf[i] for i in range(m)]

    def launch_history(self, n: int):
        """(kernel ms, frames covered) of the last n march launches, oldest first (vrt_launch_history); 0 ms = not event-timed."""
        self._require()
        ms = (C.c_float * max(n, 1))()
        fr = (C.c_int * max(n, 1))()
        m = self._lib.vrt_launch_history(self._ctx, n, ms, fr)
        if m < 0:
            _abi.check(m, "vrt_launch_history")
        return [(ms[i], fr[i]) for i in range(m)]

    def gather_ceiling(self, fmt: int = _abi.FORMAT_F32, coherent_lanes: bool = False, n_bricks: int = 32) -> float:
        """G trilinear samples per second the chip sustains for the march's inner operation in isolation (vrt_debug_gather_ceiling)."""
        self._require()
        out = C.c_float()
        _abi.check(self._lib.vrt_debug_gather_ceiling(self._ctx, int(fmt), 1 if coherent_lanes else 0, int(n_bricks), C.byref(out)), "vrt_debug_gather_ceiling")
        return float(out.value)

    def last_kernel_form(self) -> int:
        """Bit set of _abi.FORM_* naming the closest-hit kernel form the last march launch ran (vrt_debug_last_kernel_form)."""
        self._require()
        f = self._lib.vrt_debug_last_kernel_form(self._ctx)
        if f < 0:
            _abi.check(f, "vrt_debug_last_kernel_form")
        return int(f)

    def wave_records(self, which: int = 0) -> np.ndarray:
        """Per-wave records of the last launch, [waves, 8] uint32 (vrt_debug_wave_records):
        which=0 counters, which=1 diagnostic timeline (after a FLAG_DIAG_TIMELINE launch)."""
        self._require()
        n = self._lib.vrt_debug_wave_records(self._ctx, which, None, 0)
        if n < 0:
            _abi.check(int(n), "vrt_debug_wave_records")
        buf = np.zeros(int(n), dtype=np.uint32)
        self._lib.vrt_debug_wave_records(self._ctx, which, buf.ctypes.data_as(C.c_void_p), int(n))
        return buf.reshape(-1, 8)

    def _require(self) -> None:
        if not self._ctx:
            raise RuntimeError("renderer is not active (Start() not called or failed)")

    def __enter__(self):
        if not self.Start():
            raise RuntimeError("VHipRenderer.Start() failed — no HIP device or library error")
        return self

    def __exit__(self, *exc):
        self.Stop()
        return False


def algorithmic_bytes(t: dict, bytes_per_pixel: int = 16) -> int:
    """32 B per trilinear sample, 6 samples per hit normal, one framebuffer store per pixel
    rendered (= primary ray)."""
    samples = t["primary_steps"] + t["shadow_steps"]
    return 32 * samples + 32 * 6 * t["hits"] + bytes_per_pixel * t["primary_rays"]

"""ctypes mirror of include/vrt.h and the loader of the in-tree C-ABI library.

The library (volumetricraytracer_amd/lib/libvrt_hip.so) is the product; this module only
declares its entry points.  There is no fallback: if the library is missing or fails to load,
`load()` raises — nothing here can render on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

VRT_MAX_VOLUMES = 20
VRT_MAX_TEXTURES = 64
VRT_FRAMES_IN_FLIGHT = 3
VRT_MAX_RESOLUTION = 9
VRT_MAX_POINT_LIGHTS = 5
VRT_MAX_SPOT_LIGHTS = 5
VRT_MAX_INSTANCES = 64
VRT_MAX_DEVICES = 8
VRT_COMM_ID_BYTES = 128

VRT_OK = 0
VRT_ERR_INVALID = -1
VRT_ERR_NO_DEVICE = -2
VRT_ERR_HIP = -3
VRT_ERR_OOM = -4
VRT_ERR_SLOT = -5
VRT_ERR_NOT_READY = -6
VRT_ERR_UNSUPPORTED = -7

# EVRenderMode (reference Renderer/Public/Renderer.h:32-42)
MODE_INTERP = 0
MODE_INTERP_UNLIT = 1
MODE_INTERP_NOTEX = 2
MODE_INTERP_NOTEX_UNLIT = 3
MODE_CUBE = 4
MODE_CUBE_UNLIT = 5
MODE_CUBE_NOTEX = 6
MODE_CUBE_NOTEX_UNLIT = 7

FORM_FULL, FORM_PASSES, FORM_TEXTURED, FORM_MAY_BOUNCE, FORM_LEAN_REF = 1, 2, 4, 8, 16  # vrt_debug_last_kernel_form
FLAG_DIAG_TIMELINE = 4
FLAG_OUTPUT_RGBA8 = 8
FLAG_NO_TIMING = 16
FLAG_BLOCK_PER_FRAME = 64
FLAG_NO_CULL_RECT = 128
FLAG_FULL_ONE_KERNEL = 256   # full closest hit of a block of frames as ONE kernel (default: three passes)
FLAG_FULL_THREE_PASS = 512   # ... and three passes even for a lone frame
FLAG_OUTPUT_BGRA8 = 2048       # with FLAG_OUTPUT_RGBA8: B8G8R8A8 byte order, the reference's back buffer (DXConstants.cpp:21)
FLAG_REFERENCE_VIEW_VECTOR = 4096      # shade the camera ray's hit with the reference's un-normalised wo, back its secondary rays off 0.1 |dir|
FLAG_REFERENCE_BOUNDARY_TEXELS = 8192  # normal taps beyond the grid read texel 0 (the reference's out-of-bounds Load) instead of the clamped cell
FLAG_NO_HIT_POLISH = 1024     # closest hits stay at the cone threshold's stop point (rounds 1-3) instead of moving on to the crossing
HIT_POLISH_SAMPLES = 2
MAX_BLOCK_FRAMES = 48  # frames one march launch covers with their cameras in the kernarg segment (csrc/vrt_device.h kMaxBlockFrames)
MAX_LAUNCH_FRAMES = 256  # ... with their cameras copied to device memory ahead of the launch: vrt_block.n_frames' upper bound

FORMAT_F32 = 0
FORMAT_TEXEL16 = 1

PATH_AUTO = 0
PATH_DENSE = 1
PATH_BRICK = 2
PATH_BRICK_LDS = 3
PATH_CELLS = 4


class vrt_voxel(C.Structure):
    _fields_ = [("material", C.c_uint8), ("pad_", C.c_uint8 * 3), ("density", C.c_float)]


class vrt_material(C.Structure):
    _fields_ = [("tint", C.c_float * 4), ("roughness", C.c_float), ("metallic", C.c_float)]


class vrt_instance(C.Structure):
    _fields_ = [
        ("volume_slot", C.c_int32),
        ("position", C.c_float * 3),
        ("rotation", C.c_float * 4),
        ("scale", C.c_float * 3),
    ]


class vrt_point_light(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("color", C.c_float * 3),
        ("intensity", C.c_float),
        ("att_linear", C.c_float),
        ("att_exp", C.c_float),
    ]


class vrt_spot_light(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("forward", C.c_float * 3),
        ("color", C.c_float * 3),
        ("intensity", C.c_float),
        ("att_linear", C.c_float),
        ("att_exp", C.c_float),
        ("cos_angle", C.c_float),
        ("cos_falloff_angle", C.c_float),
    ]


class vrt_scene(C.Structure):
    _fields_ = [
        ("cam_position", C.c_float * 3),
        ("cam_rotation", C.c_float * 4),
        ("cam_fov_deg", C.c_float),
        ("cam_near", C.c_float),
        ("cam_far", C.c_float),
        ("light_dir", C.c_float * 3),
        ("light_strength", C.c_float),
        ("n_instances", C.c_int32),
        ("n_point_lights", C.c_int32),
        ("n_spot_lights", C.c_int32),
        ("pad_", C.c_int32),
        ("instances", vrt_instance * VRT_MAX_INSTANCES),
        ("point_lights", vrt_point_light * VRT_MAX_POINT_LIGHTS),
        ("spot_lights", vrt_spot_light * VRT_MAX_SPOT_LIGHTS),
    ]


class vrt_params(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("max_steps", C.c_int32),
        ("shadow", C.c_int32),
        ("mode", C.c_int32),
        ("path", C.c_int32),
        ("max_bounces", C.c_int32),
        ("flags", C.c_int32),
        ("eps_hit", C.c_float),
        ("eps_in", C.c_float),
        ("step_min", C.c_float),
        ("k_relax", C.c_float),
        ("cone_eps", C.c_float),
    ]


class vrt_timing(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_float),
        ("gather_ms", C.c_float),
        ("total_ms", C.c_float),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("primary_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("bounce_rays", C.c_uint64),
        ("primary_steps", C.c_uint64),
        ("shadow_steps", C.c_uint64),
        ("hits", C.c_uint64),
        ("exhausted_rays", C.c_uint64),
    ]


# name -> (restype, argtypes); every symbol include/vrt.h declares
class vrt_camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotation", C.c_float * 4), ("fov_deg", C.c_float)]


class vrt_block(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32),
        ("strip_rows", C.c_int32),
        ("first_strip", C.c_int32),
        ("strip_stride", C.c_int32),
        ("n_strips", C.c_int32),
        ("row0", C.c_int32),
        ("rows", C.c_int32),
        ("cameras", C.POINTER(vrt_camera)),
        ("frame_stride_bytes", C.c_uint64),
        ("scenes", C.POINTER(vrt_scene)),
    ]


SYMBOLS = {
    "vrt_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]),
    "vrt_destroy": (C.c_int, [C.c_void_p]),
    "vrt_volume_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_uint8, C.c_float, C.c_void_p, C.c_void_p]),
    "vrt_volume_upload_voxels": (C.c_int, [C.c_void_p, C.c_int, C.c_uint8, C.c_float, C.c_void_p]),
    "vrt_set_volume_format": (C.c_int, [C.c_void_p, C.c_int]),
    "vrt_volume_upload_texels": (C.c_int, [C.c_void_p, C.c_int, C.c_uint8, C.c_float, C.c_void_p]),
    "vrt_volume_set_material": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(vrt_material)]),
    "vrt_volume_set_metric": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float]),
    "vrt_texture_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vrt_texture_free": (C.c_int, [C.c_void_p, C.c_int]),
    "vrt_volume_set_textures": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]),
    "vrt_voxelize_mesh": (C.c_int, [C.c_void_p, C.c_int, C.c_uint8, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_size_t)]),
    "vrt_volume_download": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vrt_volume_free": (C.c_int, [C.c_void_p, C.c_int]),
    "vrt_env_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "vrt_scene_set": (C.c_int, [C.c_void_p, C.POINTER(vrt_scene)]),
    "vrt_render": (C.c_int, [C.c_void_p, C.POINTER(vrt_params), C.c_void_p]),
    "vrt_render_rows": (C.c_int, [C.c_void_p, C.POINTER(vrt_params), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vrt_render_strips": (C.c_int, [C.c_void_p, C.POINTER(vrt_params), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vrt_render_block": (C.c_int, [C.c_void_p, C.POINTER(vrt_params), C.POINTER(vrt_block), C.c_void_p, C.c_void_p]),
    "vrt_render_block_host": (C.c_int, [C.c_void_p, C.POINTER(vrt_params), C.POINTER(vrt_block), C.POINTER(C.c_void_p)]),
    "vrt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "vrt_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vrt_comm_destroy": (C.c_int, [C.c_void_p]),
    "vrt_comm_expect_sizes": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t]),
    "vrt_gather_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "vrt_exchange_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vrt_render_begin": (C.c_int, [C.c_void_p, C.POINTER(vrt_params), C.c_int]),
    "vrt_render_end": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "vrt_last_timing": (C.c_int, [C.c_void_p, C.POINTER(vrt_timing)]),
    "vrt_timing_history": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "vrt_launch_history": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "vrt_debug_wave_records": (C.c_longlong, [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]),
    "vrt_debug_last_kernel_form": (C.c_int, [C.c_void_p]),
    "vrt_debug_gather_ceiling": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_float)]),
    "vrt_strerror": (C.c_char_p, [C.c_int]),
    "vrt_version": (C.c_char_p, []),
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvrt_hip.so")

_lib = None


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
    libhsa-runtime64; if ours (from /opt/rocm) is loaded first and torch's second, the second
    ROCr instance finds no GPU.  Both have the soname libamdhip64.so.7, so importing torch
    first makes the dynamic loader resolve our DT_NEEDED entry to the copy torch already
    mapped.  Without torch installed this is a no-op and the system ROCm runtime is used."""
    import sys

    if "torch" in sys.modules or os.environ.get("VRT_NO_TORCH_PRELOAD") == "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def load(path: str | None = None) -> C.CDLL:
    """Load libvrt_hip.so and bind every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("VRT_LIB") or LIB_PATH  # VRT_LIB: an A/B build of the same library (tools/ab_lib_variants.sh)
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} is missing: the HIP library is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or volumetricraytracer_amd/csrc/build.sh). "
            "There is no CPU fallback."
        )
    _share_hip_runtime_with_torch()
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


class VrtError(RuntimeError):
    def __init__(self, status: int, what: str):
        self.status = status
        try:
            msg = load().vrt_strerror(status).decode()
        except Exception:  # pragma: no cover - only if the library vanished
            msg = "?"
        super().__init__(f"{what}: {msg} ({status})")


def check(status: int, what: str) -> None:
    if status != VRT_OK:
        raise VrtError(status, what)

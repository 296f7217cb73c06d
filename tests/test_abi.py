"""C-ABI checks that need no GPU: the library is built in-tree, loads, exports every symbol
include/vrt.h declares, and the ctypes mirror has the C layout.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

import volumetricraytracer_amd as v
from volumetricraytracer_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vrt.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vrt_[a-z_]+)\s*\(", src)))


def test_library_is_built_and_loads():
    assert os.path.exists(_abi.LIB_PATH), "run __graft_entry__.build() first"
    lib = _abi.load()
    assert lib.vrt_version().decode().endswith("gfx950")
    assert lib.vrt_strerror(0) == b"ok"
    assert b"slot" in lib.vrt_strerror(_abi.VRT_ERR_SLOT)


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 15
    lib = C.CDLL(_abi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vrt.h but not exported"
        assert n in _abi.SYMBOLS, f"{n} has no ctypes binding"
    assert sorted(_abi.SYMBOLS) == names
    # the device code object for gfx950 is embedded in the library
    blob = open(_abi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"march_kernel" in blob


def test_ctypes_layout_matches_c(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "vrt.h"\n'
        "int main(void){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(vrt_voxel), sizeof(vrt_material),"
        " sizeof(vrt_instance), sizeof(vrt_point_light), sizeof(vrt_spot_light), sizeof(vrt_scene), sizeof(vrt_params),"
        " sizeof(vrt_timing), offsetof(vrt_scene, instances), offsetof(vrt_scene, point_lights), offsetof(vrt_timing, primary_rays), sizeof(vrt_camera), sizeof(vrt_block),"
        " offsetof(vrt_block, cameras), offsetof(vrt_block, frame_stride_bytes), offsetof(vrt_block, scenes));return 0;}\n"
    )
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(_abi.vrt_voxel), C.sizeof(_abi.vrt_material), C.sizeof(_abi.vrt_instance),
            C.sizeof(_abi.vrt_point_light), C.sizeof(_abi.vrt_spot_light), C.sizeof(_abi.vrt_scene),
            C.sizeof(_abi.vrt_params), C.sizeof(_abi.vrt_timing), _abi.vrt_scene.instances.offset,
            _abi.vrt_scene.point_lights.offset, _abi.vrt_timing.primary_rays.offset, C.sizeof(_abi.vrt_camera),
            C.sizeof(_abi.vrt_block), _abi.vrt_block.cameras.offset, _abi.vrt_block.frame_stride_bytes.offset, _abi.vrt_block.scenes.offset]
    assert got == want
    assert C.sizeof(_abi.vrt_voxel) == 8  # VVoxel, Voxel.h:23-30


def test_argument_errors_without_touching_the_gpu():
    lib = _abi.load()
    assert lib.vrt_create(None, 1, None) == _abi.VRT_ERR_INVALID
    ctx = C.c_void_p()
    assert lib.vrt_create(C.byref(ctx), 0, None) == _abi.VRT_ERR_INVALID
    assert lib.vrt_create(C.byref(ctx), 9, None) == _abi.VRT_ERR_INVALID
    assert lib.vrt_destroy(None) == _abi.VRT_ERR_INVALID
    assert lib.vrt_render(None, None, None) == _abi.VRT_ERR_INVALID
    assert lib.vrt_last_timing(None, None) == _abi.VRT_ERR_INVALID
    assert lib.vrt_render_block(None, None, None, None, None) == _abi.VRT_ERR_INVALID
    assert lib.vrt_render_rows(None, None, 0, 0, None, None) == _abi.VRT_ERR_INVALID
    assert lib.vrt_gather_tiles(None, None, None, 0, 0, None) == _abi.VRT_ERR_NOT_READY  # no context, no communicator


def test_no_fallback_when_library_missing(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _abi.load(str(tmp_path / "nope.so"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "volumetricraytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f in ("vrt_kernels.hip",), f"{f} mentions the oracle"
    # the kernel file only *mentions* the oracle in its header comment; it must not include it
    k = open(os.path.join(pkg, "csrc", "vrt_kernels.hip")).read()
    assert "#include \"../../oracle" not in k and "vrt_oracle.h" not in k


def test_scene_packing_limits():
    vol = v.sphere_volume(3, 10.0, 4.0)
    sc = v.VScene(Objects=[v.VVoxelObject(Volume=vol) for _ in range(65)])
    with pytest.raises(ValueError):
        sc.to_abi()
    sc = v.VScene(Objects=[v.VVoxelObject(Volume=v.sphere_volume(1, 10.0, 4.0)) for _ in range(21)])
    with pytest.raises(ValueError):
        sc.to_abi()

"""The C++ adaptor's headers against the reference's own interface declarations.

The drop-in boundary on the C++ side is `VRenderer` (Renderer/Public/Renderer.h:44-60) and `VTextureFactory`
(Renderer/Public/TextureFactory.h:32-41): a class that overrides this build's csrc/host/HostRenderer.h must override the reference's
header too, so the virtuals' names AND parameter types have to be the same.  The reference's headers are parsed as TEXT (no compile:
they pull in boost / Eigen); when /root/reference is absent (the GPU box) the comparison is skipped and only this build's side is
checked through the library (vrh_texture_factory_probe)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/VolumetricRaytracer/VolumetricRaytracer/Renderer/Public"
HOST = os.path.join(ROOT, "volumetricraytracer_amd", "csrc", "host")


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _class_body(text, name):
    m = re.search(r"class\s+" + name + r"\b[^;{]*\{", text)
    assert m, name
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1]


def _members(body):
    """[(name, (parameter types...), kind 'virtual' | 'static' | '', is_override, is_pure)] of the member functions declared in a class body."""
    out = []
    for m in re.finditer(r"([\w:<>\s\*&]+?)\s+(\w+)\s*\(([^()]*)\)\s*(const)?\s*(override)?\s*(=\s*0)?\s*(;|\{)", body):
        ret, name, params = m.group(1).split(), m.group(2), m.group(3)
        kind = "virtual" if "virtual" in ret else "static" if "static" in ret else ""
        if name in ("VRenderer", "if", "for", "while", "return"):
            continue
        types = []
        for prm in (x.strip() for x in params.split(",") if x.strip()):
            prm = re.sub(r"\s+", " ", prm)
            prm = re.sub(r"\s*\b\w+$", "", prm) if re.search(r"[\w>&\*]\s+\w+$", prm) else prm  # drop the parameter's name
            types.append(prm.replace(" &", "&").replace(" *", "*"))
        out.append((name, tuple(types), kind, m.group(5) is not None, m.group(6) is not None))
    return out


def _signatures(body):
    return {(n, t, k) for n, t, k, _, _ in _members(body)}


def _read(path):
    with open(path, encoding="latin-1") as f:
        return _strip_comments(f.read())


needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is not present on this machine")


@needs_reference
def test_vrenderer_virtuals_are_the_reference_s():
    ref = _signatures(_class_body(_read(os.path.join(REF, "Renderer.h")), "VRenderer"))
    own = _signatures(_class_body(_read(os.path.join(HOST, "HostRenderer.h")), "VRenderer"))
    ref_virtual = {(n, t) for n, t, k in ref if k == "virtual"}
    own_virtual = {(n, t) for n, t, k in own if k == "virtual"}
    assert ("InitializeTexture", ("VObjectPtr<VTexture>",)) in ref_virtual and ("UploadToGPU", ("VObjectPtr<VTexture>",)) in ref_virtual
    assert ref_virtual == own_virtual, (sorted(ref_virtual ^ own_virtual))
    assert {(n, t) for n, t, k in ref if k == ""} >= {("SetRendererMode", ("const EVRenderMode&",))}
    assert ("SetRendererMode", ("const EVRenderMode&",), "") in own
    # the render modes carry the reference's numeric values
    modes = lambda text: re.findall(r"(\w+)\s*=\s*(\d+)", re.search(r"enum class EVRenderMode\s*\{(.*?)\}", text, re.S).group(1))  # noqa: E731
    assert modes(_read(os.path.join(REF, "Renderer.h"))) == modes(_read(os.path.join(HOST, "HostRenderer.h")))


@needs_reference
def test_texture_factory_statics_are_the_reference_s():
    ref = _signatures(_class_body(_read(os.path.join(REF, "TextureFactory.h")), "VTextureFactory"))
    own = _signatures(_class_body(_read(os.path.join(HOST, "HostRenderer.h")), "VTextureFactory"))
    assert len(ref) == 5 and all(k == "static" for _, _, k in ref)
    assert ref == own, sorted(ref ^ own)
    assert ("LoadTextureCubeFromFile", ("std::weak_ptr<VRenderer>", "const std::wstring&"), "static") in own


@needs_reference
def test_hip_renderer_overrides_every_pure_virtual():
    """VHipRenderer must be instantiable against the reference's header: every pure virtual of VRenderer has an `override` with the
    same parameter types in HipRenderer.h."""
    pure = {(n, t) for n, t, k, _, is_pure in _members(_class_body(_read(os.path.join(REF, "Renderer.h")), "VRenderer")) if is_pure}
    assert len(pure) == 7 and ("ResizeRenderOutput", ("unsigned int", "unsigned int")) in pure
    overrides = {(n, t) for n, t, _, is_override, _ in _members(_class_body(_read(os.path.join(HOST, "HipRenderer.h")), "VHipRenderer")) if is_override}
    assert pure <= overrides, sorted(pure - overrides)


def test_texture_factory_and_base_typed_virtuals_work(tmp_path):
    """This build's side without a GPU: the five factory functions through a renderer that counts InitializeTexture calls, and
    VHipRenderer::InitializeTexture with a base-typed argument."""
    lib_path = os.environ.get("VRT_HOST_LIB") or os.path.join(ROOT, "volumetricraytracer_amd", "lib", "libvrt_host.so")
    if not os.path.exists(lib_path):
        import __graft_entry__

        __graft_entry__.build()
    from volumetricraytracer_amd import _abi

    _abi.load()  # maps the HIP runtime and libvrt_hip.so first (libvrt_host.so's DT_NEEDED)
    lib = C.CDLL(lib_path)
    lib.vrh_texture_factory_probe.restype = C.c_int
    lib.vrh_texture_factory_probe.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    lib.vrh_last_error.restype = C.c_char_p
    # a 5x3 binary PPM and a folder of six 4x4 faces
    img = tmp_path / "t.ppm"
    img.write_bytes(b"P6\n5 3\n255\n" + bytes(range(45)))
    faces = tmp_path / "sky"
    faces.mkdir()
    from volumetricraytracer_amd import vox_io  # noqa: F401  (keeps the package import honest)
    import zlib, struct

    def png(path, w, h):
        raw = b"".join(b"\x00" + bytes([x * 9 % 256, y * 17 % 256, 80, 255] * 1)[0:4] * w for y in range(h) for x in [0])
        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
        path.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))

    for f in ("XP", "XM", "YP", "YM", "ZP", "ZM"):
        png(faces / (f + ".png"), 4, 4)
    dims = (C.c_int * 8)()
    rc = lib.vrh_texture_factory_probe(str(img).encode(), str(faces).encode(), dims)
    assert rc == 0, lib.vrh_last_error()
    assert list(dims) == [5, 3, 4, 3, 2, 6, 7, 5]  # five factory calls, five InitializeTexture calls
    assert lib.vrh_texture_factory_probe(str(tmp_path / "missing.png").encode(), b"", dims) == -1

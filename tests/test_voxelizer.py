"""BASELINE config 1 (CPU plumbing): glTF unit cube -> 32^3 SDF voxel volume through the C++
Voxelizer restatement, the `.vox` scene format, and the C++ <-> Python reader/writer cross-check.

The reference ships no fixtures for this path: the expectations are analytic and follow from
VolumeConverter.cpp:32-33,51-57,200-202 and GLTFImporter.cpp:56-63."""
import math
import os
import struct
import subprocess

import numpy as np
import pytest

import volumetricraytracer_amd as v
from volumetricraytracer_amd import vox_io
from volumetricraytracer_amd import voxelizer as vx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VOXELIZER = os.path.join(ROOT, "volumetricraytracer_amd", "lib", "voxelizer")


@pytest.fixture(scope="module")
def cube_volume():
    pos, nrm, idx = vx.cube_mesh(0.5)
    p, be = vx.importer_space(pos)
    assert np.allclose(be, 55.0)  # half size 50 + 5
    return vx.convert_mesh(p, idx, be, "cube_5")


def test_cube_volume_geometry(cube_volume):
    vol = cube_volume
    # resolution from the "_5" suffix, N = 33, extent = 55 * 1.25, cell = 2*extent/32, thr = cell*sqrt(3)
    assert (vol.Resolution, vol.N) == (5, 33)
    assert vol.VolumeExtends == pytest.approx(68.75)
    assert vol.GetCellSize() == pytest.approx(4.296875)
    assert vol.density_scale == pytest.approx(4.296875 * math.sqrt(3.0), rel=1e-6)
    assert vol.step_max == pytest.approx(0.5 * vol.density_scale)


def test_cube_densities_match_point_triangle_distance(cube_volume):
    vol = cube_volume
    thr = np.float32(vol.density_scale)
    p = vol.axis_positions().astype(np.float64)
    X, Z, Y = np.meshgrid(p, p, p, indexing="ij")  # density axes are (x, z, y)
    # exact unsigned distance to the surface of the cube [-50,50]^3
    q = np.stack([np.abs(X) - 50.0, np.abs(Y) - 50.0, np.abs(Z) - 50.0], -1)
    outside = np.linalg.norm(np.maximum(q, 0.0), axis=-1)
    inside = np.minimum(q.max(-1), 0.0)
    dist = np.abs(outside + inside)
    expect = dist / thr - 0.5
    # a voxel closer than thr to the mesh holds exactly dist/thr - 0.5 (it lies inside the index box of
    # its nearest triangle); the shell is unsigned: the same value either side of a face
    near = dist < thr * 0.999
    assert near.sum() > 3000
    assert np.abs(vol.density[near] - expect[near]).max() < 2e-5
    assert (vol.density[near] < 0.5).all()
    # everything else is either another triangle's (larger) distance or the untouched background 2*extent
    far = ~near
    assert (vol.density[far] >= 0.5 - 1e-5).all()
    corner = vol.density[0, 0, 0]
    assert corner == pytest.approx(2 * 68.75) or corner >= 0.5
    assert (vol.density == np.float32(137.5)).sum() > 0  # voxels outside every triangle's box
    # material flag = density <= 0 (VolumeConverter.cpp:244)
    touched = vol.density < 137.0
    assert np.array_equal(vol.material_id[touched] == 1, vol.density[touched] <= 0)


def test_resolution_suffix_rules():
    pos, nrm, idx = vx.cube_mesh(0.5)
    p, be = vx.importer_space(pos)
    assert vx.convert_mesh(p, idx, be, "cube_3").N == 9
    assert vx.convert_mesh(p, idx, be, "cube").N == 33            # no suffix -> 5
    assert vx.convert_mesh(p, idx, be, "cube_x").N == 33          # not a number -> 5
    assert vx.convert_mesh(p, idx, be, "cube_9").N == 33          # > 8 -> 5
    assert vx.convert_mesh(p, idx, be, "a_b_4").N == 17           # last underscore wins


def test_degenerate_and_ragged_input_is_tolerated():
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 2, 2]], np.float32)
    idx = np.array([0, 1, 2, 3, 3, 3, 0, 1], np.uint32)  # one good triangle, one zero-area, two dangling indices
    p, be = vx.importer_space(pos)
    vol = vx.convert_mesh(p, idx, be, "tri_4")
    assert (vol.density < 0).sum() > 0 and np.isfinite(vol.density).all()
    empty = vx.convert_mesh(p, np.zeros(0, np.uint32), be, "none_4")
    assert (empty.density == empty.density[0, 0, 0]).all()


def test_gltf_to_vox_cli_and_roundtrip(tmp_path):
    pos, nrm, idx = vx.cube_mesh(0.5)
    gltf = str(tmp_path / "scene.gltf")
    yaw = v.quat_from_axis_angle(v.UP, math.radians(30.0))
    nodes = [
        {"name": "Cube", "mesh": 0, "translation": [1.0, 2.0, 3.0], "rotation": [float(x) for x in yaw], "scale": [1.0, 2.0, 0.5]},
        {"name": "Light_Sun", "rotation": [0.0, 0.0, 0.0, 1.0], "extras": {"strength": 6.0, "color_r": 1.0, "color_g": 0.9, "color_b": 0.8}},
        {"name": "Light_Point", "translation": [0.5, 0.0, 1.0], "extras": {"strength": 40.0, "attl": 0.25, "attexp": 0.01}},
        {"name": "Light_Spot", "translation": [0.0, 1.0, 1.0], "extras": {"strength": 30.0, "fangle": 15.0, "angle": 50.0}},
        {"name": "Empty"},
    ]
    mats = [{"name": "paint", "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.6, 1.0], "metallicFactor": 0.3, "roughnessFactor": 0.7}}]
    vx.write_gltf(gltf, [("cube_5", pos, nrm, idx, 0)], nodes, mats)
    assert os.path.exists(VOXELIZER), "run __graft_entry__.build()"
    r = subprocess.run([VOXELIZER, gltf], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = str(tmp_path / "scene.vox")
    assert os.path.exists(out) and "Exported voxelized scene" in r.stdout

    sc = vox_io.load_scene(out)  # Python reader on the C++ writer's file
    assert len(sc.volumes()) == 1 and len(sc.Objects) == 1
    vol = sc.volumes()[0]
    assert (vol.Resolution, vol.N) == (5, 33) and vol.VolumeExtends == pytest.approx(68.75)
    assert np.allclose(vol.Material.AlbedoColor, [0.2, 0.4, 0.6, 1.0]) and vol.Material.Roughness == pytest.approx(0.7)
    assert vol.Material.Metallic == pytest.approx(0.3)
    o = sc.Objects[0]
    assert np.allclose(o.Position, [100, 200, 300]) and np.allclose(o.Scale, [1, 2, 0.5]) and np.allclose(o.Rotation, yaw, atol=1e-6)
    assert sc.DirectionalLight.IlluminationStrength == 6.0 and np.allclose(sc.DirectionalLight.Color, [1, 0.9, 0.8, 1])
    assert len(sc.PointLights) == 1 and sc.PointLights[0].AttenuationLinear == 0.25 and np.allclose(sc.PointLights[0].Position, [50, 0, 100])
    assert len(sc.SpotLights) == 1 and sc.SpotLights[0].FalloffAngle == 15.0 and sc.SpotLights[0].Angle == 50.0
    # same volume as the in-memory conversion
    p, be = vx.importer_space(pos)
    ref = vx.convert_mesh(p, idx, be, "cube_5")
    assert np.array_equal(ref.density, vol.density) and np.array_equal(ref.material_id, vol.material_id)

    # C++ reader + writer reproduce the file bit for bit; so does the Python writer on the Python-read scene
    again = str(tmp_path / "again.vox")
    vx.vox_rewrite(out, again)
    assert open(out, "rb").read() == open(again, "rb").read()
    py = str(tmp_path / "py.vox")
    vox_io.save_scene(sc, py)
    back = str(tmp_path / "back.vox")
    vx.vox_rewrite(py, back)  # C++ reader on the Python writer's file
    sc2 = vox_io.load_scene(back)
    assert np.array_equal(sc2.volumes()[0].density, vol.density)
    assert np.allclose(sc2.Objects[0].Rotation, o.Rotation) and sc2.SpotLights[0].Angle == 50.0

    # embedded (data: URI) buffers give the same result
    gltf2 = str(tmp_path / "embedded.gltf")
    vx.write_gltf(gltf2, [("cube_5", pos, nrm, idx, 0)], nodes, mats, embed=True)
    out2 = vx.voxelize_file(gltf2)
    assert np.array_equal(vox_io.load_scene(out2).volumes()[0].density, vol.density)


def test_vox_reader_is_order_agnostic_and_rejects_garbage(tmp_path):
    # property order on disk is hash-map order in the reference: write a volume archive in reverse order
    vol = v.sphere_volume(2, 10.0, 4.0, v.VMaterial((0.1, 0.2, 0.3, 1.0), 0.5, 0.25))
    sc = v.VScene(Objects=[v.VVoxelObject(Position=(1, 2, 3), Volume=vol)])
    sorted_path = str(tmp_path / "sorted.vox")
    vox_io.save_scene(sc, sorted_path)
    root = vox_io.read_archive(sorted_path)
    path = str(tmp_path / "rev.vox")
    with open(path, "wb") as f:  # hand-rolled writer emitting properties in reverse-sorted order
        def w(ar):
            f.write(struct.pack("<Q", len(ar.buffer)) + ar.buffer + struct.pack("<Q", len(ar.props)))
            for k in sorted(ar.props, reverse=True):
                raw = k.encode() + b"\0"
                f.write(struct.pack("<Q", len(raw)) + raw)
                w(ar.props[k])
        w(root)
    assert open(path, "rb").read() != open(sorted_path, "rb").read()
    out = str(tmp_path / "norm.vox")
    vx.vox_rewrite(path, out)  # the C++ reader copes with any order, its writer sorts again
    got = vox_io.load_scene(out)
    assert np.array_equal(got.volumes()[0].density, vol.density) and got.volumes()[0].Material.Metallic == pytest.approx(0.25)
    assert np.allclose(got.Objects[0].Position, [1, 2, 3])
    assert np.array_equal(vox_io.load_scene(path).volumes()[0].density, vol.density)  # and so does the Python reader
    bad = str(tmp_path / "bad.vox")
    open(bad, "wb").write(b"\xff" * 64)
    with pytest.raises(RuntimeError):
        vx.vox_rewrite(bad, out)
    with pytest.raises(Exception):
        vox_io.load_scene(bad)
    with pytest.raises(RuntimeError):
        vx.voxelize_file(str(tmp_path / "missing.gltf"))


def test_png_reader_survives_damaged_files(tmp_path):
    """The material-texture PNG reader on 300 damaged copies of a valid file (random bytes, some with the chunk CRCs recomputed
    so that the damage reaches the inflater and the un-filter code, some truncated): a texture or an error, never a crash."""
    import zlib

    rng = np.random.RandomState(9)
    good = str(tmp_path / "g.png")
    _write_png(good, rng.randint(0, 256, (13, 17, 4)).astype(np.uint8), 6)
    raw = open(good, "rb").read()
    ok = refused = 0
    for k in range(300):
        b = bytearray(raw)
        for _ in range(int(rng.randint(1, 4))):
            b[int(rng.randint(8, len(b)))] = int(rng.randint(0, 256))
        if k % 4 == 1:  # valid CRCs around damaged chunk data
            out, pos = bytearray(b[:8]), 8
            while pos + 12 <= len(b):
                ln = struct.unpack(">I", b[pos:pos + 4])[0]
                if pos + 12 + ln > len(b):
                    break
                t, dd = bytes(b[pos + 4:pos + 8]), bytes(b[pos + 8:pos + 8 + ln])
                out += b[pos:pos + 8] + dd + struct.pack(">I", zlib.crc32(t + dd) & 0xffffffff)
                pos += 12 + ln
            b = out
        if k % 4 == 2:
            b = b[: int(rng.randint(0, len(b)))]
        path = str(tmp_path / "f.png")
        open(path, "wb").write(bytes(b))
        try:
            img = vx.load_texture(path)
            assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 4
            ok += 1
        except RuntimeError:
            refused += 1
    assert refused > 200 and ok + refused == 300


def test_gltf_importer_survives_damaged_files(tmp_path):
    """.gltf (JSON with an embedded buffer) and .glb (binary container) with 1-3 damaged bytes, some also truncated: the C++
    importer either voxelizes the file or refuses it with an error (whose text may hold bytes of the file: the Python wrapper
    must not choke on them) — 300 mutations, no crash, no hang."""
    pos, nrm, idx = vx.cube_mesh()
    node = [{"name": "cube_3", "mesh": 0}]
    emb = str(tmp_path / "cube.gltf")
    vx.write_gltf(emb, [("cube_3", pos, nrm, idx, None)], node, embed=True)
    ext = str(tmp_path / "cube2.gltf")
    vx.write_gltf(ext, [("cube_3", pos, nrm, idx, None)], node, embed=False)
    glb = str(tmp_path / "cube.glb")
    vx.gltf_to_glb(ext, glb)
    sources = [(open(emb, "rb").read(), ".gltf"), (open(glb, "rb").read(), ".glb")]
    rng = np.random.RandomState(5)
    ok = refused = 0
    for k in range(300):
        raw, suffix = sources[k % 2]
        b = bytearray(raw)
        for _ in range(int(rng.randint(1, 4))):
            i = int(rng.randint(0, len(b)))
            b[i] = int(rng.randint(0, 256)) if k % 3 == 0 else b[i] ^ (1 << int(rng.randint(0, 8)))
        if rng.randint(0, 5) == 0:
            b = b[: int(rng.randint(0, len(b)))]
        path = str(tmp_path / ("damaged" + suffix))
        open(path, "wb").write(bytes(b))
        try:
            vx.voxelize_file(path, str(tmp_path / "o.vox"))
            ok += 1
        except RuntimeError:
            refused += 1
    assert ok > 30 and refused > 100


def test_vox_readers_survive_truncated_and_bit_flipped_files(tmp_path):
    """A .vox file is length-prefixed all the way down (SerializationManager.cpp): a damaged length must not turn into a
    crash, a hang or a multi-gigabyte allocation.  Every prefix of a small scene file and 300 single-bit flips of it go through
    the C++ reader and the Python reader: each either loads (a flip in a payload byte) or is refused with an error."""
    vol = v.sphere_volume(2, 10.0, 4.0, v.VMaterial((0.1, 0.2, 0.3, 1.0), 0.5, 0.25))
    sc = v.VScene(Objects=[v.VVoxelObject(Position=(1, 2, 3), Volume=vol)], PointLights=[v.VPointLight(Position=(5, 5, 5))])
    good = str(tmp_path / "good.vox")
    vox_io.save_scene(sc, good)
    raw = open(good, "rb").read()
    out = str(tmp_path / "out.vox")
    bad = str(tmp_path / "bad.vox")
    rng = np.random.RandomState(3)
    cuts = sorted(set([0, 1, 7, 8, 9, 15, 16, 17, len(raw) - 1] + [int(x) for x in rng.randint(0, len(raw), 120)]))
    refused = 0
    for n in cuts:  # truncations: always an error
        open(bad, "wb").write(raw[:n])
        with pytest.raises(RuntimeError):
            vx.vox_rewrite(bad, out)
        with pytest.raises(Exception):
            vox_io.load_scene(bad)
        refused += 1
    loaded = 0
    for _ in range(300):  # bit flips: an error or a scene, never anything else
        b = bytearray(raw)
        i = int(rng.randint(0, len(b)))
        b[i] ^= 1 << int(rng.randint(0, 8))
        open(bad, "wb").write(bytes(b))
        try:
            vx.vox_rewrite(bad, out)
            loaded += 1
        except RuntimeError:
            pass
        try:
            vox_io.load_scene(bad)
        except Exception:  # noqa: BLE001
            pass
    assert refused == len(cuts) and loaded > 50  # most flips land in voxel payload and load fine


def test_relative_texture_paths_resolve_against_the_vox_folder(tmp_path, monkeypatch):
    """VMaterial::Deserialize(sourcePath, archive) (Core/Private/Material.cpp:72-100): a texture path that is not absolute
    is relative to the folder of the .vox file, whatever the working directory; absolute paths stay.  Both readers."""
    scene_dir = tmp_path / "assets" / "scene"
    scene_dir.mkdir(parents=True)
    elsewhere = tmp_path / "elsewhere"
    elsewhere.mkdir()
    absolute = str(tmp_path / "abs" / "rm.png")
    mat = v.VMaterial((0.5, 0.5, 0.5, 1.0), 0.4, 0.1)
    mat.AlbedoTexturePath, mat.NormalTexturePath, mat.RMTexturePath = "tex/albedo.png", "normal.png", absolute
    vol = v.sphere_volume(2, 10.0, 4.0, mat)
    path = str(scene_dir / "s.vox")
    vox_io.save_scene(v.VScene(Objects=[v.VVoxelObject(Volume=vol)]), path)
    monkeypatch.chdir(elsewhere)
    want = (str(scene_dir / "tex" / "albedo.png"), str(scene_dir / "normal.png"), absolute)
    m = vox_io.load_scene(path).volumes()[0].Material
    assert (m.AlbedoTexturePath, m.NormalTexturePath, m.RMTexturePath) == want
    out = str(elsewhere / "copy.vox")
    vx.vox_rewrite(path, out)  # C++ reader -> C++ writer: the resolved paths are what the scene now holds
    a = vox_io.read_archive(out)["V_0"]["Material"]
    got = tuple(vox_io._read_cstring(a[k]) for k in ("AlbedoTexture", "NormalTexture", "RMTexture"))
    assert got == want
    assert vox_io.volume_from_archive(vox_io.read_archive(path)["V_0"]).Material.AlbedoTexturePath == "tex/albedo.png"  # no source: as stored


def test_glb_container_gives_the_same_scene(tmp_path):
    """Binary glTF (.glb: JSON chunk + BIN chunk = buffer 0) through the same importer: the .vox is identical to the
    one from the .gltf + .bin pair."""
    pos, nrm, idx = vx.cube_mesh(0.5)
    gltf, glb = str(tmp_path / "scene.gltf"), str(tmp_path / "packed.glb")
    nodes = [{"name": "Cube", "mesh": 0, "translation": [1.0, 2.0, 3.0]}, {"name": "Light_Sun", "extras": {"strength": 6.0}}]
    vx.write_gltf(gltf, [("cube_4", pos, nrm, idx, None)], nodes)
    vx.gltf_to_glb(gltf, glb)
    out_a, out_b = str(tmp_path / "a.vox"), str(tmp_path / "b.vox")
    ra = subprocess.run([VOXELIZER, "--out", out_a, gltf], capture_output=True, text=True)
    rb = subprocess.run([VOXELIZER, "--out", out_b, glb], capture_output=True, text=True)
    assert ra.returncode == 0 and rb.returncode == 0, ra.stderr + rb.stderr
    assert open(out_a, "rb").read() == open(out_b, "rb").read()
    # a truncated container is an error, not a crash
    open(glb, "r+b").truncate(40)
    assert subprocess.run([VOXELIZER, "--out", out_b, glb], capture_output=True, text=True).returncode == 1


def _write_png(path, img, colour, filters=(0, 1, 2, 3, 4), palette=None, idat_split=3):
    """Minimal PNG writer (8 bit, non-interlaced) that applies the given row filters in turn — exercises every
    un-filter path of the reader."""
    import struct
    import zlib

    h, w = img.shape[:2]
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour]
    rows = img.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = rows[y]
        a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (a + prev) // 2
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        raw.append(ft)
        raw.extend(((cur - pred) & 0xff).astype(np.uint8).tobytes())
        prev = cur
    z = zlib.compress(bytes(raw), 6)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, colour, 0, 0, 0)))
        if palette is not None:
            f.write(chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes()))
        f.write(chunk(b"tEXt", b"Comment\0test"))
        step = max(1, len(z) // idat_split)
        for i in range(0, len(z), step):
            f.write(chunk(b"IDAT", z[i:i + step]))
        f.write(chunk(b"IEND", b""))


def test_jpeg_texture_loader_gives_the_bytes_libjpeg_gives(tmp_path):
    """VTexture2D::LoadJPEG (csrc/host/JpegDecoder.cpp: sequential Huffman JPEG, the IJG decoder's published arithmetic — accurate
    integer IDCT, triangle-filter chroma upsampling, fixed-point YCbCr -> RGB): the committed files of tests/golden (4:2:0 at an odd
    size, 4:2:2, 4:4:4 at quality 30, grey, restart markers, optimised Huffman tables, progressive with and without restart markers)
    decode to the bytes Pillow / libjpeg-turbo decoded them to (jpeg_expected.npz, written by tests/golden/make_jpeg_fixtures.py),
    byte for byte.  With Pillow at hand: a sweep of sizes down to 1x1, qualities, samplings, restart intervals, sequential and
    progressive, also byte for byte.  Damaged files are refused."""
    gold = os.path.join(os.path.dirname(__file__), "golden")
    exp = np.load(os.path.join(gold, "jpeg_expected.npz"))
    assert len(exp.files) == 8
    for name in exp.files:
        got = vx.load_texture(os.path.join(gold, f"jpeg_{name}.jpg"))
        assert got.shape == exp[name].shape[:2] + (4,) and (got[..., 3] == 255).all(), name
        assert np.array_equal(got[..., :3], exp[name]), name
    # damaged: cut short (the entropy-coded data ends early), and a marker segment that runs past the end
    raw = open(os.path.join(gold, "jpeg_420_odd.jpg"), "rb").read()
    bad = str(tmp_path / "bad.jpg")
    for data in (raw[:200], raw[:2] + b"\xff\xdb\xff\xff" + raw[6:], b"\xff\xd8\xff\xd9", raw[:len(raw) // 2].replace(b"\xff\xc0", b"\xff\xc2", 1)):
        with open(bad, "wb") as f:
            f.write(data)
        with pytest.raises(RuntimeError):
            vx.load_texture(bad)
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.RandomState(3)
    p = str(tmp_path / "t.jpg")
    n = 0
    for (w, h) in ((64, 48), (130, 77), (8, 8), (3, 5), (64, 40), (2, 2), (1, 1), (100, 10)):
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.clip(np.stack([127 + 120 * np.sin(xx / 9.0 + yy / 17.0), 127 + 120 * np.cos(xx / 5.0), 127 + 100 * np.sin(yy / 7.0)], -1)
                    + rng.randn(h, w, 3) * 12, 0, 255).astype(np.uint8)
        for q in (30, 95):
            for sub in (0, 1, 2):
                for extra in ({}, {"restart_marker_blocks": 3}, {"progressive": True}):
                    Image.fromarray(a, "RGB").save(p, quality=q, subsampling=sub, **extra)
                    got = vx.load_texture(p)
                    assert np.array_equal(got[..., :3], np.asarray(Image.open(p).convert("RGB"))), (w, h, q, sub, extra)
                    n += 1
    assert n == 144


def test_png_and_ppm_texture_loader(tmp_path):
    """The C++ host's material-texture decoder (VTexture2D::LoadFromFile: PNG via zlib, binary PPM): every colour type
    and every row filter, split IDAT chunks, ancillary chunks; rejects what it does not support."""
    rng = np.random.default_rng(11)
    h, w = 13, 9
    rgba = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    p = str(tmp_path / "t.png")
    _write_png(p, rgba, 6)
    assert np.array_equal(vx.load_texture(p), rgba)
    _write_png(p, rgba[..., :3], 2, filters=(4, 3, 1))
    got = vx.load_texture(p)
    assert np.array_equal(got[..., :3], rgba[..., :3]) and (got[..., 3] == 255).all()
    _write_png(p, rgba[..., 0], 0, filters=(2,))
    got = vx.load_texture(p)
    assert all(np.array_equal(got[..., c], rgba[..., 0]) for c in range(3)) and (got[..., 3] == 255).all()
    _write_png(p, rgba[..., :2], 4)
    got = vx.load_texture(p)
    assert np.array_equal(got[..., 0], rgba[..., 0]) and np.array_equal(got[..., 3], rgba[..., 1])
    pal = rng.integers(0, 256, size=(7, 3), dtype=np.uint8)
    idx = rng.integers(0, 7, size=(h, w), dtype=np.uint8)
    _write_png(p, idx, 3, palette=pal)
    got = vx.load_texture(p)
    assert np.array_equal(got[..., :3], pal[idx]) and (got[..., 3] == 255).all()
    try:  # an encoder that is not this test's own
        from PIL import Image
        Image.fromarray(rgba, "RGBA").save(p, optimize=True)
        assert np.array_equal(vx.load_texture(p), rgba)
    except ImportError:
        pass
    ppm = str(tmp_path / "t.ppm")
    with open(ppm, "wb") as f:
        f.write(b"P6\n# c\n%d %d\n255\n" % (w, h) + rgba[..., :3].tobytes())
    got = vx.load_texture(ppm)
    assert np.array_equal(got[..., :3], rgba[..., :3])
    # unsupported / broken files are refused
    _write_png(p, idx, 3, palette=pal[:2])  # palette index out of range
    with pytest.raises(RuntimeError):
        vx.load_texture(p)
    open(p, "wb").write(open(ppm, "rb").read()[:20])
    with pytest.raises(RuntimeError):
        vx.load_texture(p)
    with pytest.raises(RuntimeError):
        vx.load_texture(str(tmp_path / "missing.png"))


def _point_triangle_distance(P, A, B, C):
    """Closest-point-on-triangle distance (Ericson, Real-Time Collision Detection §5.1.5), float64, vectorised over points —
    an algorithm that shares nothing with the Voxelizer's 7-region classification."""
    ab, ac, ap = B - A, C - A, P - A
    d1, d2 = ap @ ab, ap @ ac
    bp = P - B
    d3, d4 = bp @ ab, bp @ ac
    cp = P - C
    d5, d6 = cp @ ab, cp @ ac
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    out = np.empty(len(P))
    with np.errstate(divide="ignore", invalid="ignore"):
        vq = d1 / (d1 - d3)
        wq = d2 / (d2 - d6)
        wr = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        den = 1.0 / (va + vb + vc)
    closest = A + np.outer(vb * den, ab) + np.outer(vc * den, ac)  # interior
    m = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)
    closest[m] = B + np.outer(wr, C - B)[m]
    m = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
    closest[m] = A + np.outer(wq, ac)[m]
    m = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
    closest[m] = A + np.outer(vq, ab)[m]
    closest[(d6 >= 0) & (d5 <= d6)] = C
    closest[(d3 >= 0) & (d4 <= d3)] = B
    closest[(d1 <= 0) & (d2 <= 0)] = A
    out[:] = np.linalg.norm(P - closest, axis=1)
    return out


def test_general_mesh_densities_match_brute_force_distance():
    """A torus mesh (slanted, thin and obtuse triangles at every orientation): every voxel nearer than thr to the mesh
    holds exactly (distance to the nearest triangle)/thr - 0.5, the distance taken from an independent closest-point
    algorithm over ALL triangles."""
    pos, nrm, idx = vx.torus_mesh(0.55, 0.22, 24, 12)
    p, be = vx.importer_space(pos)
    vol = vx.convert_mesh(p, idx, be, "torus_4")
    thr = float(vol.density_scale)
    g = vol.axis_positions().astype(np.float64)
    X, Z, Y = np.meshgrid(g, g, g, indexing="ij")
    P = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    tri = p[idx.reshape(-1, 3)].astype(np.float64)
    best = np.full(len(P), np.inf)
    for A, B, C in tri:
        best = np.minimum(best, _point_triangle_distance(P, A, B, C))
    best = best.reshape(X.shape)
    near = best < thr * 0.999
    assert near.sum() > 800
    assert np.abs(vol.density[near] - (best[near] / thr - 0.5)).max() < 5e-5
    assert (vol.density[~near] >= 0.5 - 1e-5).all()
    assert np.array_equal(vol.material_id == 1, vol.density <= 0)


def test_skybox_from_face_images(tmp_path):
    """Six face PNGs named like the reference's Resources/Skybox folder -> one cube map in D3D face order; the miss shader
    then reads the right face (oracle env lookup, dir.xzy)."""
    from oracle.binding import env_lookup

    S = 8
    cols = {"XP": (250, 10, 10), "XM": (10, 250, 10), "YP": (10, 10, 250), "YM": (250, 250, 10), "ZP": (250, 10, 250), "ZM": (10, 250, 250)}
    for name, c in cols.items():
        img = np.zeros((S, S, 4), np.uint8)
        img[..., :3] = c
        img[..., 3] = 255
        img[0, 0, :3] = (1, 2, 3)  # a marker texel: orientation is kept as stored
        _write_png(str(tmp_path / f"{name}.png"), img, 6)
    cube = vx.load_skybox_faces(str(tmp_path))
    assert cube.shape == (6, S, S, 4)
    for f, name in enumerate(("XP", "XM", "YP", "YM", "ZP", "ZM")):
        assert tuple(cube[f, 3, 3, :3]) == cols[name] and tuple(cube[f, 0, 0, :3]) == (1, 2, 3)
    # world +X looks at face +X; world +Z (up) is cube +Y after the .xzy swizzle of the miss shader
    assert np.allclose(env_lookup(cube, (1.0, 0.05, 0.02)) * 255.0, cols["XP"], atol=0.51)
    assert np.allclose(env_lookup(cube, (0.02, 0.05, 1.0)) * 255.0, cols["YP"], atol=0.51)
    assert np.allclose(env_lookup(cube, (0.02, -1.0, 0.05)) * 255.0, cols["ZM"], atol=0.51)
    os.remove(str(tmp_path / "ZM.png"))
    with pytest.raises(RuntimeError):
        vx.load_skybox_faces(str(tmp_path))


def _dds_cube(path, faces, layout):
    """Writes a cube-map .dds by the published layout (DDS_HEADER 124 B, DDS_PIXELFORMAT 32 B, optional DDS_HEADER_DXT10):
    faces uint8 [6, S, S, 4] RGBA; layout: "dx10_rgba", "legacy_bgra_mips" (full mip chain per face), "legacy_rgb24"."""
    S = faces.shape[1]
    mips = 1
    pf_flags, fourcc, bits, masks = 0x41, 0, 32, (0x00ff0000, 0x0000ff00, 0x000000ff, 0xff000000)
    tail = b""
    if layout == "dx10_rgba":
        pf_flags, fourcc, bits, masks = 0x4, 0x30315844, 0, (0, 0, 0, 0)
        tail = struct.pack("<5I", 28, 3, 0x4, 1, 0)  # DXGI_FORMAT_R8G8B8A8_UNORM, TEXTURE2D, TEXTURECUBE, arraySize 1
    elif layout == "legacy_bgra_mips":
        mips = int(np.log2(S)) + 1
    elif layout == "legacy_rgb24":
        pf_flags, bits, masks = 0x40, 24, (0x000000ff, 0x0000ff00, 0x00ff0000, 0)
    hdr = struct.pack("<4s7I44x", b"DDS ", 124, 0x1 | 0x2 | 0x4 | 0x1000 | (0x20000 if mips > 1 else 0), S, S, S * 4, 0, mips)
    hdr += struct.pack("<8I", 32, pf_flags, fourcc, bits, *masks)
    hdr += struct.pack("<5I", 0x1008 | (0x400000 if mips > 1 else 0), 0x200 | 0xfc00, 0, 0, 0)
    assert len(hdr) == 128
    body = b""
    for f in range(6):
        img = faces[f]
        for m in range(mips):
            lvl = img[::1 << m, ::1 << m]
            if layout == "dx10_rgba":
                body += lvl.tobytes()
            elif layout == "legacy_rgb24":
                body += np.ascontiguousarray(lvl[..., :3]).tobytes()
            else:
                body += np.ascontiguousarray(lvl[..., [2, 1, 0, 3]]).tobytes()
    open(path, "wb").write(hdr + tail + body)


@pytest.mark.parametrize("layout", ["dx10_rgba", "legacy_bgra_mips", "legacy_rgb24"])
def test_skybox_from_a_dds_cube_map(tmp_path, layout):
    """The reference loads its sky box from a .dds cube map (VTextureFactory::LoadTextureCubeFromFile,
    Renderer/Private/TextureFactory.cpp:28-67; Resources/Skybox/Skybox.dds is missing from the checkout).  Uncompressed
    cube maps in the DX10 and the legacy header layouts, with and without a mip chain, decode to the six RGBA faces;
    a 2D texture, a format the loader does not decode (signed BC5) or a truncated file is refused."""
    rng = np.random.default_rng(4)
    faces = rng.integers(0, 256, size=(6, 8, 8, 4), dtype=np.uint8)
    path = str(tmp_path / "sky.dds")
    _dds_cube(path, faces, layout)
    cube = vx.load_skybox_faces(path)
    want = faces.copy()
    if layout == "legacy_rgb24":
        want[..., 3] = 255
    assert cube.shape == (6, 8, 8, 4) and np.array_equal(cube, want)
    raw = bytearray(open(path, "rb").read())
    bad = str(tmp_path / "bad.dds")
    open(bad, "wb").write(raw[:-17])  # truncated
    with pytest.raises(RuntimeError):
        vx.load_skybox_faces(bad)
    flat = bytearray(raw)
    if layout == "dx10_rgba":
        flat[136:140] = struct.pack("<I", 0)      # miscFlag without TEXTURECUBE
    else:
        flat[112:116] = struct.pack("<I", 0)      # dwCaps2 without CUBEMAP
    open(bad, "wb").write(flat)
    with pytest.raises(RuntimeError):
        vx.load_skybox_faces(bad)
    bc = bytearray(raw)
    bc[80:88] = struct.pack("<I4s", 0x4, b"BC5S")  # signed BC5: a block-compressed format the loader does not decode
    open(bad, "wb").write(bc)
    with pytest.raises(RuntimeError):
        vx.load_skybox_faces(bad)
    # a damaged header: an absurd mip count (with and without DDSD_MIPMAPCOUNT set) is refused or ignored at once — never a
    # loop of 4e9 iterations or a shift by >= 32
    import time
    for flag in (0x20000, 0):
        huge = bytearray(raw)
        hdr_flags = struct.unpack("<I", raw[8:12])[0]
        huge[8:12] = struct.pack("<I", (hdr_flags & ~0x20000) | flag)
        huge[28:32] = struct.pack("<I", 0xffffffff)
        open(bad, "wb").write(huge)
        t0 = time.perf_counter()
        if flag and layout != "legacy_bgra_mips":
            with pytest.raises(RuntimeError):   # claims a full mip chain the file does not hold: truncated
                vx.load_skybox_faces(bad)
        elif flag:
            assert np.array_equal(vx.load_skybox_faces(bad), want)  # the count is clamped to the chain a face of this size can have
        elif layout == "legacy_bgra_mips":
            # without the flag the field is undefined: one level per face is read; this file's faces are a mip chain apart, so
            # the faces after the first come out wrong, but nothing hangs or crashes
            assert vx.load_skybox_faces(bad).shape == (6, 8, 8, 4)
        else:
            assert np.array_equal(vx.load_skybox_faces(bad), want)
        assert time.perf_counter() - t0 < 1.0
    # the suffix rule is case-insensitive, and the same for every caller
    upper = str(tmp_path / "SKY.DdS")
    open(upper, "wb").write(raw)
    assert np.array_equal(vx.load_skybox_faces(upper), want)


def _bc_decode_block(b, bc):
    """One 4x4 block of BC1 / BC2 / BC3 by the published rules ("Texture Block Compression in Direct3D 11"), float32 like the
    loader: end points c/31, c/63; palette at 1/3, 2/3 (BC1 with c0 <= c1: midpoint + transparent black); byte = floor(c*255 + 0.5)."""
    f = np.float32
    if bc in (4, 5):  # one / two interpolated 8-byte blocks: red (and green); blue 0, alpha 255
        def channel(blk):
            a0, a1 = int(blk[0]), int(blk[1])
            al = [a0, a1]
            for k in range(2, 8):
                if a0 > a1:
                    a = (f(8 - k) * f(a0) + f(k - 1) * f(a1)) / f(7)
                elif k < 6:
                    a = (f(6 - k) * f(a0) + f(k - 1) * f(a1)) / f(5)
                else:
                    a = f(0) if k == 6 else f(255)
                al.append(int(f(a) + f(0.5)))
            bits = int.from_bytes(bytes(blk[2:8]), "little")
            return [al[(bits >> (3 * i)) & 7] for i in range(16)]
        out = np.zeros((16, 4), np.uint8)
        out[:, 0] = channel(b[:8])
        if bc == 5:
            out[:, 1] = channel(b[8:16])
        out[:, 3] = 255
        return out.reshape(4, 4, 4)
    col = b if bc == 1 else b[8:]
    c0, c1 = int(col[0]) | int(col[1]) << 8, int(col[2]) | int(col[3]) << 8

    def expand(c):
        return np.array([f((c >> 11) & 31) / f(31), f((c >> 5) & 63) / f(63), f(c & 31) / f(31)], dtype=f)

    p0, p1 = expand(c0), expand(c1)
    four = bc != 1 or c0 > c1
    if four:
        p2, p3, a3 = p0 + (p1 - p0) * f(1.0 / 3.0), p0 + (p1 - p0) * f(2.0 / 3.0), 255
    else:
        p2, p3, a3 = p0 + (p1 - p0) * f(0.5), np.zeros(3, f), 0
    pal = [p0, p1, p2, p3]
    idx = int.from_bytes(bytes(col[4:8]), "little")
    out = np.zeros((16, 4), np.uint8)
    if bc == 3:
        a0, a1 = int(b[0]), int(b[1])
        alpha = [a0, a1]
        for k in range(2, 8):
            if a0 > a1:
                a = (f(8 - k) * f(a0) + f(k - 1) * f(a1)) / f(7)
            elif k < 6:
                a = (f(6 - k) * f(a0) + f(k - 1) * f(a1)) / f(5)
            else:
                a = f(0) if k == 6 else f(255)
            alpha.append(int(f(a) + f(0.5)))
        abits = int.from_bytes(bytes(b[2:8]), "little")
    for i in range(16):
        k = (idx >> (2 * i)) & 3
        out[i, :3] = (pal[k] * f(255) + f(0.5)).astype(np.uint8)
        if bc == 1:
            out[i, 3] = 255 if k < 3 or four else a3
        elif bc == 2:
            out[i, 3] = ((int(b[i >> 1]) >> ((i & 1) * 4)) & 15) * 17
        else:
            out[i, 3] = alpha[(abits >> (3 * i)) & 7]
    return out.reshape(4, 4, 4)


@pytest.mark.parametrize("bc,dx10", [(1, False), (2, False), (3, False), (1, True), (3, True), (4, False), (5, False), (4, True), (5, True)])
def test_skybox_from_a_block_compressed_dds_cube_map(tmp_path, bc, dx10):
    """SURVEY §8(f)-4's last leftover: the reference's Skybox.dds (missing from the checkout) goes through DirectXTex, which
    writes BC1 ... BC5 as readily as RGBA.  Cube maps of random blocks (every palette mode, both interpolated-alpha modes, a mip chain)
    in the legacy FourCC and the DX10 header layouts decode to what the published block rules give, texel for texel; a truncated
    file and BC7 are refused."""
    rng = np.random.default_rng(40 + bc)
    S, block = 12, 8 if bc in (1, 4) else 16
    mips = 3
    faces_blocks = []
    body = b""
    want = np.zeros((6, S, S, 4), np.uint8)
    for f in range(6):
        for m in range(mips):
            w = max(S >> m, 1)
            nbk = ((w + 3) // 4) ** 2
            blocks = rng.integers(0, 256, size=(nbk, block), dtype=np.uint8)
            if m == 0:
                blocks[0, block - 8:block - 4] = [0x00, 0x10, 0xff, 0xf0]   # c0 < c1: BC1's three-colour + transparent mode
                if bc == 3:
                    blocks[1, 0:2] = [10, 200]                              # a0 < a1: BC3's six-value alpha mode with 0 and 255
                bw = (w + 3) // 4
                for k in range(nbk):
                    tex = _bc_decode_block(blocks[k], bc)
                    by, bx = divmod(k, bw)
                    want[f, by * 4:by * 4 + 4, bx * 4:bx * 4 + 4] = tex[: min(4, S - by * 4), : min(4, S - bx * 4)]
            body += blocks.tobytes()
    fourcc = {1: b"DXT1", 2: b"DXT3", 3: b"DXT5", 4: b"ATI1", 5: b"ATI2"}[bc]
    hdr = struct.pack("<4s7I44x", b"DDS ", 124, 0x1 | 0x2 | 0x4 | 0x1000 | 0x20000 | 0x80000, S, S, ((S + 3) // 4) ** 2 * block, 0, mips)
    if dx10:
        hdr += struct.pack("<2I4s5I", 32, 0x4, b"DX10", 0, 0, 0, 0, 0)
        tail = struct.pack("<5I", {1: 71, 2: 74, 3: 77, 4: 80, 5: 83}[bc], 3, 0x4, 1, 0)
    else:
        hdr += struct.pack("<2I4s5I", 32, 0x4, fourcc, 0, 0, 0, 0, 0)
        tail = b""
    hdr += struct.pack("<5I", 0x1008 | 0x400000, 0x200 | 0xfc00, 0, 0, 0)
    assert len(hdr) == 128
    path = str(tmp_path / "sky_bc.dds")
    open(path, "wb").write(hdr + tail + body)
    cube = vx.load_skybox_faces(path)
    assert cube.shape == (6, S, S, 4) and np.array_equal(cube, want)
    bad = str(tmp_path / "bad.dds")
    open(bad, "wb").write((hdr + tail + body)[:-9])
    with pytest.raises(RuntimeError):
        vx.load_skybox_faces(bad)
    if dx10:
        open(bad, "wb").write(hdr + struct.pack("<5I", 98, 3, 0x4, 1, 0) + body)  # BC7_UNORM: not decoded
        with pytest.raises(RuntimeError):
            vx.load_skybox_faces(bad)


@pytest.mark.skipif(not os.path.isdir("/root/reference/VolumetricRaytracer/VolumetricRaytracer/Resources/Skybox"),
                    reason="the reference checkout (with its Resources/Skybox PNGs) is only present on the build machine")
def test_reference_skybox_pngs_decode():
    """The reference's own six sky box faces (data files, Resources/Skybox/*.png) through this build's PNG reader: 1024^2
    RGBA each, identical to what PIL decodes."""
    d = "/root/reference/VolumetricRaytracer/VolumetricRaytracer/Resources/Skybox"
    cube = vx.load_skybox_faces(d)
    assert cube.shape == (6, 1024, 1024, 4)
    try:
        from PIL import Image
    except ImportError:
        return
    for f, name in enumerate(("XP", "XM", "YP", "YM", "ZP", "ZM")):
        assert np.array_equal(cube[f], np.array(Image.open(os.path.join(d, name + ".png")).convert("RGBA")))

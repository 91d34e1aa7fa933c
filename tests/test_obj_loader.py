"""The OBJ / MTL reader and the OBJ -> Mesh-node converters against the reference's own obj tests
(obj/obj_test.go:12-190, transcribed) and node.go:101-309; the PNG decoder against its own writer."""
import os

import numpy as np
import pytest

from raytrace_amd import obj, phys


def test_parse_basic():  # TestParseOBJ_Basic, obj_test.go:12-43
    data = """
# Simple triangle
v 0.0 0.0 0.0
v 1.0 0.0 0.0
v 0.0 1.0 0.0
f 1 2 3
"""
    o = obj.ParseFS(obj.MapFS({"triangle.obj": data}), "triangle.obj")
    assert len(o.Vertices) == 3 and len(o.Faces) == 1


def test_parse_materials():  # TestParseOBJ_Materials, obj_test.go:46-98
    obj_data = """
# Cube with materials
v -1.0 -1.0 -1.0
v 1.0 -1.0 -1.0
v 1.0 1.0 -1.0
v -1.0 1.0 -1.0
v -1.0 -1.0 1.0
v 1.0 -1.0 1.0
v 1.0 1.0 1.0
v -1.0 1.0 1.0

usemtl Material001
mtllib cube.mtl

f 1 2 3 4
f 5 6 7 8
"""
    mtl_data = """
# Material definition
newmtl Material001
Kd 0.8 0.8 0.8
map_Kd texture.jpg
"""
    fsys = obj.MapFS({"cube.obj": obj_data, "cube.mtl": mtl_data, "texture.jpg": b"fake image data"})
    o = obj.ParseFS(fsys, "cube.obj")
    assert len(o.Materials) == 1
    mat = o.Materials["Material001"]
    assert mat.Diffuse == (0.8, 0.8, 0.8)
    assert mat.Texture == "texture.jpg"
    assert all(f.Material == "Material001" for f in o.Faces)


def test_parse_negative_indices():  # TestParseOBJ_NegativeIndices, obj_test.go:101-131
    data = """
v 0.0 0.0 0.0
v 1.0 0.0 0.0
v 1.0 1.0 0.0
v 0.0 1.0 0.0
f -4 -3 -2 -1
"""
    o = obj.ParseFS(obj.MapFS({"quad.obj": data}), "quad.obj")
    assert len(o.Faces) == 1
    assert [i[0] for i in o.Faces[0].Indices] == [1, 2, 3, 4]


def test_parse_invalid_syntax():  # TestParseOBJ_InvalidSyntax, obj_test.go:134-147
    with pytest.raises(obj.ParseError) as e:
        obj.ParseFS(obj.MapFS({"invalid.obj": "\nv 0.0 0.0\nf 1 2\n"}), "invalid.obj")
    assert e.value.Line == 2 and e.value.Filename == "invalid.obj"


def test_parse_empty_file_and_missing_file():  # TestParseOBJ_EmptyFile, obj_test.go:150-167
    o = obj.ParseFS(obj.MapFS({"empty.obj": ""}), "empty.obj")
    assert len(o.Vertices) == 0 and len(o.Faces) == 0
    with pytest.raises(obj.ParseError, match="failed to read file"):
        obj.ParseFS(obj.MapFS({}), "nothing.obj")


def test_index_forms_and_errors():  # obj.go:359-439
    data = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvn 0 0 1\nf 1/1/1 2/2/1 3//1\nf 1/2 2/1 3/1\n"
    o = obj.ParseFS(obj.MapFS({"a.obj": data}), "dir/sub/a.obj")  # only the base name is looked up
    assert o.Faces[0].Indices == [(1, 1, 1), (2, 2, 1), (3, 0, 1)]
    assert o.Faces[1].Indices == [(1, 2, 0), (2, 1, 0), (3, 1, 0)]
    for bad in ("f 1 2 4\n", "f 1 2 0\n", "f 1/1/1/1 2 3\n", "f 1 2 x\n"):
        with pytest.raises(obj.ParseError, match="invalid face index"):
            obj.ParseFS(obj.MapFS({"a.obj": "v 0 0 0\nv 1 0 0\nv 0 1 0\n" + bad}), "a.obj")
    with pytest.raises(obj.ParseError, match="defined before any newmtl"):
        obj.ParseFS(obj.MapFS({"a.obj": "mtllib m.mtl\n", "m.mtl": "Kd 1 1 1\n"}), "a.obj")


def _quad_obj():
    return """mtllib q.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
usemtl red
f 1/1 2/2 3/3 4/4
usemtl tex
f 1/1 2/2 5
f 1 1 2
"""


def test_convert_object_to_nodes():  # node.go:101-169, 266-309
    tex = np.zeros((2, 3, 4), dtype=np.uint8)
    tex[..., 0] = 200
    tex[..., 3] = 255
    fsys = obj.MapFS({"q.obj": _quad_obj(), "q.mtl": "newmtl red\nKd 1 0 0\nnewmtl tex\nmap_Kd t.png\n", "t.png": obj.encode_png_rgba8(tex)})
    o = obj.ParseFS(fsys, "q.obj")
    nodes = phys.ConvertObjectToNodes(o, fsys)
    assert [n.Name for n in nodes] == ["red", "tex"]
    red, texn = nodes
    assert red.Shape.Face.shape == (2, 3, 5)  # the quad is fan-triangulated: (1,2,3), (1,3,4)
    assert np.array_equal(red.Shape.Face[1, :, :3], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    assert np.array_equal(red.Shape.Face[1, :, 3:], [[0, 0], [1, 1], [0, 1]])
    assert isinstance(red.Material, phys.Emitter) and red.Material.Texture.Color == (1.0, 0.0, 0.0)
    # the degenerate face (two identical vertices) is dropped; a missing vt gives uv = (0, 0)
    assert texn.Shape.Face.shape == (1, 3, 5)
    assert np.array_equal(texn.Shape.Face[0, 2], [0, 0, 1, 0, 0])
    t = texn.Material.Texture
    assert isinstance(t, phys.TextureImage) and (t.Interp, t.WrapMode) == ("bilinear", "repeat")
    assert np.array_equal(t.Image, tex)
    # no materials at all: one "default" node with the reference's gray Emitter
    o2 = obj.ParseFS(obj.MapFS({"p.obj": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n"}), "p.obj")
    (n2,) = phys.ConvertObjectToNodes(o2, obj.MapFS({}))
    assert n2.Name == "default" and n2.Material.Texture.Color == (0.8, 0.8, 0.8)
    assert phys.ConvertObjectToShape(o2).Face.shape == (1, 3, 5)


@pytest.mark.parametrize("shape", [(1, 1), (5, 7), (64, 33)])
def test_png_round_trip(shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, size=(shape[0], shape[1], 4), dtype=np.uint8)
    assert np.array_equal(obj.decode_png(obj.encode_png_rgba8(img)), img)
    with pytest.raises(ValueError, match="not a PNG"):
        obj.decode_png(b"fake image data")


def _png(w, h, depth, ctype, rows, extra=b""):
    import struct, zlib

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xffffffff)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + extra + \
        chunk(b"IDAT", zlib.compress(b"".join(rows))) + chunk(b"IEND", b"")


def test_png_filters_and_colour_types():
    # every filter type on an RGB image: build the filtered rows by hand from known pixels
    px = np.array([[[10, 20, 30], [40, 60, 90], [250, 5, 7]], [[1, 2, 3], [200, 100, 50], [9, 9, 9]],
                   [[7, 7, 7], [8, 8, 8], [9, 10, 11]], [[0, 255, 0], [255, 0, 255], [3, 2, 1]], [[5, 6, 7], [8, 9, 10], [11, 12, 13]]], dtype=np.int32)
    h, w = px.shape[:2]
    flat = px.reshape(h, -1)
    rows = []
    for y, ft in enumerate([0, 1, 2, 3, 4]):
        cur, prev = flat[y], (flat[y - 1] if y else np.zeros_like(flat[0]))
        out = bytearray([ft])
        for i in range(w * 3):
            a = cur[i - 3] if i >= 3 else 0
            b = prev[i]
            c = prev[i - 3] if i >= 3 else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out.append((cur[i] - pred) & 255)
        rows.append(bytes(out))
    got = obj.decode_png(_png(w, h, 8, 2, rows))
    assert got.dtype == np.uint8 and np.array_equal(got[..., :3], px) and (got[..., 3] == 255).all()
    # 8-bit gray, and 16-bit RGBA (comes back alpha-premultiplied, as Go's Color.RGBA() reports)
    g = obj.decode_png(_png(2, 1, 8, 0, [b"\x00\x10\xf0"]))
    assert np.array_equal(g[0], [[16, 16, 16, 255], [240, 240, 240, 255]])
    import struct
    r16 = obj.decode_png(_png(1, 1, 16, 6, [b"\x00" + struct.pack(">HHHH", 65535, 32768, 0, 32768)]))
    assert r16.dtype == np.uint16 and np.array_equal(r16[0, 0], [32768, 32768 * 32768 // 65535, 0, 32768])


@pytest.mark.skipif(not os.path.isdir("/root/reference/obj/testdata/capsule.obj"), reason="the reference tree is not on this machine")
def test_reference_capsule_fixture_parses():
    """obj/testdata/capsule.obj (MeshLab export, 5252 vertices / 10000 faces, PNG texture), read in place."""
    fsys = obj.DirFS("/root/reference/obj/testdata/capsule.obj")
    o = obj.ParseFS(fsys, "capsule.obj")
    assert len(o.Vertices) == 5252 and len(o.Faces) == 10000
    assert o.Materials["material_0"].Texture if "material_0" in o.Materials else True
    nodes = phys.ConvertObjectToNodes(o, fsys)
    assert sum(len(n.Shape.Face) for n in nodes) <= 10000
    for n in nodes:
        n.Validate()


# ---- native scanner (rt_obj_scan_text) against the line parser -----------------------------------------------------


def _both_ways(files, name, monkeypatch):
    """Parse with the native scanner forced on (any size) and with it off; returns (native, lines)."""
    monkeypatch.setattr(obj, "_NATIVE_MIN_BYTES", 0)
    monkeypatch.setenv("RT_OBJ_NATIVE", "1")
    a = obj.ParseFS(obj.MapFS(files), name)
    monkeypatch.setenv("RT_OBJ_NATIVE", "0")
    b = obj.ParseFS(obj.MapFS(files), name)
    return a, b


def _same_object(a, b):
    for x, y, w in ((a.Vertices, b.Vertices, 3), (a.TexCoords, b.TexCoords, 2), (a.Normals, b.Normals, 3)):
        assert np.array_equal(np.asarray(x, dtype=np.float64).reshape(-1, w), np.asarray(y, dtype=np.float64).reshape(-1, w), equal_nan=True)
    assert list(a.Faces) == list(b.Faces)
    assert a.Materials == b.Materials


def test_native_scanner_is_used_and_agrees_with_the_line_parser(monkeypatch):
    text = "\n".join([
        "# a comment", "", "mtllib a.mtl b.mtl", "o thing", "g group one",
        "v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "v 0.5 0.5 1e0", "v -1.5e-3 +2.25 .5", "  v 3. 4 5  \r",
        "vt 0 0", "vt 1 0", "vt 1 1", "vt 0.25 0.75",
        "vn 0 0 1", "vn 0 1 0",
        "f 1 2 3", "usemtl red", "f 1/1 2/2 3/3 4/4", "f 1//1 2//1 5//2", "f 1/1/1 2/2/1   5/3/2",
        "usemtl blue stuff", "f -1 -2 -3", "f -7/-4/-2 2/2/-1 3/3/1 4/4/1 5/1/1",
        "usemtl red", "f 7 6 5", "s off", "lonetoken", "unknown directive here",
    ]) + "\n"
    files = {"m.obj": text, "a.mtl": "newmtl red\nKd 1 0 0\n", "b.mtl": "newmtl blue stuff\nKd 0 0 1\nmap_Kd t.png\n"}
    a, b = _both_ways(files, "m.obj", monkeypatch)
    assert isinstance(a.Faces, obj.FaceTable) and isinstance(b.Faces, list)
    _same_object(a, b)
    assert len(a.Faces) == 7 and a.Faces[2].Material == "red" and a.Faces[-1].Indices == [(7, 0, 0), (6, 0, 0), (5, 0, 0)]
    # the converters take the arrays directly and give the same meshes, node for node
    for name, sel in a.Faces.groups().items():
        want = phys._faces_of(b, [f for f in b.Faces if (f.Material or "default") == name])
        assert np.array_equal(phys._faces_of(a, (a.Faces, sel)), want), name
    assert np.array_equal(phys.ConvertObjectToShape(a).Face, phys.ConvertObjectToShape(b).Face)


@pytest.mark.parametrize("bad", [
    "v 1 2", "v 1 2 x", "v 1 2 3 4", "v 1  2 3", "vt 1 2 3", "vn 1 2", "f 1 2", "f 1 2 9", "f 1/ 2 3", "f 1/1/1/1 2 3", "f 0 1 2",
    "usemtl", "mtllib", "mtllib missing.mtl", "v 1_0 2 3", "v nan 0 0", "v 0x10 0 0", "f 1\t2 3", "v 1 2 3\x0b",
    "v 1 2 3 # trailing comment", "v 1e400 0 0", "f +1 2 3", "f 1/1 2 3", "\xa0v 1 2 3", "usemtl café",
])
def test_lines_the_scanner_leaves_to_the_line_parser_behave_as_before(bad, monkeypatch):
    """Anything but the well-formed subset: the scanner steps aside, so both ways give the same object or the same error."""
    text = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\n" + bad + "\nf 1 2 3\n"
    results = []
    for native in ("1", "0"):
        monkeypatch.setattr(obj, "_NATIVE_MIN_BYTES", 0)
        monkeypatch.setenv("RT_OBJ_NATIVE", native)
        try:
            results.append(obj.ParseFS(obj.MapFS({"m.obj": text}), "m.obj"))
        except obj.ParseError as e:
            results.append((e.Filename, e.Line, e.LineText, e.Msg))
    if isinstance(results[1], tuple):
        assert results[0] == results[1]
    else:
        _same_object(results[0], results[1])


def test_native_scanner_on_a_written_mesh(monkeypatch):
    """scenes.write_obj -> ParseFS -> ConvertObjectToShape gives the faces back, through the scanner and at its speed."""
    import time
    from raytrace_amd import scenes
    faces = scenes.scan_standin_mesh(5)  # 20480 faces
    obj_text, mtl_text = scenes.write_obj(faces)
    files = {"scan.obj": obj_text, "scan.mtl": mtl_text}
    assert len(obj_text) > obj._NATIVE_MIN_BYTES  # large enough to take the scanner by itself
    monkeypatch.setenv("RT_OBJ_NATIVE", "1")
    t0 = time.perf_counter()
    a = obj.ParseFS(obj.MapFS(files), "scan.obj")
    t_native = time.perf_counter() - t0
    assert isinstance(a.Faces, obj.FaceTable)
    got = phys.ConvertObjectToShape(a).Face
    keep = np.isfinite(faces).all(axis=(1, 2))
    assert len(got) == int(keep.sum()) and np.array_equal(got, faces[keep])
    monkeypatch.setenv("RT_OBJ_NATIVE", "0")
    t0 = time.perf_counter()
    b = obj.ParseFS(obj.MapFS(files), "scan.obj")
    t_lines = time.perf_counter() - t0
    _same_object(a, b)
    assert t_native < t_lines

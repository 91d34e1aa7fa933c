"""JSON scene ingress / export: the text the reference's marshallers produce (phys/json.go:52-100, scene.go:82-159,
node.go:48-97, shape_bvh.go:275-372, texture_*.go, material_emitter.go:33-68) read and written by
raytrace_amd.scene_json.  The reference holds no scene file and no test of its JSON, so these tests check the
format against statements taken from the marshallers' source and against itself (write -> read -> write);
parity with a Go-written file is unpinned."""
import json

import numpy as np
import pytest

from raytrace_amd import abi, phys, scenes
from raytrace_amd import scene_json as J


def test_floats_are_written_as_encoding_json_writes_them():
    # strconv 'f' with the shortest digits; 'e' below 1e-6 and from 1e21 with e-0X cleaned up to e-X
    table = [(0.0, "0"), (-0.0, "-0"), (1.0, "1"), (100.0, "100"), (0.5, "0.5"), (1e-6, "0.000001"), (9.99e-7, "9.99e-7"),
             (1e-7, "1e-7"), (1.234e-10, "1.234e-10"), (1e20, "100000000000000000000"), (1e21, "1e+21"), (1.5e300, "1.5e+300"),
             (123456789.125, "123456789.125"), (1 / 3, "0.3333333333333333"), (5e-324, "5e-324"), (-2.5e-9, "-2.5e-9"),
             (1e-100, "1e-100")]
    for x, want in table:
        assert J.go_float(x) == want, x
        assert float(J.go_float(x)) == x
    for bad in (float("nan"), float("inf"), -float("inf")):
        with pytest.raises(ValueError, match="unsupported value"):
            J.go_float(bad)
    assert J.go_string('a<b>&"c"') == '"a\\u003cb\\u003e\\u0026\\"c\\""'


def test_scene_layout_follows_the_marshallers():
    sc = scenes.cornell(dx=8, dy=8, spp=1)
    text = J.MarshalScene(sc)
    raw = json.loads(text)
    assert list(raw) == ["Camera", "Light", "Node", "RenderOptions"]  # a Go map: sorted keys (scene.go:107-113)
    assert list(raw["RenderOptions"]) == ["Seed", "RaysPerPixel", "MaxRayDepth", "Dx", "Dy"]  # a struct: declaration order
    node = raw["Node"][0]
    assert list(node) == ["Material", "Name", "Shape"]  # node.go:57-61
    assert list(node["Shape"]) == ["Data", "Type"]  # json.go:74-77
    assert raw["Camera"][0]["Type"] == "PinholeCamera" and raw["Light"][0]["Type"] == "PointLight"
    lam = next(n for n in raw["Node"] if n["Material"]["Type"] == "Lambertian")["Material"]["Data"]
    # Lambertian has no marshaller: its Texture field is the bare encoding of the TextureUniform inside, whose own
    # marshaller adds the in-band type (texture_uniform.go:27-37)
    assert list(lam["Texture"]) == ["Type", "Color"] and lam["Texture"]["Type"] == "TextureUniform"
    em = next(n for n in raw["Node"] if n["Material"]["Type"] == "Emitter")["Material"]["Data"]
    assert em["Type"] == "Emitter" and list(em["Texture"]) == ["Data", "Type"]  # material_emitter.go:34-46
    sph = next(n for n in raw["Node"] if n["Shape"]["Type"] == "Sphere")["Shape"]["Data"]
    assert list(sph) == ["Center", "Radius"]  # UVMap is omitempty (shape_sphere.go:255)


def _same(a, b):
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        a, b = np.asarray(a), np.asarray(b)
        if a.dtype.kind == "f" and not np.array_equal(np.signbit(a), np.signbit(b)):
            return False  # negative zeros keep their sign
        return np.array_equal(a, b)
    if hasattr(a, "__dataclass_fields__"):
        return type(a) is type(b) and all(_same(getattr(a, k), getattr(b, k)) for k in a.__dataclass_fields__)
    if isinstance(a, phys.Mesh):
        return isinstance(b, phys.Mesh) and np.array_equal(a.Face, b.Face)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        return float(a) == float(b)
    return a == b


@pytest.mark.parametrize("make", [
    lambda: scenes.cornell(dx=8, dy=8, spp=1),
    lambda: scenes.teapot(dx=8, dy=8, spp=1, aperture=2e6),
    lambda: scenes.scan(level=1, dx=8, dy=8, spp=1, tex_size=8),
    lambda: scenes.scan(level=1, dx=8, dy=8, spp=1, tex_size=8, lit=True),
    lambda: scenes.material(dx=8, dy=8, spp=1, tex_size=(6, 5)),
    lambda: scenes.charuco(dx=8, dy=6, spp=1, leds=True),
], ids=["cornell", "teapot", "scan", "scan-lit", "material", "charuco-leds"])
def test_every_configuration_survives_a_round_trip(make):
    sc = make()
    text = J.MarshalScene(sc)
    back = J.UnmarshalScene(text)
    assert _same(sc, back)
    again = J.MarshalScene(back) == text  # compared outside the assert: pytest would diff megabytes of text
    assert again
    back.Validate()


def test_all_shapes_textures_and_16_bit_images():
    img16 = (np.arange(3 * 5 * 4, dtype=np.uint32) * 977 % 65536).astype(np.uint16).reshape(3, 5, 4)
    checker = phys.TextureCheckerboard(phys.TextureUniform((1, 0, 0)),
                                       phys.TextureCheckerboard(phys.TextureImage(img16, "bilinear", "clamp", "a<b>.png"),
                                                                phys.TextureUniform((0, 0, 1e-9)), 3.5), 20.0)
    tri = phys.TriangleUV((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0), (1, 0), (0, 1), (0, 0, 1))
    xf = phys.TransformedShape(phys.TransformedShape(phys.Cylinder((0, 0, 0), (0, 0, 2), 0.5, 3.0),
                                                     phys.Transform((1, 2, 3), phys.RotationMatrixY(0.3), (1, 2, 1))),
                               phys.Transform((0, 0, -1), phys.RotationMatrixZ(-1.1), (1, 1, 1)))
    nodes = [
        phys.Node("a", phys.Sphere((0, 1, 2), 3.0, abi.UVMAP_BOX), phys.Lambertian(checker)),
        phys.Node("b", tri, phys.Metal((0.9, 0.8, 0.7), 0.05)),
        phys.Node("c", phys.Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0)), phys.Dielectric(1.5, 1.0, 0.01)),
        phys.Node("d", xf, phys.DebugNormal()),
        phys.Node("e", phys.Face(scenes.random_faces(1)[0]), phys.DebugUV()),
        phys.Node("f", phys.Quad((0, 0, 0), (0, 0, 1), 2.0, 1.0), phys.Emitter(checker)),
    ]
    cams = [phys.FocusableCamera((0, 0, 5), (0, 0, 0), (0, 1, 0), 0.5, 0.5, 0.1, 5.0),
            phys.CalibratedCamera(phys.CameraIntrinsics(1440, 1080, 1700.5, 1701.25, 720.0, 540.0, K1=-0.1, P2=1e-4, K6=2e-7),
                                  phys.CameraExtrinsics((1, 2, 3), (0, 0, 0), (0, 0, -1)))]
    sc = phys.Scene(Camera=cams, Node=nodes, Light=[], RenderOptions=phys.RenderOptions(-7, 3, 4, 5, 6))
    text = J.MarshalScene(sc)
    raw = json.loads(text)
    assert raw["Node"][0]["Shape"]["Data"]["UVMap"] == 2
    assert raw["Node"][3]["Shape"]["Data"]["Shape"]["Shape"]["Direction"] == {"X": 0, "Y": 0, "Z": 2}  # bare, untagged
    assert raw["RenderOptions"]["Seed"] == -7
    back = J.UnmarshalScene(text)
    assert _same(sc, back) and J.MarshalScene(back) == text
    assert back.Node[0].Material.Texture.Even.Odd.Image.dtype == np.uint16
    assert back.Node[0].Material.Texture.Even.Odd.FilePath == "a<b>.png"


def test_strict_reader_has_the_reference_registry_and_messages():
    sc = scenes.scan(level=0, dx=8, dy=8, spp=1, tex_size=4)
    sc.Node[0].Material = phys.Emitter(phys.TextureUniform((1, 2, 3)))
    text = J.MarshalScene(sc)  # orthographic camera, mesh, emitter: all registered, all with marshallers where needed
    assert _same(J.UnmarshalScene(text, strict=True), sc)
    # PinholeCamera, FocusableCamera and Triangle are never registered (no RegisterInterfaceType call names them)
    corn = J.MarshalScene(scenes.cornell(dx=8, dy=8, spp=1))
    with pytest.raises(ValueError, match=r"unsupported type: `PinholeCamera`; has it been registered\?"):
        J.UnmarshalScene(corn, strict=True)
    raw = json.loads(corn)
    raw["Camera"] = json.loads(text)["Camera"]
    # Lambertian.Texture is an interface field without a marshaller: encoding/json cannot fill it
    with pytest.raises(ValueError, match="cannot unmarshal object into Go struct field"):
        J.UnmarshalScene(json.dumps(raw), strict=True)
    assert len(J.UnmarshalScene(json.dumps(raw)).Node) == len(raw["Node"])  # the default reader types it by its keys
    # unknown names and in-band mismatches, both readers
    bad = json.loads(text)
    bad["Node"][0]["Shape"]["Type"] = "Torus"
    with pytest.raises(ValueError, match=r"unsupported type: `Torus`"):
        J.UnmarshalScene(json.dumps(bad))
    bad = json.loads(text)
    bad["Node"][0]["Material"]["Data"]["Type"] = "Lambertian"
    with pytest.raises(ValueError, match="invalid type: expected Emitter, got Lambertian"):
        J.UnmarshalScene(json.dumps(bad))
    bad = json.loads(text)
    bad["Node"][0]["Material"], bad["Node"][0]["Shape"] = bad["Node"][0]["Shape"], bad["Node"][0]["Material"]
    with pytest.raises(ValueError, match="expected Shape, got"):
        J.UnmarshalScene(json.dumps(bad))
    with pytest.raises(ValueError):
        J.UnmarshalScene("{not json")


def test_animated_camera_cannot_be_written():
    sc = scenes.cornell(dx=8, dy=8, spp=1)
    sc.Camera = [phys.AnimatedCamera(lambda u: sc.Camera[0])]
    with pytest.raises(ValueError, match="unsupported type: func"):
        J.MarshalScene(sc)


def _walk(bvh, faces_seen, depth=0):
    """Checks one BVH object of the export and returns (nodes, leaves, max depth)."""
    assert list(bvh) == ["Type", "Left", "Right", "Bounds"] and bvh["Type"] == "BVH"  # shape_bvh.go:276-281
    lo = np.array([bvh["Bounds"]["Min"][k] for k in "XYZ"])
    hi = np.array([bvh["Bounds"]["Max"][k] for k in "XYZ"])
    left, right = bvh["Left"], bvh["Right"]
    if left["Type"] != "BVH":  # a leaf: Left and Right are the same Group, or the same lone Face (shape_bvh.go:91-108)
        assert left == right
        shapes = [left] if left["Type"] == "Face" else left["Data"]["Shapes"]
        if left["Type"] == "Group":
            assert left["Data"]["Type"] == "Group" and 2 <= len(shapes)
        for s in shapes:
            assert s["Type"] == "Face"
            v = np.array([[p["Position"][k] for k in "XYZ"] for p in s["Data"]["Vertex"]])
            assert (v >= lo).all() and (v <= hi).all()
            faces_seen.append(v)
        return 1, 1, depth, len(shapes)
    n = [1, 0, depth, 0]
    for child in (left, right):
        assert child["Type"] == "BVH"
        c = _walk(child["Data"], faces_seen, depth + 1)
        clo = np.array([child["Data"]["Bounds"]["Min"][k] for k in "XYZ"])
        chi = np.array([child["Data"]["Bounds"]["Max"][k] for k in "XYZ"])
        assert (clo >= lo).all() and (chi <= hi).all()
        n = [n[0] + c[0], n[1] + c[1], max(n[2], c[2]), max(n[3], c[3])]
    return tuple(n)


@pytest.mark.parametrize("faces", [scenes.scan_standin_mesh(level=2), scenes.random_faces(300, seed=3), scenes.random_faces(1),
                                   scenes.random_faces(5, seed=9)], ids=["icosphere-320", "random-300", "one", "five"])
def test_exported_bvh_is_the_tree_the_oracle_builds(faces):
    """The BVH field of a Mesh: every face exactly once, boxes nested, and the same node / leaf / depth counts as
    the oracle's restatement of NewBVH on the same faces."""
    from oracle import binding
    mesh = phys.Mesh(faces)
    sc = phys.Scene(Camera=[phys.OrthographicCamera((0, 0, 500), (0, 0, 0), (0, 1, 0), 250.0, 250.0)],
                    Node=[phys.Node("m", mesh, phys.DebugNormal())], RenderOptions=phys.RenderOptions(0, 1, 1, 4, 4))
    raw = json.loads(J.MarshalScene(sc))
    data = raw["Node"][0]["Shape"]["Data"]
    assert list(data) == ["Face", "BVH"] and len(data["Face"]) == len(faces)
    seen = []
    nodes, leaves, depth, max_leaf = _walk(data["BVH"], seen)
    assert len(seen) == len(faces)
    want = {tuple(f[:, :3].ravel()) for f in mesh.Face}
    assert {tuple(v.ravel()) for v in seen} == want
    with binding.OracleScene(sc) as orc:
        info = orc.bvh_info(0)
    assert (nodes, leaves, depth, max_leaf) == (info.n_nodes, info.n_leaves, info.max_depth, info.max_leaf_size)
    tree, leaf = J.mesh_bvh(mesh.Face)
    assert len(tree) == nodes and sorted(leaf.tolist()) == list(range(len(faces)))


def test_threaded_passes_of_the_builder_give_the_same_tree(monkeypatch):
    """Nodes of 131 072 or more faces run their bounds, binning and partition passes in chunks on several threads, and the
    flattened arrays are written by three threads side by side: the tree must be the one a single thread builds, node for
    node and face for face, whatever the number of chunks, and have the oracle's counts."""
    from oracle import binding
    faces = scenes.scan_standin_mesh(level=7)  # 327 680 faces: the root and its children are wide nodes
    out = []
    for threads in ("1", "7", "16"):
        monkeypatch.setenv("RT_HOST_THREADS", threads)
        tree, leaf = J.mesh_bvh(faces)
        out.append((tree.tobytes(), leaf.tobytes(), len(tree)))
    assert out[0] == out[1] == out[2]
    sc = phys.Scene(Camera=[phys.OrthographicCamera((0, 0, 500), (0, 0, 0), (0, 1, 0), 250.0, 250.0)],
                    Node=[phys.Node("m", phys.Mesh(faces), phys.DebugNormal())], RenderOptions=phys.RenderOptions(0, 1, 1, 4, 4))
    with binding.OracleScene(sc) as orc:
        info = orc.bvh_info(0)
    tree = np.frombuffer(out[0][0], dtype=abi.BVH_NODE_DTYPE)
    assert len(tree) == info.n_nodes and int((tree["right"] < 0).sum()) == info.n_leaves
    assert int(tree["count"].max()) == info.max_leaf_size


def test_image_pixels_can_come_from_the_file_path(tmp_path):
    """A TextureImage written without pixels ("Image": null) is loaded again from FilePath, the field the
    reference keeps "for serialization" (texture_image.go:20)."""
    from raytrace_amd import obj
    img = scenes.procedural_texture(8)
    (tmp_path / "t.png").write_bytes(obj.encode_png_rgba8(img))
    text = J.MarshalScene(phys.Scene(Camera=[phys.OrthographicCamera((0, 0, 5), (0, 0, 0), (0, 1, 0), 2.0, 2.0)],
                                     Node=[phys.Node("q", phys.Quad((0, 0, 0), (0, 0, 1), 1.0, 1.0),
                                                     phys.Emitter(phys.TextureImage(None, "nearest", "repeat", "t.png")))],
                                     RenderOptions=phys.RenderOptions(0, 1, 1, 4, 4)))
    assert '"Image":null' in text
    back = J.UnmarshalScene(text, asset_dir=str(tmp_path))
    t = back.Node[0].Material.Texture
    assert np.array_equal(t.Image, obj.decode_png(obj.encode_png_rgba8(img))) and (t.Interp, t.WrapMode, t.FilePath) == ("nearest", "repeat", "t.png")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "scan", "material", "charuco"])
def test_a_scene_read_from_json_renders_the_same_image(name):
    """Scene -> JSON -> scene, both rendered through the C ABI: same bytes, same counters."""
    sc = {"cornell": lambda: scenes.cornell(dx=96, dy=96, spp=8),
          "scan": lambda: scenes.scan(level=4, dx=128, dy=96, spp=4, tex_size=64),
          "material": lambda: scenes.material(dx=96, dy=96, spp=4, tex_size=(40, 36)),
          "charuco": lambda: scenes.charuco(dx=120, dy=90, spp=2)}[name]()
    back = J.UnmarshalScene(J.MarshalScene(sc))
    a = phys.Render(sc)
    b = phys.Render(back)
    assert np.array_equal(a.Image, b.Image)
    assert (a.Stats.TotalRays, a.Stats.RaysLeftScene, a.Stats.RaysExceededDepth) == \
           (b.Stats.TotalRays, b.Stats.RaysLeftScene, b.Stats.RaysExceededDepth)


def test_bvh_build_reports_the_size_it_needs():
    """rt_bvh_build with too small a node array writes nothing and says how many nodes the tree has."""
    import ctypes as C
    from raytrace_amd import _lib
    lib = _lib.host_library()
    faces = np.ascontiguousarray(scenes.random_faces(40, seed=5))
    leaf = np.full(40, -7, dtype=np.int32)
    nodes = np.zeros(1, dtype=abi.BVH_NODE_DTYPE)
    count = C.c_uint64(0)
    assert lib.rt_bvh_build(faces.ctypes.data, 40, nodes.ctypes.data, 1, leaf.ctypes.data, C.byref(count)) != abi.RT_OK
    assert count.value > 1 and (leaf == -7).all()
    tree, order = J.mesh_bvh(faces)
    assert len(tree) == count.value and sorted(order.tolist()) == list(range(40))
    inner = tree[tree["right"] >= 0]
    assert (inner["count"] == 0).all() and (tree[tree["right"] < 0]["count"] > 0).all()
    assert lib.rt_bvh_build(faces.ctypes.data, 0, nodes.ctypes.data, 1, leaf.ctypes.data, C.byref(count)) != abi.RT_OK


def _create_from_text(text):
    """rt_scene_create_from_json on a text that must be refused while it is read (no GPU is touched before the reader is done)."""
    import ctypes as C
    from raytrace_amd import _lib
    lib = _lib.host_library()
    handle = C.c_void_p()
    raw = text.encode()
    rc = lib.rt_scene_create_from_json(raw, len(raw), C.byref(handle), None)
    return rc, (lib.rt_last_error() or b"").decode(), handle


def _scene_with_texture(texture_json):
    return ('{"Camera":[],"Light":[],"Node":[{"Name":"n","Shape":{"Type":"Sphere","Data":{"Center":{"X":0,"Y":0,"Z":0},"Radius":1,"UVMap":0}},'
            '"Material":{"Type":"Emitter","Data":{"Type":"Emitter","Texture":' + texture_json + '}}}],"RenderOptions":{"Seed":0,"RaysPerPixel":1,"MaxRayDepth":1,"Dx":8,"Dy":8}}')


def test_an_image_stride_that_would_wrap_the_bounds_check_is_refused():
    """Stride comes straight from the text: 2^62 x (h - 1) wraps to 0 in 64 bits and would pass a check written as a product."""
    import base64
    pix = base64.b64encode(bytes(4 * 2 * 5)).decode()
    for stride in (2 ** 62, 2 ** 61 + 8, 9 * 10 ** 18):
        image = '{"Pix":"%s","Stride":%d,"Rect":{"Min":{"X":0,"Y":0},"Max":{"X":2,"Y":5}}}' % (pix, stride)
        tex = '{"Type":"TextureImage","Data":{"Image":%s,"FilePath":"","Interp":"nearest","WrapMode":"repeat"}}' % image
        rc, msg, handle = _create_from_text(_scene_with_texture(tex))
        assert rc == abi.RT_ERR_INVALID and "stride" in msg and not handle.value, (stride, rc, msg)


def test_textures_and_shapes_nested_beyond_the_limit_are_refused_not_followed():
    depth = 20000  # enough frames of the reader's recursion to run a thread's stack out, were it followed
    flat = '{"Type":"TextureUniform","Data":{"Type":"TextureUniform","Color":{"X":1,"Y":1,"Z":1}}}'
    head = '{"Type":"TextureCheckerboard","Data":{"Type":"TextureCheckerboard","Frequency":1,"Even":' + flat + ',"Odd":'
    rc, msg, handle = _create_from_text(_scene_with_texture(head * depth + flat + "}}" * depth))
    assert rc == abi.RT_ERR_INVALID and "nested deeper" in msg and not handle.value, (rc, msg[:200])
    sphere = '{"Type":"Sphere","Data":{"Center":{"X":0,"Y":0,"Z":0},"Radius":1,"UVMap":0}}'
    head = '{"Type":"TransformedShape","Data":{"Shape":'
    tail = ',"Transform":{"Translate":{"X":0,"Y":0,"Z":0},"Rotate":{"M":[[1,0,0],[0,1,0],[0,0,1]]},"Scale":{"X":1,"Y":1,"Z":1}}}}'
    text = ('{"Camera":[],"Light":[],"Node":[{"Name":"n","Shape":' + head * depth + sphere + tail * depth + ',"Material":{"Type":"DebugNormal","Data":{}}}],'
            '"RenderOptions":{"Seed":0,"RaysPerPixel":1,"MaxRayDepth":1,"Dx":8,"Dy":8}}')
    rc, msg, handle = _create_from_text(text)
    assert rc == abi.RT_ERR_INVALID and "nested deeper" in msg and not handle.value, (rc, msg[:200])

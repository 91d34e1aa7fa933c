"""rt_scene_create_from_json on the GPU: the library reads the text Scene.MarshalJSON writes (scene.go:82-159,
json.go:52-100) and the render is compared with the oracle built from the same text, and with the scene the
host-side flattener uploads (which must give the very same bytes: both describe the same arrays)."""
import ctypes as C

import numpy as np
import pytest

from raytrace_amd import abi, phys, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.load()
    return binding


def zoo() -> phys.Scene:
    """Every shape, material, texture and camera kind the format can carry, transformed shapes nested."""
    rng = np.random.default_rng(3)
    tex8 = rng.integers(0, 256, size=(9, 11, 4), dtype=np.uint8)
    tex16 = rng.integers(0, 65536, size=(5, 7, 4)).astype(np.uint16)
    tex16[..., :3] = np.minimum(tex16[..., :3], tex16[..., 3:4])
    check = phys.TextureCheckerboard(phys.TextureUniform((0.9, 0.2, 0.1)), phys.TextureImage(tex8, "bilinear", "clamp"), 6.0)
    ico_v, ico_t = scenes.icosphere(2)
    mesh = phys.NewMesh(scenes.faces_from(ico_v * 0.8 + np.array([2.0, 0.0, 0.0]), np.abs(ico_v[:, :2]) * 0.5, ico_t))
    xf = phys.Transform((0.0, 2.0, 0.0), phys.RotationMatrixZ(0.4), (1.0, 0.5, 1.0))
    xf2 = phys.Transform((-2.0, 0.0, 0.5), phys.RotationMatrixX(0.7), (1.0, 1.0, 1.0))
    nodes = [
        phys.Node("floor", phys.Quad((0.0, 0.0, -1.0), (0.0, 0.0, 1.0), 12.0, 12.0), phys.Lambertian(check)),
        phys.Node("ball", phys.Sphere((0.0, 0.0, 0.0), 0.8, 2), phys.Emitter(phys.TextureImage(tex16, "", "repeat"))),
        phys.Node("mesh", mesh, phys.DebugNormal()),
        phys.Node("can", phys.Cylinder((-2.0, -2.0, -1.0), (0.0, 0.0, 1.0), 0.5, 1.5), phys.Metal((0.8, 0.8, 0.6), 0.1)),
        phys.Node("squashed", phys.TransformedShape(phys.Sphere((0.0, 0.0, 0.0), 0.7), xf), phys.Dielectric(1.5, 1.0, 0.05)),
        phys.Node("nested", phys.TransformedShape(phys.TransformedShape(phys.Cylinder((0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.3, 1.0), xf), xf2),
                  phys.DebugUV()),
        phys.Node("tri", phys.Triangle((2.0, 2.0, -0.5), (3.0, 2.0, -0.5), (2.0, 3.0, 0.5)), phys.Lambertian(phys.TextureUniform((0.2, 0.7, 0.3)))),
        phys.Node("triuv", phys.TriangleUV((2.0, -3.0, -0.5), (3.0, -3.0, -0.5), (2.0, -2.0, 0.5), (0.0, 0.0), (1.0, 0.0), (0.0, 1.0),
                                           (0.0, -0.7071067811865476, 0.7071067811865476)), phys.Emitter(check)),
        phys.Node("face", phys.Face(np.array([[-3.0, 2.0, -0.5, 0.0, 0.0], [-2.0, 2.0, -0.5, 1.0, 0.0], [-3.0, 3.0, 0.2, 0.0, 1.0]])),
                  phys.DebugUV()),
    ]
    cams = [
        phys.OrthographicCamera((6.0, -7.0, 5.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 9.0, 9.0),
        phys.FocusableCamera((6.0, -7.0, 5.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 0.9, 0.9, 0.005, 1.0),
        phys.PinholeCamera((-4.0, -6.0, -3.0), (0.0, -12.0, 1.0), (8.0, 0.0, 0.0), (0.0, 0.0, 8.0)),
        phys.CalibratedCamera(phys.CameraIntrinsics(96, 96, 110.0, 112.0, 47.5, 48.5, 0.05, -0.01, 0.001, -0.002, 0.003, 0.0, 0.0, 0.0),
                              phys.CameraExtrinsics((5.0, 6.0, 7.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0))),
    ]
    lights = [phys.PointLight((3.0, -4.0, 8.0), (40.0, 38.0, 35.0)), phys.PointLight((-5.0, 2.0, 6.0), (10.0, 12.0, 20.0))]
    return phys.Scene(Camera=cams, Node=nodes, Light=lights, RenderOptions=phys.RenderOptions(7, 8, 5, 96, 96))


def _close(a, b):
    diff = np.abs(a.astype(int) - b.astype(int))
    return diff.max() <= 1 and (diff > 0).mean() < 1e-3, (diff.max(), (diff > 0).mean())


def test_every_kind_of_value_read_by_the_library_renders_as_the_oracle(ob):
    sc = zoo()
    sc.Validate()
    text = sc.MarshalJSON()
    back = phys.Scene.UnmarshalJSON(text)  # what the oracle is built from: the same text
    dev, cams, opts = phys.DeviceScene.from_json(text)
    with dev, phys.DeviceScene(sc) as flat, ob.OracleScene(back) as orc:
        assert dev.json_info.n_cameras == 4 and dev.json_info.n_nodes == len(sc.Node) and dev.json_info.n_lights == 2
        assert (opts.Seed, opts.RaysPerPixel, opts.MaxRayDepth, opts.Dx, opts.Dy) == (7, 8, 5, 96, 96)
        for i, cam in enumerate(sc.Camera):
            assert bytes(cams[i]) == bytes(phys.camera_desc(cam)), f"camera {i} read differently"
            art = dev.render(cams[i], opts, abi.MODE_REFERENCE)
            ref = flat.render(cam, sc.RenderOptions, abi.MODE_REFERENCE)
            assert np.array_equal(art.Image, ref.Image), f"camera {i}: JSON route and flattener route differ"
            assert (art.Stats.TotalRays, art.Stats.RaysLeftScene, art.Stats.RaysExceededDepth, art.Stats.ShadowRays) == \
                   (ref.Stats.TotalRays, ref.Stats.RaysLeftScene, ref.Stats.RaysExceededDepth, ref.Stats.ShadowRays)
            img, st, _, _ = orc.render(back.Camera[i], back.RenderOptions)
            assert abs(art.Stats.TotalRays - st.total_rays) <= 1e-3 * st.total_rays
            diff = np.abs(art.Image[..., :3].astype(int) - img[..., :3].astype(int))
            assert (diff > 1).mean() < 5e-3, (i, (diff > 1).mean(), diff.max())
            assert (img[..., :3].max(axis=2) > 0).mean() > 0.1


@pytest.mark.parametrize("name", ["scan", "teapot", "charuco"])
def test_configurations_through_the_json_route(ob, name):
    sc = {"scan": lambda: scenes.scan(level=5, dx=160, dy=96, spp=2, tex_size=64, lit=True),
          "teapot": lambda: scenes.teapot(dx=160, dy=96, spp=2),
          "charuco": lambda: scenes.charuco(dx=160, dy=120, spp=2)}[name]()
    text = sc.MarshalJSON()
    back = phys.Scene.UnmarshalJSON(text)
    dev, cams, opts = phys.DeviceScene.from_json(text)
    with dev, phys.DeviceScene(sc) as flat, ob.OracleScene(back) as orc:
        for mode in (abi.MODE_REFERENCE, abi.MODE_FAST):
            art = dev.render(cams[0], opts, mode)
            ref = flat.render(sc.Camera[0], sc.RenderOptions, mode)
            assert np.array_equal(art.Image, ref.Image)
            assert art.Stats.TotalRays == ref.Stats.TotalRays and art.Stats.RaysLeftScene == ref.Stats.RaysLeftScene
        img, st, _, _ = orc.render(back.Camera[0], back.RenderOptions)
        assert abs(art.Stats.TotalRays - st.total_rays) <= 1e-3 * st.total_rays
        diff = np.abs(art.Image[..., :3].astype(int) - img[..., :3].astype(int))
        assert (diff > 1).mean() < 5e-3, ((diff > 1).mean(), diff.max())
        if dev.json_info.n_nodes and name != "charuco":
            assert dev.bvh_info(0).n_faces == flat.bvh_info(0).n_faces


def test_errors_of_the_json_route():
    from raytrace_amd import _lib
    lib = _lib.load()

    def rc(text):
        h = C.c_void_p()
        code = lib.rt_scene_create_from_json(text.encode(), len(text.encode()), C.byref(h), None)
        assert not h.value or code == abi.RT_OK
        if h.value:
            lib.rt_scene_destroy(h)
        return code, (lib.rt_last_error() or b"").decode()

    node = '{"Name":"a","Shape":{"Data":{"Center":{"X":0,"Y":0,"Z":0},"Radius":1},"Type":"Sphere"},"Material":{"Data":%s,"Type":"%s"}}'
    assert rc('{"Node":[' + node % ("{}", "DebugUV") + "]}")[0] == abi.RT_OK  # info may be NULL
    code, msg = rc('{"Node":[' + node % ('{"Image":null,"FilePath":"a.png"}', "TextureImage") + "]}")
    assert code == abi.RT_ERR_INVALID and "expected Material" in msg
    code, msg = rc('{"Node":[' + node % ('{"Type":"Emitter","Texture":{"Data":{"Image":null,"FilePath":"a.png","Interp":"","WrapMode":""},"Type":"TextureImage"}}', "Emitter") + "]}")
    assert code == abi.RT_ERR_UNSUPPORTED and "a.png" in msg
    code, msg = rc('{"Node":[' + node % ('{"Type":"Lambertian","Texture":null}', "Emitter") + "]}")
    assert code == abi.RT_ERR_INVALID and "invalid type: expected Emitter, got Lambertian" in msg
    code, msg = rc('{"Node":[' + node % ("{}", "Glass") + "]}")
    assert code == abi.RT_ERR_INVALID and "unsupported type: `Glass`; has it been registered?" in msg
    code, msg = rc('{"Node":[' + node % ("{}", "DebugUV") + "," + node % ("{}", "DebugNormal") + "]}")
    assert code == abi.RT_ERR_INVALID and "duplicate node name: a" in msg
    code, msg = rc('{"Node":[{"Name":"a"')
    assert code == abi.RT_ERR_INVALID and "unexpected end of JSON input" in msg

"""The C-ABI library builds, loads on a machine without a GPU, and exports what include/rt_hip.h declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def library():
    from raytrace_amd import _lib
    path = _lib.lib_path()
    if not os.path.exists(path):  # a fresh checkout: the library is a build product, not a tracked file
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return _lib.open_library()


def test_header_declares_the_entry_points():
    names = declared_functions()
    for must in ("rt_init", "rt_shutdown", "rt_last_error", "rt_scene_create", "rt_scene_destroy", "rt_render", "rt_render_film",
                 "rt_film_resolve", "rt_cancel", "rt_intersect_batch", "rt_cast_batch", "rt_scene_bvh_info", "rt_obj_scan_text",
                 "rt_obj_scan_free"):
        assert must in names, must


def test_library_exports_every_declared_symbol():
    lib = library()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in rt_hip.h but not exported"


def test_struct_sizes_match_the_header():
    """ctypes mirrors against the sizes the C compiler computes for the header's structs."""
    import subprocess, sys, tempfile
    from raytrace_amd import abi
    pairs = [("rt_sphere", abi.Sphere), ("rt_quad", abi.Quad), ("rt_cylinder", abi.Cylinder), ("rt_triangle", abi.Triangle),
             ("rt_face", abi.Face), ("rt_mesh", abi.Mesh), ("rt_transformed", abi.Transformed), ("rt_image", abi.Image),
             ("rt_texture", abi.Texture), ("rt_material", abi.Material), ("rt_light", abi.Light), ("rt_node", abi.Node),
             ("rt_scene_desc", abi.SceneDesc), ("rt_camera_desc", abi.CameraDesc), ("rt_render_opts", abi.RenderOpts),
             ("rt_stats", abi.Stats), ("rt_hit", abi.Hit), ("rt_bvh_info", abi.BvhInfo), ("rt_obj_event", abi.ObjEvent),
             ("rt_obj_scan", abi.ObjScan), ("rt_shade_in", abi.ShadeIn), ("rt_shade_out", abi.ShadeOut),
             ("rt_json_scene_info", abi.JsonSceneInfo)]
    src = '#include <stdio.h>\n#include "rt_hip.h"\nint main(void){' + "".join(
        f'printf("%zu\\n", sizeof({c}));' for c, _ in pairs) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        cfile, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(cfile, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    for (cname, ct), size in zip(pairs, sizes):
        assert C.sizeof(ct) == size, f"{cname}: header {size} bytes, ctypes mirror {C.sizeof(ct)}"


def test_product_path_fails_loudly_without_a_gpu():
    """No CPU fallback: on a machine without a GPU the library refuses to initialise and says why."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = library()
    ids = (C.c_int * 1)(0)
    rc = lib.rt_init(ids, 1)
    assert rc != 0
    assert len(lib.rt_last_error()) > 0
    from raytrace_amd import phys, scenes
    with pytest.raises(Exception):
        phys.Render(scenes.scan(level=1, dx=16, dy=16, spp=1, tex_size=8))


def test_bvh_node_record_matches_the_header():
    """rt_bvh_build's node record is read through a numpy dtype: same size and field offsets as the C struct."""
    import subprocess, tempfile
    import numpy as np
    from raytrace_amd import abi
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "rt_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(rt_bvh_node), '
           'offsetof(rt_bvh_node, bounds_min), offsetof(rt_bvh_node, bounds_max), offsetof(rt_bvh_node, right), '
           'offsetof(rt_bvh_node, first), offsetof(rt_bvh_node, count));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        cfile, exe = os.path.join(d, "s.c"), os.path.join(d, "s")
        open(cfile, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe])
        want = [int(x) for x in subprocess.check_output([exe]).split()]
    dt = np.dtype(abi.BVH_NODE_DTYPE)
    got = [dt.itemsize] + [dt.fields[k][1] for k in ("bounds_min", "bounds_max", "right", "first", "count")]
    assert got == want


def test_json_reader_parses_on_the_host_and_stops_at_the_missing_gpu():
    """rt_scene_create_from_json reads the whole text before it touches a device: a malformed or unsupported text is
    reported as such on any machine, and a text that reads well ends at rt_scene_create (no GPU here: not initialised)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raytrace_amd import abi, scenes
    lib = library()

    def rc(text):
        raw = text.encode()
        h = C.c_void_p()
        code = lib.rt_scene_create_from_json(raw, len(raw), C.byref(h), None)
        assert not h.value
        return code, (lib.rt_last_error() or b"").decode()

    for sc in (scenes.cornell(dx=16, dy=16, spp=1), scenes.material(dx=16, dy=16, spp=1, tex_size=(8, 8)),
               scenes.scan(level=2, dx=16, dy=16, spp=1, tex_size=8, lit=True)):
        code, msg = rc(sc.MarshalJSON())
        assert code == abi.RT_ERR_NOT_INITIALIZED, msg
    for text, want in [
        ('{"Node":[{"Name":"a"', "unexpected end of JSON input"),
        ('{"Camera":[{"Data":{},"Type":"FisheyeCamera"}]}', "unsupported type: `FisheyeCamera`; has it been registered?"),
        ('{"Light":[{"Data":{},"Type":"Sphere"}]}', "expected Light, got *phys.Sphere"),
        ('{"Node":[{"Name":"n","Material":{"Data":{},"Type":"DebugUV"},"Shape":{"Data":{"Center":{"X":"1"}},"Type":"Sphere"}}]}',
         "not a number"),
        ('{"Node":[{"Name":"n","Material":{"Data":{},"Type":"DebugUV"},"Shape":{"Data":{"Face":[],"BVH":{"Type":"Group"}},"Type":"Mesh"}}]}',
         "invalid type: expected BVH, got Group"),
        ('{"RenderOptions":{"Dx":1.5}}', "fraction"),
        ('{"Node":[]} x', "after top-level value"),
    ]:
        code, msg = rc(text)
        assert code == abi.RT_ERR_INVALID and want in msg, (text, code, msg)

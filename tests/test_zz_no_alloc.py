"""rt_render_film_noalloc against rt_render_film: the same film bit for bit and the same statistics, on two calls in a
row (the second on a smaller frame, so the static scratch still holds the first call's schedule and counters), and the
limits of the entry point.  Collected last on purpose: the round's GPU run stops at the first failure."""
import ctypes as C

import numpy as np
import pytest

from raytrace_amd import abi, phys, scenes

pytestmark = pytest.mark.gpu


def _film(lib, entry, dev, sc, dx, dy, spp, mode=abi.MODE_FAST, band=(0, 1)):
    import torch
    from raytrace_amd import _lib
    o = phys.RenderOptions(sc.RenderOptions.Seed, spp, sc.RenderOptions.MaxRayDepth, dx, dy)
    film = torch.zeros((dy, dx, 3), dtype=torch.float64, device="cuda")
    cd = phys.camera_desc(sc.Camera[0])
    ro = phys.render_opts(o, mode, band[0], band[1], torch.cuda.current_stream().cuda_stream)
    st = abi.Stats()
    _lib.check(entry(dev.handle, C.byref(cd), C.byref(ro), C.c_void_p(film.data_ptr()), C.byref(st)), lib)
    torch.cuda.synchronize()
    return film.cpu().numpy(), st


def _same(a, b):
    fa, sa = a
    fb, sb = b
    assert np.array_equal(fa, fb)
    assert fa.any()
    for f in ("total_rays", "rays_exceeded_depth", "rays_left_scene", "shadow_rays", "culled_rays", "traverse_rays"):
        assert getattr(sa, f) == getattr(sb, f), f


def test_no_alloc_render_matches_rt_render_film_twice():
    from raytrace_amd import _lib
    lib = _lib.load()
    sc = scenes.scan(level=5, dx=512, dy=384, spp=34, tex_size=256)
    with phys.DeviceScene(sc) as dev:
        # 3072 tiles x 34 samples: pilot pass, occupancy map and the cost-ordered schedule are all live
        big = _film(lib, lib.rt_render_film_noalloc, dev, sc, 512, 384, 34)
        _same(big, _film(lib, lib.rt_render_film, dev, sc, 512, 384, 34))
        assert big[1].culled_rays > 0
        # a smaller frame right after: the scratch still holds the schedule and the counters of the first call
        small = _film(lib, lib.rt_render_film_noalloc, dev, sc, 160, 96, 3)
        _same(small, _film(lib, lib.rt_render_film, dev, sc, 160, 96, 3))
        # the first frame again, and a band of it in reference mode
        _same(_film(lib, lib.rt_render_film_noalloc, dev, sc, 512, 384, 34), big)
        _same(_film(lib, lib.rt_render_film_noalloc, dev, sc, 200, 150, 2, abi.MODE_REFERENCE, (1, 3)),
              _film(lib, lib.rt_render_film, dev, sc, 200, 150, 2, abi.MODE_REFERENCE, (1, 3)))


def test_no_alloc_render_leaves_the_free_device_memory_as_it_was():
    """The entry point's whole contract: on a handle that has rendered once, a call (pilot pass, occupancy map and schedule
    included: the camera changes) allocates and frees nothing on the device."""
    import torch
    from raytrace_amd import _lib
    lib = _lib.load()
    sc = scenes.scan(level=5, dx=512, dy=384, spp=34, tex_size=256)
    o = phys.RenderOptions(0, 34, 10, 512, 384)
    film = torch.zeros((384, 512, 3), dtype=torch.float64, device="cuda")
    ro = phys.render_opts(o, abi.MODE_FAST, 0, 1, torch.cuda.current_stream().cuda_stream)
    with phys.DeviceScene(sc) as dev:
        other = phys.OrthographicCamera(LookFrom=(70.0e9, 60.0e9, 100.0e9), LookAt=(0.0, 0.0, 0.0), VUp=(0.0, 0.0, 1.0), FOVHeight=250.0,
                                        FOVWidth=250.0 * 512 / 384)
        cams = [phys.camera_desc(sc.Camera[0]), phys.camera_desc(other)]
        _lib.check(lib.rt_render_film_noalloc(dev.handle, C.byref(cams[0]), C.byref(ro), C.c_void_p(film.data_ptr()), None), lib)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        lowest = free_before
        for cd in (cams[1], cams[0], cams[1]):
            st = abi.Stats()
            _lib.check(lib.rt_render_film_noalloc(dev.handle, C.byref(cd), C.byref(ro), C.c_void_p(film.data_ptr()), C.byref(st)), lib)
            lowest = min(lowest, torch.cuda.mem_get_info()[0])
            assert st.total_rays == 512 * 384 * 34
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before and lowest == free_before


def test_no_alloc_render_of_a_second_scene_does_not_reuse_the_first_schedule():
    from raytrace_amd import _lib
    lib = _lib.load()
    a = scenes.scan(level=4, dx=512, dy=384, spp=34, tex_size=128)
    b = scenes.charuco(dx=512, dy=384, spp=34)  # emitters only, calibrated camera: no culling, other shapes
    with phys.DeviceScene(a) as da, phys.DeviceScene(b) as db:
        fa = _film(lib, lib.rt_render_film_noalloc, da, a, 512, 384, 34)
        fb = _film(lib, lib.rt_render_film_noalloc, db, b, 512, 384, 34)
        _same(fa, _film(lib, lib.rt_render_film, da, a, 512, 384, 34))
        _same(fb, _film(lib, lib.rt_render_film, db, b, 512, 384, 34))


def test_no_alloc_render_rejects_what_does_not_fit():
    import torch
    from raytrace_amd import _lib
    lib = _lib.load()
    lit = scenes.scan(level=3, dx=64, dy=64, spp=1, tex_size=64, lit=True)
    film = torch.zeros((64, 64, 3), dtype=torch.float64, device="cuda")
    with phys.DeviceScene(lit) as dev:
        cd = phys.camera_desc(lit.Camera[0])
        ro = phys.render_opts(lit.RenderOptions)
        rc = lib.rt_render_film_noalloc(dev.handle, C.byref(cd), C.byref(ro), C.c_void_p(film.data_ptr()), None)
        assert rc == abi.RT_ERR_UNSUPPORTED and b"scatters rays" in lib.rt_last_error()
    emit = scenes.scan(level=3, dx=64, dy=64, spp=1, tex_size=64)
    with phys.DeviceScene(emit) as dev:
        cd = phys.camera_desc(emit.Camera[0])
        ro = phys.render_opts(phys.RenderOptions(0, 1, 5, 4160, 4096))  # 66 x 64 groups of 64 tiles: above the limit
        rc = lib.rt_render_film_noalloc(dev.handle, C.byref(cd), C.byref(ro), C.c_void_p(film.data_ptr()), None)
        assert rc == abi.RT_ERR_UNSUPPORTED and b"too large for the static scratch" in lib.rt_last_error()
        assert np.count_nonzero(film.cpu().numpy()) == 0  # rejected before anything ran

"""rt_render_frames: the frame loop of example/charuco (main.go:259-339) behind the boundary.  Every frame must be the
bytes and the statistics rt_render gives for that camera, with and without the overlap of copy-out and trace."""
import ctypes as C

import numpy as np
import pytest

from raytrace_amd import abi, phys, scenes

pytestmark = pytest.mark.gpu


def _stats(s):
    return (s.TotalRays, s.RaysLeftScene, s.RaysExceededDepth, s.ShadowRays)


@pytest.mark.parametrize("name", ["charuco", "cornell", "scan_lit"])
def test_frames_equal_single_renders(name):
    n = 7
    if name == "charuco":
        base = scenes.charuco(dx=360, dy=270, spp=2)
        cams = [scenes.charuco(dx=8, dy=8, spp=1, frame=f, frames_per_orbit=n).Camera[0] for f in range(n)]
    elif name == "cornell":
        base = scenes.cornell(dx=96, dy=80, spp=8)
        c0 = base.Camera[0]
        assert isinstance(c0, phys.PinholeCamera)
        step = 0.01 * np.linalg.norm(np.asarray(c0.Horizontal))
        move = lambda p, f: tuple(np.asarray(p, dtype=np.float64) + step * f)
        cams = [phys.PinholeCamera(move(c0.LowerLeftCorner, f), move(c0.Origin, f), c0.Horizontal, c0.Vertical) for f in range(n)]
    else:
        base = scenes.scan(level=5, dx=160, dy=96, spp=4, tex_size=64, lit=True)
        c0 = base.Camera[0]
        cams = [type(c0)(**{**c0.__dict__, "FOVHeight": c0.FOVHeight * (1 + 0.05 * f), "FOVWidth": c0.FOVWidth * (1 + 0.05 * f)}) for f in range(n)]
    o = base.RenderOptions
    with phys.DeviceScene(base) as dev:
        single = [dev.render(c, o) for c in cams]
        for overlap in (False, True):
            imgs, st = dev.render_frames(cams, o, overlap=overlap)
            for i in range(n):
                assert np.array_equal(imgs[i], single[i].Image), (name, overlap, i)
                assert _stats(st[i]) == _stats(single[i].Stats), (name, overlap, i)
        assert len({single[i].Image.tobytes() for i in range(n)}) == n  # the cameras really differ
        one, st1 = dev.render_frames(cams[3:4], o, overlap=True)
        assert np.array_equal(one[0], single[3].Image)
        none, st0 = dev.render_frames([], o)
        assert none.shape == (0, o.Dy, o.Dx, 4) and st0 == []


def test_frames_with_strides_and_errors():
    from raytrace_amd import _lib
    base = scenes.charuco(dx=100, dy=60, spp=1)
    cams = [scenes.charuco(dx=8, dy=8, spp=1, frame=f, frames_per_orbit=5).Camera[0] for f in range(5)]
    o = base.RenderOptions
    with phys.DeviceScene(base) as dev:
        lib = dev._lib
        single = [dev.render(c, o).Image for c in cams]
        descs = (abi.CameraDesc * 5)(*[phys.camera_desc(c) for c in cams])
        ro = phys.render_opts(o)
        stride, frame_stride = 100 * 4 + 24, (100 * 4 + 24) * 60 + 1000
        buf = np.full(5 * frame_stride, 0xAB, dtype=np.uint8)
        ptr = buf.ctypes.data_as(C.POINTER(C.c_uint8))
        _lib.check(lib.rt_render_frames(dev.handle, descs, 5, C.byref(ro), ptr, stride, frame_stride, None, 1), lib)
        for i in range(5):
            fr = buf[i * frame_stride:(i + 1) * frame_stride]
            rows = fr[:stride * 60].reshape(60, stride)
            assert np.array_equal(rows[:, :400].reshape(60, 100, 4), single[i])
            assert (rows[:, 400:] == 0xAB).all() and (fr[stride * 60:] == 0xAB).all()  # padding untouched
        assert lib.rt_render_frames(dev.handle, descs, 5, C.byref(ro), ptr, 399, frame_stride, None, 1) == abi.RT_ERR_INVALID
        assert lib.rt_render_frames(dev.handle, descs, 5, C.byref(ro), ptr, stride, stride * 59, None, 1) == abi.RT_ERR_INVALID
        assert lib.rt_render_frames(dev.handle, descs, -1, C.byref(ro), ptr, stride, frame_stride, None, 1) == abi.RT_ERR_INVALID
        bad = phys.render_opts(phys.RenderOptions(0, 0, 5, 100, 60))  # RaysPerPixel 0: the render's own validation, third frame on
        assert lib.rt_render_frames(dev.handle, descs, 5, C.byref(bad), ptr, stride, frame_stride, None, 1) == abi.RT_ERR_INVALID
        assert b"RaysPerPixel" in lib.rt_last_error()
        # the scene is still usable
        assert np.array_equal(dev.render(cams[2], o).Image, single[2])

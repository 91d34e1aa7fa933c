"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE.

Imported only by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg
of bench.py.  The scene structures are the ones of include/rt_hip.h
(raytrace_amd.abi), so the oracle and the HIP library see identical inputs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytrace_amd import abi

_DIR = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_DIR, "liboracle.so")
_LIB = None


class RenderExtra(C.Structure):
    _fields_ = [("n_threads", C.c_int32), ("tile_step", C.c_int32), ("film", C.c_void_p),
                ("boxes_tested", C.c_uint64), ("prims_tested", C.c_uint64), ("mesh_rays", C.c_uint64),
                ("threads_used", C.c_int32), ("reserved", C.c_int32)]


def build():
    subprocess.run(["make", "-C", _DIR, "liboracle.so"], check=True, capture_output=True)


def load():
    global _LIB
    if _LIB is None:
        src = os.path.join(_DIR, "oracle.cpp")
        if not os.path.exists(_PATH) or os.path.getmtime(_PATH) < os.path.getmtime(src):
            build()
        lib = C.CDLL(_PATH)
        P = C.POINTER
        lib.oracle_last_error.restype = C.c_char_p
        lib.oracle_scene_create.argtypes = [P(abi.SceneDesc), P(C.c_void_p)]
        lib.oracle_scene_destroy.argtypes = [C.c_void_p]
        lib.oracle_scene_destroy.restype = None
        lib.oracle_scene_bvh_info.argtypes = [C.c_void_p, C.c_uint64, P(abi.BvhInfo)]
        lib.oracle_intersect_batch.argtypes = [C.c_void_p, P(C.c_double), P(C.c_double), C.c_uint64, C.c_double,
                                               C.c_double, P(abi.Hit), C.c_int]
        lib.oracle_cast_batch.argtypes = [P(abi.CameraDesc), P(C.c_double), P(C.c_uint32), C.c_uint64, C.c_int64,
                                          P(C.c_double), P(C.c_double)]
        lib.oracle_render.argtypes = [C.c_void_p, P(abi.CameraDesc), P(abi.RenderOpts), P(C.c_uint8), C.c_size_t,
                                      P(abi.Stats), P(RenderExtra)]
        lib.oracle_microfacet.argtypes = [C.c_int, C.c_double] + [P(C.c_double)] * 6
        lib.oracle_shade_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p]
        lib.oracle_r3.argtypes = [C.c_int, P(C.c_double), P(C.c_double), C.c_double, C.c_double, P(C.c_double)]
        lib.oracle_aabb_intersects.argtypes = [P(C.c_double)] * 4
        lib.oracle_aabb_hit.argtypes = [P(C.c_double)] * 4 + [C.c_double, C.c_double]
        lib.oracle_node_bounds.argtypes = [C.c_void_p, C.c_uint64, P(C.c_double)]
        lib.oracle_apply_filter.argtypes = [P(C.c_uint8), C.c_int32, C.c_int32, C.c_int, P(C.c_uint8)]
        lib.oracle_apply_filter.restype = C.c_int
        lib.oracle_texture_at.argtypes = [C.c_void_p, C.c_uint64, P(C.c_double), C.c_uint64, P(C.c_double)]
        lib.oracle_random_stream.argtypes = [C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, P(C.c_double)]
        lib.oracle_philox4x32_10.argtypes = [P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
        lib.oracle_philox4x32_10.restype = None
        lib.oracle_sample.argtypes = [C.c_int, C.c_int64, C.c_uint32, C.c_uint32, P(C.c_double), P(C.c_double)]
        _LIB = lib
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class OracleScene:
    def __init__(self, scene):
        from raytrace_amd.phys import FlatScene
        self.lib = load()
        self.flat = scene if isinstance(scene, FlatScene) else FlatScene(scene)
        self.handle = C.c_void_p()
        rc = self.lib.oracle_scene_create(C.byref(self.flat.desc), C.byref(self.handle))
        if rc != 0:
            raise RuntimeError(f"oracle_scene_create: {self.lib.oracle_last_error().decode()}")

    def close(self):
        if self.handle:
            self.lib.oracle_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def intersect(self, origins, directions, tmin=1e-6, tmax=1.7976931348623157e308, n_threads=0):
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        hits = np.zeros(len(o), dtype=abi.HIT_DTYPE)
        rc = self.lib.oracle_intersect_batch(self.handle, _dp(o), _dp(d), len(o), tmin, tmax,
                                             C.cast(hits.ctypes.data, C.POINTER(abi.Hit)), n_threads)
        assert rc == 0
        return hits

    def render(self, camera, options, n_threads=0, tile_step=1, want_film=False, band_rank=0, band_world=1):
        """Returns (rgba uint8 (Dy,Dx,4), abi.Stats, RenderExtra, film or None)."""
        from raytrace_amd.phys import camera_desc, render_opts
        img = np.zeros((options.Dy, options.Dx, 4), dtype=np.uint8)
        film = np.zeros((options.Dy, options.Dx, 3), dtype=np.float64) if want_film else None
        st = abi.Stats()
        ex = RenderExtra(n_threads, tile_step, film.ctypes.data if want_film else None)
        cd = camera_desc(camera)
        ro = render_opts(options, abi.MODE_REFERENCE, band_rank, band_world)
        rc = self.lib.oracle_render(self.handle, C.byref(cd), C.byref(ro), img.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    options.Dx * 4, C.byref(st), C.byref(ex))
        if rc != 0:
            raise RuntimeError(f"oracle_render: {self.lib.oracle_last_error().decode()}")
        return img, st, ex, film

    def shade(self, hits, seed=0):
        h = np.ascontiguousarray(hits, dtype=abi.SHADE_IN_DTYPE)
        out = np.zeros(len(h), dtype=abi.SHADE_OUT_DTYPE)
        assert self.lib.oracle_shade_batch(self.handle, h.ctypes.data, len(h), int(seed), out.ctypes.data) == 0
        return out

    def bvh_info(self, mesh_index=0):
        info = abi.BvhInfo()
        assert self.lib.oracle_scene_bvh_info(self.handle, mesh_index, C.byref(info)) == 0
        return info

    def node_bounds(self, node):
        out = (C.c_double * 6)()
        assert self.lib.oracle_node_bounds(self.handle, node, out) == 0
        return np.array(out[:3]), np.array(out[3:])

    def texture_at(self, tex, uv):
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        rgb = np.zeros((len(uv), 3))
        assert self.lib.oracle_texture_at(self.handle, tex, _dp(uv), len(uv), _dp(rgb)) == 0
        return rgb


def cast(camera, st, pixel_sample, seed=0):
    from raytrace_amd.phys import camera_desc
    lib = load()
    st = np.ascontiguousarray(st, dtype=np.float64).reshape(-1, 2)
    ps = np.ascontiguousarray(pixel_sample, dtype=np.uint32).reshape(-1, 2)
    o = np.zeros((len(st), 3))
    d = np.zeros((len(st), 3))
    cd = camera_desc(camera)
    assert lib.oracle_cast_batch(C.byref(cd), _dp(st), ps.ctypes.data_as(C.POINTER(C.c_uint32)), len(st), seed, _dp(o),
                                 _dp(d)) == 0
    return o, d


def microfacet(which, roughness, f0, a, b=(0, 0, 0), c=(0, 0, 0), d=(0, 0, 0)):
    out = (C.c_double * 3)()
    assert load().oracle_microfacet(which, roughness, _d3(f0), _d3(a), _d3(b), _d3(c), _d3(d), out) == 0
    return out[0] if which < 3 else np.array(out[:])


def r3(op, a, b=(0, 0, 0), s=0.0, t=0.0):
    out = (C.c_double * 3)()
    assert load().oracle_r3(op, _d3(a), _d3(b), s, t, out) == 0
    return out[0] if op in (5, 7) else np.array(out[:])


def aabb_intersects(amin, amax, bmin, bmax):
    return bool(load().oracle_aabb_intersects(_d3(amin), _d3(amax), _d3(bmin), _d3(bmax)))


def aabb_hit(bmin, bmax, o, d, tmin, tmax):
    return bool(load().oracle_aabb_hit(_d3(bmin), _d3(bmax), _d3(o), _d3(d), tmin, tmax))


def random_stream(seed, pixel, sample, path, n):
    out = np.zeros(n)
    assert load().oracle_random_stream(seed, pixel, sample, path, n, _dp(out)) == 0
    return out


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    out = (C.c_uint32 * 4)()
    load().oracle_philox4x32_10(c, k, out)
    return list(out)


def sample(which, seed, pixel, sample_idx, normal=(0, 0, 1)):
    out = (C.c_double * 3)()
    assert load().oracle_sample(which, seed, pixel, sample_idx, _d3(normal), out) == 0
    return np.array(out[:])


def apply_filter(rgba, filter_kind):
    """ApplySeparableFilterRGBA (reconstruct.go:95-167) on an (H, W, 4) uint8 image."""
    lib = load()
    src = np.ascontiguousarray(rgba, dtype=np.uint8)
    dst = np.zeros_like(src)
    rc = lib.oracle_apply_filter(src.ctypes.data_as(C.POINTER(C.c_uint8)), src.shape[1], src.shape[0], int(filter_kind),
                                 dst.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0
    return dst

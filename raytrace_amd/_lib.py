"""Loader for librt_hip.so (the C ABI of include/rt_hip.h).  There is no fallback:
if the library is missing or fails to load, every render call raises."""
import ctypes as C
import os

from . import abi

_LIB = None
_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librt_hip.so")

SYMBOLS = [
    "rt_apply_filter", "rt_init", "rt_shutdown", "rt_last_error", "rt_scene_create", "rt_scene_destroy", "rt_render", "rt_render_film",
    "rt_film_resolve", "rt_cancel", "rt_intersect_batch", "rt_cast_batch", "rt_scene_bvh_info", "rt_obj_scan_text", "rt_obj_scan_free",
    "rt_bvh_build", "rt_render_film_noalloc", "rt_microfacet_batch", "rt_shade_batch", "rt_scene_create_from_json", "rt_render_frames",
]


def lib_path() -> str:
    return _PATH


def open_library() -> C.CDLL:
    """dlopen the library and declare signatures; does not touch the GPU."""
    if not os.path.exists(_PATH):
        raise RuntimeError(f"{_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    # PyTorch wheels bundle their own libamdhip64 (same SONAME).  A process that also uses torch
    # (device tensors, torch.distributed / RCCL) must end up with ONE HIP runtime: importing
    # torch first makes this library bind to the copy torch loaded.  RT_NO_TORCH=1 skips this
    # and binds to the system ROCm runtime (what a Go / C host gets).
    if os.environ.get("RT_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(_PATH)
    P = C.POINTER
    lib.rt_init.argtypes = [P(C.c_int), C.c_int]
    lib.rt_init.restype = C.c_int
    lib.rt_shutdown.argtypes = []
    lib.rt_shutdown.restype = None
    lib.rt_last_error.argtypes = []
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_scene_create.argtypes = [P(abi.SceneDesc), P(C.c_void_p)]
    lib.rt_scene_create.restype = C.c_int
    lib.rt_scene_create_from_json.argtypes = [C.c_char_p, C.c_size_t, P(C.c_void_p), P(abi.JsonSceneInfo)]
    lib.rt_scene_create_from_json.restype = C.c_int
    lib.rt_scene_destroy.argtypes = [C.c_void_p]
    lib.rt_scene_destroy.restype = None
    lib.rt_render.argtypes = [C.c_void_p, P(abi.CameraDesc), P(abi.RenderOpts), P(C.c_uint8), C.c_size_t, P(abi.Stats)]
    lib.rt_render.restype = C.c_int
    lib.rt_render_frames.argtypes = [C.c_void_p, P(abi.CameraDesc), C.c_int32, P(abi.RenderOpts), P(C.c_uint8), C.c_size_t, C.c_size_t,
                                     P(abi.Stats), C.c_int32]
    lib.rt_render_frames.restype = C.c_int
    lib.rt_render_film.argtypes = [C.c_void_p, P(abi.CameraDesc), P(abi.RenderOpts), C.c_void_p, P(abi.Stats)]
    lib.rt_render_film.restype = C.c_int
    lib.rt_render_film_noalloc.argtypes = [C.c_void_p, P(abi.CameraDesc), P(abi.RenderOpts), C.c_void_p, P(abi.Stats)]
    lib.rt_render_film_noalloc.restype = C.c_int
    lib.rt_film_resolve.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, P(C.c_uint8), C.c_size_t, C.c_void_p]
    lib.rt_film_resolve.restype = C.c_int
    lib.rt_cancel.argtypes = [C.c_void_p]
    lib.rt_cancel.restype = None
    lib.rt_intersect_batch.argtypes = [C.c_void_p, P(C.c_double), P(C.c_double), C.c_uint64, C.c_double, C.c_double,
                                       C.c_int, P(abi.Hit)]
    lib.rt_intersect_batch.restype = C.c_int
    lib.rt_cast_batch.argtypes = [P(abi.CameraDesc), P(C.c_double), P(C.c_uint32), C.c_uint64, C.c_int64, P(C.c_double),
                                  P(C.c_double)]
    lib.rt_cast_batch.restype = C.c_int
    lib.rt_microfacet_batch.argtypes = [C.c_int, C.c_uint64, P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_double)]
    lib.rt_microfacet_batch.restype = C.c_int
    lib.rt_shade_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_int, C.c_void_p]
    lib.rt_shade_batch.restype = C.c_int
    lib.rt_apply_filter.argtypes = [P(C.c_uint8), C.c_size_t, C.c_int32, C.c_int32, C.c_int, P(C.c_uint8), C.c_size_t]
    lib.rt_apply_filter.restype = C.c_int
    lib.rt_scene_bvh_info.argtypes = [C.c_void_p, C.c_uint64, P(abi.BvhInfo)]
    lib.rt_scene_bvh_info.restype = C.c_int
    lib.rt_obj_scan_text.argtypes = [C.c_char_p, C.c_uint64, P(abi.ObjScan)]
    lib.rt_obj_scan_text.restype = C.c_int
    lib.rt_obj_scan_free.argtypes = [P(abi.ObjScan)]
    lib.rt_obj_scan_free.restype = None
    lib.rt_bvh_build.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, P(C.c_uint64)]
    lib.rt_bvh_build.restype = C.c_int
    return lib


def check(code: int, lib) -> None:
    if code != abi.RT_OK:
        msg = lib.rt_last_error()
        raise RuntimeError(f"librt_hip error {code}: {msg.decode() if msg else ''}")


def load(device: int = None) -> C.CDLL:
    """Open the library and initialise it on `device` (default: LOCAL_RANK or 0).  RT_DEVICES="0,1,..." selects
    several devices for this process instead: every scene is then replicated on all of them and a frame is divided
    among them inside the library (an ordinal may be repeated: see rt_init)."""
    global _LIB
    if _LIB is None:
        lib = open_library()
        if device is None and os.environ.get("RT_DEVICES"):
            devices = [int(x) for x in os.environ["RT_DEVICES"].split(",")]
        else:
            devices = [int(os.environ.get("LOCAL_RANK", "0")) if device is None else device]
        init(lib, devices)
        _LIB = lib
    return _LIB


def init(lib, devices) -> None:
    """rt_init(devices): may be called again to change the selection (scenes created before keep theirs)."""
    ids = (C.c_int * len(devices))(*devices)
    check(lib.rt_init(ids, len(devices)), lib)


_HOST_LIB = None


def host_library():
    """The library for its host-only entry points (rt_obj_scan_text): opened, not initialised, no GPU needed.
    None when it cannot be opened (not built, no HIP runtime on this machine)."""
    global _HOST_LIB
    if _LIB is not None:
        return _LIB
    if _HOST_LIB is None:
        try:
            _HOST_LIB = open_library()
        except (RuntimeError, OSError):
            _HOST_LIB = False
    return _HOST_LIB or None

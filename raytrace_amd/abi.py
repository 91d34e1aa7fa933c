"""ctypes mirror of include/rt_hip.h (struct layouts and enums only; no logic).

Both the HIP library (raytrace_amd/_lib.py) and the CPU oracle used by the tests
consume these exact structures, so a scene is flattened once and handed to both.
"""
import ctypes as C

RECON_BOX, RECON_TENT, RECON_MITCHELL = 0, 1, 2  # rt_recon_filter

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_UNSUPPORTED = -2
RT_ERR_DEVICE = -3
RT_ERR_CANCELLED = -4
RT_ERR_NOMEM = -5
RT_ERR_NOT_INITIALIZED = -6

# rt_shape_kind
SHAPE_SPHERE, SHAPE_QUAD, SHAPE_CYLINDER, SHAPE_TRIANGLE, SHAPE_MESH, SHAPE_TRANSFORMED = range(6)
# rt_uvmap_kind
UVMAP_EQUIRECT, UVMAP_EQUIRECT_SOUTH_UP, UVMAP_BOX = range(3)
# rt_triangle_kind
TRI_TRIANGLE, TRI_TRIANGLE_UV, TRI_FACE = range(3)
# rt_texture_kind
TEX_UNIFORM, TEX_CHECKERBOARD, TEX_IMAGE = range(3)
INTERP_NEAREST, INTERP_BILINEAR = range(2)
WRAP_REPEAT, WRAP_CLAMP = range(2)
IMG_NRGBA8, IMG_RGBA16 = range(2)
# rt_material_kind
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_EMITTER, MAT_DEBUG_UV, MAT_DEBUG_NORMAL = range(6)
# rt_camera_kind
CAM_PINHOLE, CAM_FOCUSABLE, CAM_ORTHOGRAPHIC, CAM_CALIBRATED = range(4)
# rt_mode
MODE_REFERENCE, MODE_FAST = range(2)
# rt_device_split
SPLIT_AUTO, SPLIT_ROWS, SPLIT_SAMPLES = range(3)

D3 = C.c_double * 3


class Sphere(C.Structure):
    _fields_ = [("center", D3), ("radius", C.c_double), ("uvmap", C.c_int32), ("reserved", C.c_int32)]


class Quad(C.Structure):
    _fields_ = [("center", D3), ("normal", D3), ("width", C.c_double), ("height", C.c_double)]


class Cylinder(C.Structure):
    _fields_ = [("origin", D3), ("direction", D3), ("radius", C.c_double), ("height", C.c_double)]


class Triangle(C.Structure):
    _fields_ = [("p", (C.c_double * 3) * 3), ("uv", (C.c_double * 2) * 3), ("normal", D3),
                ("kind", C.c_int32), ("reserved", C.c_int32)]


class Face(C.Structure):
    _fields_ = [("v", (C.c_double * 5) * 3)]


class Mesh(C.Structure):
    _fields_ = [("first_face", C.c_uint64), ("n_faces", C.c_uint64)]


class Transformed(C.Structure):
    _fields_ = [("child_kind", C.c_int32), ("child_index", C.c_int32), ("translation", D3),
                ("rotation", C.c_double * 9), ("scale", D3)]


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("format", C.c_int32), ("reserved", C.c_int32),
                ("pixels", C.c_void_p)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("odd", C.c_int32), ("even", C.c_int32), ("image", C.c_int32),
                ("interp", C.c_int32), ("wrap", C.c_int32), ("color", D3), ("frequency", C.c_double)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("albedo", D3), ("fuzz", C.c_double),
                ("ior_interior", C.c_double), ("ior_exterior", C.c_double), ("roughness", C.c_double)]


class Light(C.Structure):
    _fields_ = [("position", D3), ("intensity", D3)]


class Node(C.Structure):
    _fields_ = [("shape_kind", C.c_int32), ("shape_index", C.c_int32), ("material", C.c_int32),
                ("reserved", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("nodes", C.POINTER(Node)), ("n_nodes", C.c_uint64),
        ("spheres", C.POINTER(Sphere)), ("n_spheres", C.c_uint64),
        ("quads", C.POINTER(Quad)), ("n_quads", C.c_uint64),
        ("cylinders", C.POINTER(Cylinder)), ("n_cylinders", C.c_uint64),
        ("triangles", C.POINTER(Triangle)), ("n_triangles", C.c_uint64),
        ("meshes", C.POINTER(Mesh)), ("n_meshes", C.c_uint64),
        ("faces", C.POINTER(Face)), ("n_faces", C.c_uint64),
        ("transformed", C.POINTER(Transformed)), ("n_transformed", C.c_uint64),
        ("materials", C.POINTER(Material)), ("n_materials", C.c_uint64),
        ("textures", C.POINTER(Texture)), ("n_textures", C.c_uint64),
        ("images", C.POINTER(Image)), ("n_images", C.c_uint64),
        ("lights", C.POINTER(Light)), ("n_lights", C.c_uint64),
    ]


class CameraDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("reserved", C.c_int32),
        ("look_from", D3), ("look_at", D3), ("vup", D3),
        ("fov_height", C.c_double), ("fov_width", C.c_double),
        ("aperture", C.c_double), ("working_distance", C.c_double),
        ("lower_left_corner", D3), ("origin", D3), ("horizontal", D3), ("vertical", D3),
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double), ("p2", C.c_double),
        ("k3", C.c_double), ("k4", C.c_double), ("k5", C.c_double), ("k6", C.c_double),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("seed", C.c_int64), ("rays_per_pixel", C.c_int32), ("max_ray_depth", C.c_int32),
        ("dx", C.c_int32), ("dy", C.c_int32), ("mode", C.c_int32),
        ("band_rank", C.c_int32), ("band_world", C.c_int32), ("sample_begin", C.c_int32),
        ("sample_count", C.c_int32), ("device_split", C.c_int32), ("stream", C.c_void_p),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("total_rays", C.c_uint64), ("rays_exceeded_depth", C.c_uint64), ("rays_left_scene", C.c_uint64),
        ("shadow_rays", C.c_uint64), ("render_seconds", C.c_double), ("traverse_seconds", C.c_double),
        ("traverse_launches", C.c_uint64), ("traverse_rays", C.c_uint64), ("culled_rays", C.c_uint64),
    ]


class JsonSceneInfo(C.Structure):  # rt_json_scene_info
    _fields_ = [("opts", RenderOpts), ("n_cameras", C.c_uint64), ("n_nodes", C.c_uint64), ("n_lights", C.c_uint64),
                ("cameras", CameraDesc * 4)]


class Hit(C.Structure):
    _fields_ = [
        ("node", C.c_int32), ("prim", C.c_int32), ("t", C.c_double), ("at", D3), ("normal", D3),
        ("uv", C.c_double * 2), ("boxes_tested", C.c_uint32), ("prims_tested", C.c_uint32),
    ]


class BvhInfo(C.Structure):
    _fields_ = [
        ("n_faces", C.c_uint64), ("n_nodes", C.c_uint64), ("n_leaves", C.c_uint64),
        ("max_depth", C.c_uint32), ("max_leaf_size", C.c_uint32), ("build_seconds", C.c_double),
    ]


HIT_DTYPE = [
    ("node", "<i4"), ("prim", "<i4"), ("t", "<f8"), ("at", "<f8", (3,)), ("normal", "<f8", (3,)),
    ("uv", "<f8", (2,)), ("boxes_tested", "<u4"), ("prims_tested", "<u4"),
]


BVH_NODE_DTYPE = [  # rt_bvh_node
    ("bounds_min", "<f8", (3,)), ("bounds_max", "<f8", (3,)), ("right", "<i4"), ("first", "<i4"), ("count", "<i4"), ("pad", "<i4"),
]


class ObjEvent(C.Structure):  # rt_obj_event
    _fields_ = [("kind", C.c_int32), ("line", C.c_uint32), ("line_off", C.c_uint64), ("line_len", C.c_uint64),
                ("arg_off", C.c_uint64), ("arg_len", C.c_uint64)]


class ObjScan(C.Structure):  # rt_obj_scan
    _fields_ = [("n_vertices", C.c_uint64), ("n_texcoords", C.c_uint64), ("n_normals", C.c_uint64), ("n_faces", C.c_uint64),
                ("n_corners", C.c_uint64), ("n_events", C.c_uint64),
                ("vertices", C.POINTER(C.c_double)), ("texcoords", C.POINTER(C.c_double)), ("normals", C.POINTER(C.c_double)),
                ("corners", C.POINTER(C.c_int32)), ("face_start", C.POINTER(C.c_uint64)), ("face_event", C.POINTER(C.c_int32)),
                ("events", C.POINTER(ObjEvent)), ("fallback_line", C.c_uint64), ("owner", C.c_void_p)]


OBJ_USEMTL, OBJ_MTLLIB = 0, 1


class ShadeIn(C.Structure):  # rt_shade_in
    _fields_ = [("material", C.c_int32), ("node", C.c_int32), ("d", D3), ("radiance", D3), ("t", C.c_double), ("at", D3),
                ("normal", D3), ("uv", C.c_double * 2), ("pixel", C.c_uint32), ("sample", C.c_uint32), ("path", C.c_uint32),
                ("draws", C.c_uint32)]


class ShadeOut(C.Structure):  # rt_shade_out
    _fields_ = [("emission", D3), ("dir", D3 * 2), ("rad", D3 * 2), ("direct", D3), ("n_children", C.c_int32),
                ("draws_after", C.c_uint32), ("shadow_rays", C.c_uint32), ("reserved", C.c_uint32)]


SHADE_IN_DTYPE = [("material", "<i4"), ("node", "<i4"), ("d", "<f8", (3,)), ("radiance", "<f8", (3,)), ("t", "<f8"), ("at", "<f8", (3,)),
                  ("normal", "<f8", (3,)), ("uv", "<f8", (2,)), ("pixel", "<u4"), ("sample", "<u4"), ("path", "<u4"), ("draws", "<u4")]
SHADE_OUT_DTYPE = [("emission", "<f8", (3,)), ("dir", "<f8", (2, 3)), ("rad", "<f8", (2, 3)), ("direct", "<f8", (3,)),
                   ("n_children", "<i4"), ("draws_after", "<u4"), ("shadow_rays", "<u4"), ("reserved", "<u4")]
MF_D, MF_G1, MF_G, MF_F, MF_EVALUATE = range(5)  # rt_microfacet_fn

"""Host-side mirror of the reference's scene API for the render path.

The names, fields and validation errors follow package phys of the reference
(phys/scene.go, node.go, shape_*.go, material_*.go, texture_*.go, camera_*.go,
light.go, render.go) so that scenes and tests read like the reference's own.
Nothing here computes pixels: `Render` flattens the scene into the C structures
of include/rt_hip.h and calls librt_hip.so, exactly what the Go `phys.Render`
would do through cgo.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence

import numpy as np

from . import abi

# phys/distance.go:10-19 -- distances are float64 nanometres.
NM = Nanometer = 1.0
UM = Micrometer = 1000.0 * NM
MM = Millimeter = 1000.0 * UM
M = Meter = 1000.0 * MM

EPS = 1e-6  # phys/math_extra.go:6


def _v3(v) -> tuple:
    v = tuple(float(x) for x in v)
    if len(v) != 3:
        raise ValueError(f"expected 3 components, got {v}")
    return v


def _len(v) -> float:
    return math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])


def _bad(v) -> bool:
    return any(math.isnan(x) or math.isinf(x) for x in v)


# ---- shapes ----------------------------------------------------------------------


@dataclass
class Sphere:  # shape_sphere.go:245-269
    Center: Sequence[float]
    Radius: float
    UVMap: int = abi.UVMAP_EQUIRECT

    def Validate(self):
        if self.Radius <= 0:
            raise ValueError(f"invalid Sphere radius: {self.Radius} (has it been set?)")


@dataclass
class Quad:  # shape_quad.go:15-36
    Center: Sequence[float]
    Normal: Sequence[float]
    Width: float
    Height: float

    def Validate(self):
        if self.Width <= 0:
            raise ValueError(f"invalid Quad width: {self.Width} (has it been set?)")
        if self.Height <= 0:
            raise ValueError(f"invalid Quad height: {self.Height} (has it been set?)")
        n = _v3(self.Normal)
        if n == (0.0, 0.0, 0.0):
            raise ValueError(f"invalid Quad Normal: {n} (has it been set?)")
        if _len(n) != 1:
            raise ValueError(f"invalid Quad Normal should be a unit vector, got: {n}")


@dataclass
class Cylinder:  # shape_cylinder.go:13-36
    Origin: Sequence[float]
    Direction: Sequence[float]
    Radius: float
    Height: float

    def Validate(self):
        if self.Radius <= 0:
            raise ValueError(f"invalid radius: {self.Radius} (has it been set?)")
        if self.Height <= 0:
            raise ValueError(f"invalid height: {self.Height} (has it been set?)")
        d = _v3(self.Direction)
        if d == (0.0, 0.0, 0.0):
            raise ValueError(f"invalid direction: {d} (has it been set?)")
        if _len(d) < 1 - 1e-6 or _len(d) > 1 + 1e-6:
            raise ValueError(f"direction should be a unit vector, got: {d} which has length {_len(d)}")


def _tri_area_check(name, p0, p1, p2):
    if p0 == p1 or p0 == p2 or p1 == p2:
        raise ValueError(f"invalid {name}: two or more vertices are identical")
    e1 = [p1[i] - p0[i] for i in range(3)]
    e2 = [p2[i] - p0[i] for i in range(3)]
    cr = (e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0])
    if 0.5 * _len(cr) < 1e-12:
        raise ValueError(f"invalid {name}: triangle is degenerate (zero or near-zero area)")
    return e1, e2, cr


@dataclass
class Triangle:  # shape_triangle.go:43-86
    P0: Sequence[float]
    P1: Sequence[float]
    P2: Sequence[float]

    def Validate(self):
        _tri_area_check("Triangle", _v3(self.P0), _v3(self.P1), _v3(self.P2))


@dataclass
class TriangleUV:  # shape_triangle_uv.go:13-113
    P0: Sequence[float]
    P1: Sequence[float]
    P2: Sequence[float]
    UV0: Sequence[float]
    UV1: Sequence[float]
    UV2: Sequence[float]
    Normal: Sequence[float]

    def Validate(self):
        for lbl, p in (("P0", self.P0), ("P1", self.P1), ("P2", self.P2), ("UV0", self.UV0), ("UV1", self.UV1),
                       ("UV2", self.UV2)):
            if _bad([float(x) for x in p]):
                raise ValueError(f"invalid TriangleUV: {lbl} contains NaN or Inf: {p}")
        n = _v3(self.Normal)
        if n == (0.0, 0.0, 0.0):
            raise ValueError("invalid TriangleUV: normal is a zero vector")
        if _len(n) < 1 - EPS or _len(n) > 1 + EPS:
            raise ValueError(f"invalid TriangleUV: normal is not a unit vector (length: {_len(n)})")
        e1, e2, cr = _tri_area_check("TriangleUV", _v3(self.P0), _v3(self.P1), _v3(self.P2))
        d1 = sum(n[i] * e1[i] for i in range(3))
        d2 = sum(n[i] * e2[i] for i in range(3))
        if abs(d1) > EPS or abs(d2) > EPS:
            raise ValueError("invalid TriangleUV: normal is not orthogonal to triangle edges")
        l = _len(cr)
        if sum(n[i] * cr[i] / l for i in range(3)) < 1 - EPS:
            raise ValueError("invalid TriangleUV: normal does not match the direction of the cross product")
        ue1 = (self.UV1[0] - self.UV0[0], self.UV1[1] - self.UV0[1])
        ue2 = (self.UV2[0] - self.UV0[0], self.UV2[1] - self.UV0[1])
        if 0.5 * (ue1[0] * ue2[1] - ue1[1] * ue2[0]) < 1e-12:
            raise ValueError("invalid TriangleUV: UV coordinates form a degenerate triangle")


@dataclass
class Face:  # shape_mesh.go:12-91 -- a lone mesh face used as a shape
    Vertex: Any  # 3 x (x, y, z, u, v)

    def Validate(self):
        v = np.asarray(self.Vertex, dtype=np.float64).reshape(3, 5)
        if not np.isfinite(v).all():
            raise ValueError("invalid Face: vertex contains NaN or Inf")
        _tri_area_check("Face", tuple(v[0, :3]), tuple(v[1, :3]), tuple(v[2, :3]))


class Mesh:
    """phys.Mesh (shape_mesh.go:164-200).  `Face` is an (n, 3, 5) float64 array:
    per face three vertices of (x, y, z, u, v) -- the memory layout of Go's []Face.
    The BVH is built by the library when the scene is uploaded."""

    def __init__(self, faces):
        self.Face = np.ascontiguousarray(np.asarray(faces, dtype=np.float64).reshape(-1, 3, 5))

    def Validate(self):
        if len(self.Face) == 0:
            raise ValueError("Mesh must contain at least one face")
        if not np.isfinite(self.Face).all():
            raise ValueError("Mesh face is invalid: contains NaN or Inf")


def NewMesh(faces) -> Mesh:  # shape_mesh.go:170-181
    m = Mesh(faces)
    m.Validate()
    return m


@dataclass
class Transform:  # transform.go:53-66
    Translation: Sequence[float] = (0.0, 0.0, 0.0)
    Rotation: Any = ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))
    Scale: Sequence[float] = (1.0, 1.0, 1.0)


def NewTransform() -> Transform:
    return Transform()


def RotationMatrixX(a):  # r3/mat.go:57-67
    c, s = math.cos(a), math.sin(a)
    return ((1, 0, 0), (0, c, -s), (0, s, c))


def RotationMatrixY(a):  # r3/mat.go:69-79
    c, s = math.cos(a), math.sin(a)
    return ((c, 0, s), (0, 1, 0), (-s, 0, c))


def RotationMatrixZ(a):  # r3/mat.go:82-92
    c, s = math.cos(a), math.sin(a)
    return ((c, -s, 0), (s, c, 0), (0, 0, 1))


@dataclass
class TransformedShape:  # shape_transformed_shape.go:12-22
    Shape: Any
    Transform: Transform

    def Validate(self):
        if self.Shape is None:
            raise ValueError("TransformedShape: Shape is nil")
        self.Shape.Validate()


# ---- textures ----------------------------------------------------------------------


@dataclass
class TextureUniform:  # texture_uniform.go:13-24
    Color: Sequence[float]

    def Validate(self):
        pass


@dataclass
class TextureCheckerboard:  # texture_checkerboard.go:21-54
    Odd: Any
    Even: Any
    Frequency: float

    def Validate(self):
        if self.Odd is None:
            raise ValueError("error TextureCheckerboard.Odd Texture is nil")
        if self.Even is None:
            raise ValueError("error TextureCheckerboard.Even Texture is nil")
        if self.Frequency <= 0:
            raise ValueError(f"error TextureChecker.Frequency is negative: {self.Frequency}")


@dataclass
class TextureImage:  # texture_image.go:18-30
    """Image is an (H, W, 4) array: uint8 = non-premultiplied NRGBA as png.Decode
    returns it, uint16 = alpha-premultiplied RGBA as color.Color.RGBA() reports."""
    Image: Any
    Interp: str = ""
    WrapMode: str = ""
    FilePath: str = ""

    def Validate(self):
        if self.Image is None:
            raise ValueError("image texture is nil")


# ---- materials ----------------------------------------------------------------------


@dataclass
class Lambertian:  # material_lambertian.go:10-16
    Texture: Any

    def Validate(self):
        self.Texture.Validate()


@dataclass
class Metal:  # material_metal.go:13-26
    Albedo: Sequence[float]
    Fuzz: float = 0.0

    def Validate(self):
        if min(self.Albedo) < 0:
            raise ValueError("invalid Metal albedo must be positive")
        if self.Fuzz < 0 or self.Fuzz > 1:
            raise ValueError("invalid Metal fuzz must be in the range [0, 1]")


@dataclass
class Dielectric:  # material_dielectric.go:13-27
    RefractiveIndexInterior: float
    RefractiveIndexExterior: float
    Roughness: float = 0.0

    def Validate(self):
        if self.RefractiveIndexInterior < 1 or self.RefractiveIndexExterior < 1:
            raise ValueError(f"invalid Dielectric refractive index: {self}")
        if self.Roughness < 0 or self.Roughness > 1:
            raise ValueError(f"invalid Dielectric roughness: {self}")


@dataclass
class Emitter:  # material_emitter.go:14-20
    Texture: Any

    def Validate(self):
        self.Texture.Validate()


@dataclass
class DebugUV:  # material_debug_uv.go:11-15
    def Validate(self):
        pass


@dataclass
class DebugNormal:  # material_debug_normal.go:20-29
    def Validate(self):
        pass


@dataclass
class PointLight:  # light.go:16-26
    Position: Sequence[float]
    RadiantIntensity: Sequence[float]

    def Validate(self):
        if min(self.RadiantIntensity) < 0:
            raise ValueError(f"invalid PointLight RadiantIntensity: {self.RadiantIntensity} (should be non-negative)")


# ---- cameras ----------------------------------------------------------------------


def _basis_check(name, look_from, look_at, vup):
    f, a, u = _v3(look_from), _v3(look_at), _v3(vup)
    if any(math.isnan(x) for x in f + a + u):
        raise ValueError(f"{name} has NaN values")
    if any(math.isinf(x) for x in f + a + u):
        raise ValueError(f"{name} has Inf values")
    if f == a:
        raise ValueError(f"{name} LookFrom and LookAt points are the same")
    if u == (0.0, 0.0, 0.0):
        raise ValueError(f"{name} VUp vector is zero")
    w = np.subtract(f, a)
    w = w / np.linalg.norm(w)
    uu = np.cross(u, w)
    n = np.linalg.norm(uu)
    uu = uu / n if n else uu * 0
    vv = np.cross(w, uu)
    if abs(uu @ vv) > EPS or abs(uu @ w) > EPS or abs(vv @ w) > EPS or n == 0:
        raise ValueError(f"{name} basis vectors are not orthogonal")


@dataclass
class PinholeCamera:  # camera_pinhole.go:12-60
    LowerLeftCorner: Sequence[float]
    Origin: Sequence[float]
    Horizontal: Sequence[float]
    Vertical: Sequence[float]

    def Validate(self):
        if _bad(_v3(self.LowerLeftCorner) + _v3(self.Origin)):
            raise ValueError("PinholeCamera has NaN values")
        h, v = _v3(self.Horizontal), _v3(self.Vertical)
        if h == (0.0, 0.0, 0.0):
            raise ValueError("PinholeCamera Horizontal vector is zero")
        if v == (0.0, 0.0, 0.0):
            raise ValueError("PinholeCamera Vertical vector is zero")
        if _bad(h + v):
            raise ValueError("PinholeCamera has NaN in vectors")
        if not np.any(np.cross(h, v)):
            raise ValueError("PinholeCamera Horizontal and Vertical vectors are colinear")


@dataclass
class FocusableCamera:  # camera_focusable.go:14-99
    LookFrom: Sequence[float]
    LookAt: Sequence[float]
    VUp: Sequence[float]
    FOVHeight: float
    FOVWidth: float
    Aperture: float
    WorkingDistance: float

    def Validate(self):
        if self.FOVHeight <= 0 or self.FOVWidth <= 0:
            raise ValueError("FocusableCamera FOVHeight and FOVWidth must be positive")
        if self.WorkingDistance <= 0:
            raise ValueError("FocusableCamera WorkingDistance must be positive")
        if self.Aperture < 0:
            raise ValueError("FocusableCamera Aperture cannot be negative")
        _basis_check("FocusableCamera", self.LookFrom, self.LookAt, self.VUp)


@dataclass
class OrthographicCamera:  # camera_orthographic.go:14-74
    LookFrom: Sequence[float]
    LookAt: Sequence[float]
    VUp: Sequence[float]
    FOVHeight: float
    FOVWidth: float

    def Validate(self):
        if self.FOVHeight <= 0 or self.FOVWidth <= 0:
            raise ValueError("Cast FOVHeight and FOVWidth must be positive")
        _basis_check("Camera", self.LookFrom, self.LookAt, self.VUp)


@dataclass
class CameraIntrinsics:  # camera_calibrated.go:110-150
    Width: int
    Height: int
    Fx: float
    Fy: float
    Cx: float
    Cy: float
    K1: float = 0.0
    K2: float = 0.0
    P1: float = 0.0
    P2: float = 0.0
    K3: float = 0.0
    K4: float = 0.0
    K5: float = 0.0
    K6: float = 0.0

    def Validate(self):
        if self.Width <= 0 or self.Height <= 0:
            raise ValueError(f"CameraIntrinsics bad image size: {self.Width}x{self.Height}")
        if not (self.Fx > 0 and self.Fy > 0):
            raise ValueError(f"CameraIntrinsics bad focal lengths: Fx={self.Fx} Fy={self.Fy}")
        if math.isnan(self.Cx) or math.isnan(self.Cy):
            raise ValueError("CameraIntrinsics NaN principal point")


@dataclass
class CameraExtrinsics:  # camera_calibrated.go:237-266
    LookFrom: Sequence[float]
    LookAt: Sequence[float]
    VUp: Sequence[float]

    def Validate(self):
        _basis_check("CameraExtrinsics", self.LookFrom, self.LookAt, self.VUp)


@dataclass
class CalibratedCamera:  # camera_calibrated.go:32-90
    Intrinsics: CameraIntrinsics
    Extrinsics: CameraExtrinsics

    def Validate(self):
        self.Intrinsics.Validate()
        self.Extrinsics.Validate()


@dataclass
class AnimatedCamera:  # camera_animated.go:53-131
    """A camera factory over normalized progress U.  The reference builds the concrete camera inside every
    Cast call (a Go closure, camera_animated.go:70-74); since U is fixed for a frame the result is the same
    camera for every ray, so it is built once here, before the camera crosses the C ABI."""
    Build: Any          # callable u in [0, 1) -> camera
    U: float = 0.0
    Period: float = 0.0  # seconds per cycle; only WithTime / Advance use it

    def _wrap01(self) -> float:
        return self.U - math.floor(self.U)

    def Resolve(self):
        cam = self.Build(self._wrap01())
        return cam.Resolve() if isinstance(cam, AnimatedCamera) else cam

    def Validate(self):
        if self.Build is None:
            raise ValueError("AnimatedCamera.Build is nil")
        cam = self.Build(self._wrap01())
        if cam is None:
            raise ValueError(f"AnimatedCamera.Build({self._wrap01():g}) returned nil Camera")
        cam.Validate()

    def WithU(self, u: float) -> "AnimatedCamera":
        return AnimatedCamera(self.Build, float(u), self.Period)

    def WithTime(self, t: float) -> "AnimatedCamera":
        if self.Period == 0:
            raise ValueError("AnimatedCamera.WithTime: Period is zero")
        return self.WithU(t / self.Period)

    def Advance(self, dt: float) -> "AnimatedCamera":
        if self.Period == 0:
            raise ValueError("AnimatedCamera.Advance: Period is zero")
        return self.WithU(self.U + dt / self.Period)

    def Frames(self, n: int):
        if n <= 0 or self.Build is None:
            return None
        return [self.Build(i / n) for i in range(n)]


# camera_calibrated_prefab.go:4-47
IntrinsicsFireflyDLComputar16mm = CameraIntrinsics(1440, 1080, 4736.71083258, 4743.03975965, 770.21866744, 483.50827668,
                                                   -0.09261328, -1.43023836, -0.00215911, -0.00187976, 188.33757455)
IntrinsicsFireflyDLComputar12mm = CameraIntrinsics(1440, 1080, 3613.49651386, 3617.43390846, 837.17440873, 412.78087519,
                                                   -0.18060152, 2.48103332, -0.00571920, 0.00121639, -29.84726761)
IntrinsicsFireflyDLGeneric6mm = CameraIntrinsics(1440, 1080, 1804.17453167, 1804.69144616, 756.49974101, 481.63486915,
                                                 -0.50722235, 0.44907698, 0.00151234, -0.00094105, -0.72605770)


# ---- scene ----------------------------------------------------------------------------


@dataclass
class Node:  # node.go:21-26
    Name: str
    Shape: Any
    Material: Any

    def Validate(self):  # node.go:28-46
        if self.Shape is None:
            raise ValueError("node has no shape")
        if self.Material is None:
            raise ValueError("node has no material")
        self.Shape.Validate()
        self.Material.Validate()


@dataclass
class RenderOptions:  # render.go:73-98
    Seed: int = 0
    RaysPerPixel: int = 1
    MaxRayDepth: int = 5
    Dx: int = 0
    Dy: int = 0

    def Validate(self):
        if self.Seed < 0:
            raise ValueError(f"bad Seed must be non-negative but got {self.Seed}")
        if self.RaysPerPixel <= 0:
            raise ValueError(f"bad RaysPerPixel must be positive but got {self.RaysPerPixel}")
        if self.MaxRayDepth <= 0:
            raise ValueError(f"bad MaxRayDepth must be positive but got {self.MaxRayDepth}")
        if self.Dx <= 0:
            raise ValueError(f"bad Dx must be positive but got {self.Dx}")
        if self.Dy <= 0:
            raise ValueError(f"bad Dy must be positive but got {self.Dy}")


@dataclass
class Scene:  # scene.go:9-79
    Camera: List[Any] = field(default_factory=list)
    Node: List[Node] = field(default_factory=list)
    Light: List[Any] = field(default_factory=list)
    RenderOptions: RenderOptions = field(default_factory=RenderOptions)

    def Add(self, *nodes):
        self.Node.extend(nodes)

    def Validate(self):
        try:
            self.RenderOptions.Validate()
        except ValueError as e:
            raise ValueError(f"bad RenderOptions={self.RenderOptions} err={e}")
        if not self.Camera:
            raise ValueError("no cameras in the scene")
        names = set()
        for n in self.Node:
            try:
                n.Validate()
            except ValueError as e:
                raise ValueError(f"node {n.Name}: {e}")
            if n.Name in names:
                raise ValueError(f"duplicate node name: {n.Name}")
            names.add(n.Name)
        for i, c in enumerate(self.Camera):
            try:
                c.Validate()
            except ValueError as e:
                raise ValueError(f"bad camera {i}: {e}")
        for i, l in enumerate(self.Light):
            try:
                l.Validate()
            except ValueError as e:
                raise ValueError(f"bad light {i}: {e}")

    def MarshalJSON(self) -> str:  # scene.go:82-114
        from . import scene_json
        return scene_json.MarshalScene(self)

    @staticmethod
    def UnmarshalJSON(text, strict: bool = False, asset_dir=None) -> "Scene":  # scene.go:117-159
        from . import scene_json
        return scene_json.UnmarshalScene(text, strict, asset_dir)


@dataclass
class RenderStats:  # render.go:38-45
    RaysExceededDepth: int = 0
    RaysLeftScene: int = 0
    TotalRays: int = 0
    RenderTime: float = 0.0  # seconds
    Dx: int = 0
    Dy: int = 0
    ShadowRays: int = 0
    TraverseSeconds: float = 0.0
    TraverseLaunches: int = 0
    TraverseRays: int = 0
    CulledRays: int = 0  # camera rays counted (TotalRays, RaysLeftScene) without being generated: their tiles cannot see the scene


@dataclass
class RenderArtifact:  # render.go:101-104
    Image: np.ndarray  # (Dy, Dx, 4) uint8, row 0 = top (image.RGBA.Pix)
    Stats: RenderStats


# ---- flattening into the C description -------------------------------------------------


class FlatScene:
    """The arrays of rt_scene_desc for one Scene.  Keeps every buffer alive."""

    def __init__(self, scene: Scene):
        self.spheres: list = []
        self.quads: list = []
        self.cylinders: list = []
        self.triangles: list = []
        self.meshes: list = []
        self.face_blocks: list = []
        self.transformed: list = []
        self.materials: list = []
        self.textures: list = []
        self.images: list = []
        self._image_arrays: list = []
        self._n_faces = 0
        self._shape_ids: dict = {}
        self._tex_ids: dict = {}
        self._mat_ids: dict = {}
        nodes = []
        for n in scene.Node:
            kind, idx = self._shape(n.Shape)
            nodes.append(abi.Node(kind, idx, self._material(n.Material), 0))
        lights = [abi.Light(abi.D3(*_v3(l.Position)), abi.D3(*_v3(l.RadiantIntensity))) for l in scene.Light]
        self._keep = []

        def arr(ctype, items):
            a = (ctype * max(1, len(items)))(*items)
            self._keep.append(a)
            return C.cast(a, C.POINTER(ctype)), len(items)

        d = abi.SceneDesc()
        d.nodes, d.n_nodes = arr(abi.Node, nodes)
        d.spheres, d.n_spheres = arr(abi.Sphere, self.spheres)
        d.quads, d.n_quads = arr(abi.Quad, self.quads)
        d.cylinders, d.n_cylinders = arr(abi.Cylinder, self.cylinders)
        d.triangles, d.n_triangles = arr(abi.Triangle, self.triangles)
        d.meshes, d.n_meshes = arr(abi.Mesh, self.meshes)
        if self.face_blocks:
            self.faces = np.ascontiguousarray(np.concatenate(self.face_blocks, axis=0)) if len(self.face_blocks) > 1 \
                else self.face_blocks[0]
        else:
            self.faces = np.zeros((1, 3, 5))
        d.faces = C.cast(self.faces.ctypes.data, C.POINTER(abi.Face))
        d.n_faces = self._n_faces
        d.transformed, d.n_transformed = arr(abi.Transformed, self.transformed)
        d.materials, d.n_materials = arr(abi.Material, self.materials)
        d.textures, d.n_textures = arr(abi.Texture, self.textures)
        d.images, d.n_images = arr(abi.Image, self.images)
        d.lights, d.n_lights = arr(abi.Light, lights)
        self.desc = d

    def _shape(self, s):
        key = id(s)
        if key in self._shape_ids:
            return self._shape_ids[key]
        if isinstance(s, Sphere):
            self.spheres.append(abi.Sphere(abi.D3(*_v3(s.Center)), float(s.Radius), int(s.UVMap), 0))
            r = (abi.SHAPE_SPHERE, len(self.spheres) - 1)
        elif isinstance(s, Quad):
            self.quads.append(abi.Quad(abi.D3(*_v3(s.Center)), abi.D3(*_v3(s.Normal)), float(s.Width), float(s.Height)))
            r = (abi.SHAPE_QUAD, len(self.quads) - 1)
        elif isinstance(s, Cylinder):
            self.cylinders.append(abi.Cylinder(abi.D3(*_v3(s.Origin)), abi.D3(*_v3(s.Direction)), float(s.Radius),
                                               float(s.Height)))
            r = (abi.SHAPE_CYLINDER, len(self.cylinders) - 1)
        elif isinstance(s, (Triangle, TriangleUV, Face)):
            t = abi.Triangle()
            if isinstance(s, Face):
                v = np.asarray(s.Vertex, dtype=np.float64).reshape(3, 5)
                for i in range(3):
                    for k in range(3):
                        t.p[i][k] = v[i, k]
                    t.uv[i][0], t.uv[i][1] = v[i, 3], v[i, 4]
                t.kind = abi.TRI_FACE
            else:
                for i, p in enumerate((s.P0, s.P1, s.P2)):
                    for k in range(3):
                        t.p[i][k] = float(p[k])
                if isinstance(s, TriangleUV):
                    for i, q in enumerate((s.UV0, s.UV1, s.UV2)):
                        t.uv[i][0], t.uv[i][1] = float(q[0]), float(q[1])
                    for k in range(3):
                        t.normal[k] = float(s.Normal[k])
                    t.kind = abi.TRI_TRIANGLE_UV
                else:
                    t.kind = abi.TRI_TRIANGLE
            self.triangles.append(t)
            r = (abi.SHAPE_TRIANGLE, len(self.triangles) - 1)
        elif isinstance(s, Mesh):
            self.meshes.append(abi.Mesh(self._n_faces, len(s.Face)))
            self.face_blocks.append(s.Face)
            self._n_faces += len(s.Face)
            r = (abi.SHAPE_MESH, len(self.meshes) - 1)
        elif isinstance(s, TransformedShape):
            ck, ci = self._shape(s.Shape)
            rot = [float(x) for row in s.Transform.Rotation for x in row]
            self.transformed.append(abi.Transformed(ck, ci, abi.D3(*_v3(s.Transform.Translation)),
                                                    (C.c_double * 9)(*rot), abi.D3(*_v3(s.Transform.Scale))))
            r = (abi.SHAPE_TRANSFORMED, len(self.transformed) - 1)
        else:
            raise TypeError(f"unsupported shape type {type(s).__name__}")
        self._shape_ids[key] = r
        return r

    def _texture(self, t) -> int:
        key = id(t)
        if key in self._tex_ids:
            return self._tex_ids[key]
        x = abi.Texture()
        if isinstance(t, TextureUniform):
            x.kind = abi.TEX_UNIFORM
            x.color = abi.D3(*_v3(t.Color))
        elif isinstance(t, TextureCheckerboard):
            x.kind = abi.TEX_CHECKERBOARD
            x.odd = self._texture(t.Odd)
            x.even = self._texture(t.Even)
            x.frequency = float(t.Frequency)
        elif isinstance(t, TextureImage):
            img = np.ascontiguousarray(t.Image)
            if img.ndim != 3 or img.shape[2] != 4 or img.dtype not in (np.uint8, np.uint16):
                raise TypeError("TextureImage.Image must be (H, W, 4) uint8 or uint16")
            self._image_arrays.append(img)
            self.images.append(abi.Image(img.shape[1], img.shape[0],
                                         abi.IMG_NRGBA8 if img.dtype == np.uint8 else abi.IMG_RGBA16, 0,
                                         img.ctypes.data))
            x.kind = abi.TEX_IMAGE
            x.image = len(self.images) - 1
            x.interp = abi.INTERP_BILINEAR if t.Interp == "bilinear" else abi.INTERP_NEAREST
            x.wrap = abi.WRAP_CLAMP if t.WrapMode == "clamp" else abi.WRAP_REPEAT
        else:
            raise TypeError(f"unsupported texture type {type(t).__name__}")
        self.textures.append(x)
        self._tex_ids[key] = len(self.textures) - 1
        return self._tex_ids[key]

    def _material(self, m) -> int:
        key = id(m)
        if key in self._mat_ids:
            return self._mat_ids[key]
        x = abi.Material()
        x.texture = -1
        if isinstance(m, Lambertian):
            x.kind, x.texture = abi.MAT_LAMBERTIAN, self._texture(m.Texture)
        elif isinstance(m, Metal):
            x.kind, x.albedo, x.fuzz = abi.MAT_METAL, abi.D3(*_v3(m.Albedo)), float(m.Fuzz)
        elif isinstance(m, Dielectric):
            x.kind = abi.MAT_DIELECTRIC
            x.ior_interior, x.ior_exterior = float(m.RefractiveIndexInterior), float(m.RefractiveIndexExterior)
            x.roughness = float(m.Roughness)
        elif isinstance(m, Emitter):
            x.kind, x.texture = abi.MAT_EMITTER, self._texture(m.Texture)
        elif isinstance(m, DebugUV):
            x.kind = abi.MAT_DEBUG_UV
        elif isinstance(m, DebugNormal):
            x.kind = abi.MAT_DEBUG_NORMAL
        else:
            raise TypeError(f"unsupported material type {type(m).__name__}")
        self.materials.append(x)
        self._mat_ids[key] = len(self.materials) - 1
        return self._mat_ids[key]


def camera_desc(cam) -> abi.CameraDesc:
    if isinstance(cam, abi.CameraDesc):
        return cam
    if isinstance(cam, AnimatedCamera):
        cam = cam.Resolve()
    d = abi.CameraDesc()
    if isinstance(cam, PinholeCamera):
        d.kind = abi.CAM_PINHOLE
        d.lower_left_corner = abi.D3(*_v3(cam.LowerLeftCorner))
        d.origin = abi.D3(*_v3(cam.Origin))
        d.horizontal = abi.D3(*_v3(cam.Horizontal))
        d.vertical = abi.D3(*_v3(cam.Vertical))
        return d
    if isinstance(cam, CalibratedCamera):
        d.kind = abi.CAM_CALIBRATED
        ci, ce = cam.Intrinsics, cam.Extrinsics
        d.look_from, d.look_at, d.vup = abi.D3(*_v3(ce.LookFrom)), abi.D3(*_v3(ce.LookAt)), abi.D3(*_v3(ce.VUp))
        d.width, d.height = int(ci.Width), int(ci.Height)
        for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "k4", "k5", "k6"):
            setattr(d, k, float(getattr(ci, k.capitalize())))
        return d
    d.look_from, d.look_at, d.vup = abi.D3(*_v3(cam.LookFrom)), abi.D3(*_v3(cam.LookAt)), abi.D3(*_v3(cam.VUp))
    d.fov_height, d.fov_width = float(cam.FOVHeight), float(cam.FOVWidth)
    if isinstance(cam, FocusableCamera):
        d.kind = abi.CAM_FOCUSABLE
        d.aperture, d.working_distance = float(cam.Aperture), float(cam.WorkingDistance)
    elif isinstance(cam, OrthographicCamera):
        d.kind = abi.CAM_ORTHOGRAPHIC
    else:
        raise TypeError(f"unsupported camera type {type(cam).__name__}")
    return d


def render_opts(o: RenderOptions, mode=abi.MODE_FAST, band_rank=0, band_world=1, stream=None, sample_begin=0,
                sample_count=0, device_split=abi.SPLIT_AUTO) -> abi.RenderOpts:
    return abi.RenderOpts(int(o.Seed), int(o.RaysPerPixel), int(o.MaxRayDepth), int(o.Dx), int(o.Dy), int(mode),
                          int(band_rank), int(band_world), int(sample_begin), int(sample_count), int(device_split), stream)


# ---- the render entry point -------------------------------------------------------------


class DeviceScene:
    """A scene resident on the GPU (rt_scene_create .. rt_scene_destroy)."""

    def __init__(self, scene: Scene):
        from . import _lib
        self._lib = _lib.load()
        self.flat = FlatScene(scene)
        self.handle = C.c_void_p()
        _lib.check(self._lib.rt_scene_create(C.byref(self.flat.desc), C.byref(self.handle)), self._lib)

    @classmethod
    def from_json(cls, text):
        """rt_scene_create_from_json: the text of Scene.MarshalJSON (scene.go:82-114) read by the library itself.
        Returns (DeviceScene, cameras, RenderOptions): the cameras as abi.CameraDesc, which render() accepts."""
        from . import _lib
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self.flat = None
        self.handle = C.c_void_p()
        raw = text.encode() if isinstance(text, str) else bytes(text)
        info = abi.JsonSceneInfo()
        _lib.check(self._lib.rt_scene_create_from_json(raw, len(raw), C.byref(self.handle), C.byref(info)), self._lib)
        o = info.opts
        self.json_info = info
        cameras = [info.cameras[i] for i in range(min(int(info.n_cameras), len(info.cameras)))]
        return self, cameras, RenderOptions(o.seed, o.rays_per_pixel, o.max_ray_depth, o.dx, o.dy)

    def close(self):
        if self.handle:
            self._lib.rt_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, camera, options: RenderOptions, mode=abi.MODE_FAST, device_split=abi.SPLIT_AUTO) -> RenderArtifact:
        from . import _lib
        img = np.zeros((options.Dy, options.Dx, 4), dtype=np.uint8)
        st = abi.Stats()
        cd = camera_desc(camera)
        ro = render_opts(options, mode, device_split=device_split)
        _lib.check(self._lib.rt_render(self.handle, C.byref(cd), C.byref(ro), img.ctypes.data_as(C.POINTER(C.c_uint8)),
                                       options.Dx * 4, C.byref(st)), self._lib)
        return RenderArtifact(img, stats_from(st, options))

    def render_frames(self, cameras, options: RenderOptions, mode=abi.MODE_FAST, overlap=True, out=None):
        """rt_render_frames: one frame per camera over this resident scene (the loop of example/charuco/main.go:259-339).
        Returns ((n, Dy, Dx, 4) uint8, [RenderStats]); with overlap a frame copies out while the next one is traced."""
        from . import _lib
        n = len(cameras)
        descs = (abi.CameraDesc * max(1, n))(*[camera_desc(c) for c in cameras])
        imgs = np.zeros((n, options.Dy, options.Dx, 4), dtype=np.uint8) if out is None else out
        assert imgs.shape == (n, options.Dy, options.Dx, 4) and imgs.dtype == np.uint8 and imgs.flags.c_contiguous
        st = (abi.Stats * max(1, n))()
        ro = render_opts(options, mode)
        _lib.check(self._lib.rt_render_frames(self.handle, descs, n, C.byref(ro), imgs.ctypes.data_as(C.POINTER(C.c_uint8)),
                                              options.Dx * 4, options.Dx * 4 * options.Dy, st, 1 if overlap else 0), self._lib)
        return imgs, [stats_from(st[i], options) for i in range(n)]

    def intersect(self, origins, directions, tmin=EPS, tmax=1.7976931348623157e308, mode=abi.MODE_REFERENCE):
        from . import _lib
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        hits = np.zeros(len(o), dtype=abi.HIT_DTYPE)
        _lib.check(self._lib.rt_intersect_batch(self.handle, o.ctypes.data_as(C.POINTER(C.c_double)),
                                                d.ctypes.data_as(C.POINTER(C.c_double)), len(o), tmin, tmax, int(mode),
                                                C.cast(hits.ctypes.data, C.POINTER(abi.Hit))), self._lib)
        return hits

    def shade(self, hits: np.ndarray, seed=0, mode=abi.MODE_REFERENCE) -> np.ndarray:
        """Material.Resolve + ComputeDirectLighting for explicit hits (rt_shade_batch): hits is an array of
        abi.SHADE_IN_DTYPE, the result one of abi.SHADE_OUT_DTYPE."""
        from . import _lib
        h = np.ascontiguousarray(hits, dtype=abi.SHADE_IN_DTYPE)
        out = np.zeros(len(h), dtype=abi.SHADE_OUT_DTYPE)
        _lib.check(self._lib.rt_shade_batch(self.handle, h.ctypes.data, len(h), int(seed), int(mode), out.ctypes.data), self._lib)
        return out

    def bvh_info(self, mesh_index=0) -> abi.BvhInfo:
        from . import _lib
        info = abi.BvhInfo()
        _lib.check(self._lib.rt_scene_bvh_info(self.handle, mesh_index, C.byref(info)), self._lib)
        return info


def microfacet_batch(which: int, roughness, f0, args) -> np.ndarray:
    """MicrofacetBRDF.D / G1 / G / F / Evaluate on the device (rt_microfacet_batch).  args: (n, 4, 3) in the order of the
    Go signature (unused vectors zero); returns (n, 3), scalar results in column 0."""
    from . import _lib
    lib = _lib.load()
    a = np.ascontiguousarray(args, dtype=np.float64).reshape(-1, 4, 3)
    r = np.ascontiguousarray(np.broadcast_to(np.asarray(roughness, dtype=np.float64), (len(a),)))
    f = np.ascontiguousarray(np.broadcast_to(np.asarray(f0, dtype=np.float64), (len(a), 3)))
    out = np.zeros((len(a), 3))
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    _lib.check(lib.rt_microfacet_batch(int(which), len(a), dp(r), dp(f), dp(a), dp(out)), lib)
    return out


def stats_from(st: abi.Stats, options: RenderOptions) -> RenderStats:
    return RenderStats(st.rays_exceeded_depth, st.rays_left_scene, st.total_rays, st.render_seconds, options.Dx,
                       options.Dy, st.shadow_rays, st.traverse_seconds, st.traverse_launches, st.traverse_rays, st.culled_rays)


def Render(scene: Scene, mode=abi.MODE_FAST) -> RenderArtifact:
    """phys.Render (render.go:351-364): validate, take Camera[0], render."""
    try:
        scene.Validate()
    except ValueError as e:
        raise ValueError(f"invalid scene: {e}")
    try:
        with DeviceScene(scene) as dev:
            return dev.render(scene.Camera[0], scene.RenderOptions, mode)
    except RuntimeError as e:
        raise RuntimeError(f"failed to render scene: {e}")


# ---- reconstruction filters (reconstruct.go:16-193) ---------------------------------------------


@dataclass
class ReconFilter:  # reconstruct.go:18-27; the kernel itself lives in the library, selected by `kind`
    Name: str
    Radius: float
    kind: int


def BoxFilter() -> ReconFilter:  # reconstruct.go:31-43
    return ReconFilter("Box(0.5)", 0.5, abi.RECON_BOX)


def TentFilter() -> ReconFilter:  # reconstruct.go:47-60
    return ReconFilter("Tent(1)", 1.0, abi.RECON_TENT)


def MitchellNetravaliFilter() -> ReconFilter:  # reconstruct.go:66-87
    return ReconFilter("Mitchell-Netravali(B=1/3,C=1/3)", 2.0, abi.RECON_MITCHELL)


def ApplySeparableFilterRGBA(src: np.ndarray, f: ReconFilter) -> np.ndarray:
    """reconstruct.go:95-167 on the GPU: (H, W, 4) uint8 in, a new (H, W, 4) uint8 out."""
    from . import _lib
    lib = _lib.load()
    img = np.ascontiguousarray(src, dtype=np.uint8)
    if img.ndim != 3 or img.shape[2] != 4:
        raise ValueError("expected an (H, W, 4) uint8 image")
    out = np.empty_like(img)
    h, w = img.shape[:2]
    _lib.check(lib.rt_apply_filter(img.ctypes.data_as(C.POINTER(C.c_uint8)), w * 4, w, h, f.kind,
                                   out.ctypes.data_as(C.POINTER(C.c_uint8)), w * 4), lib)
    return out


# ---- OBJ -> nodes (node.go:101-309) and file textures (texture_image.go:33-84) ------------------


def NewTextureImageFS(fsys, file_path: str, interp: str = "", wrap_mode: str = "") -> TextureImage:  # texture_image.go:33-49
    from . import obj as _obj
    return TextureImage(_obj.decode_png(fsys.read(file_path)), interp, wrap_mode, file_path)


def NewTextureImage(file_path: str, interp: str = "", wrap_mode: str = "") -> TextureImage:  # texture_image.go:60-84
    from . import obj as _obj
    with open(file_path, "rb") as f:
        return TextureImage(_obj.decode_png(f.read()), interp, wrap_mode, file_path)


def _faces_of(src, faces) -> np.ndarray:
    """Fan triangulation + vertex lookup + Face.Validate filtering (node.go:121-150, 172-193): (n, 3, 5) float64."""
    pos = np.asarray(src.Vertices, dtype=np.float64).reshape(-1, 3)
    tex = np.asarray(src.TexCoords, dtype=np.float64).reshape(-1, 2)
    if isinstance(faces, tuple):  # (FaceTable, indices of the faces wanted or None): the arrays of the native scanner
        t = faces[0].fan(faces[1])
    else:
        tri = []
        for face in faces:
            idx = face.Indices
            if len(idx) < 3:
                continue
            for i in range(1, len(idx) - 1):
                tri.append((idx[0], idx[i], idx[i + 1]))
        t = np.asarray(tri, dtype=np.int64).reshape(-1, 3, 3)  # (n, 3 corners, (v, vt, vn))
    if len(t) == 0:
        return np.zeros((0, 3, 5))
    v = t[..., 0] - 1
    if (v < 0).any() or (v >= len(pos)).any():
        raise ValueError("vertex index out of range")
    out = np.zeros((len(t), 3, 5))
    out[..., :3] = pos[v]
    vt = t[..., 1]
    has = (vt > 0) & (vt - 1 < len(tex))  # no texture coordinate: uv = (0, 0)
    if len(tex):
        out[..., 3:] = np.where(has[..., None], tex[np.clip(vt - 1, 0, len(tex) - 1)], 0.0)
    # Face.Validate (shape_mesh.go:23-91): finite, no two identical vertices, area >= 1e-12; failing faces are skipped
    p0, p1, p2 = out[:, 0, :3], out[:, 1, :3], out[:, 2, :3]
    ok = np.isfinite(out).all(axis=(1, 2))
    ok &= ~((p0 == p1).all(1) | (p0 == p2).all(1) | (p1 == p2).all(1))
    with np.errstate(invalid="ignore", over="ignore"):
        area = 0.5 * np.linalg.norm(np.cross(p1 - p0, p2 - p0), axis=1)
    ok &= ~(area < 1e-12)
    return np.ascontiguousarray(out[ok])


def ConvertObjectToShape(src) -> Mesh:  # node.go:195-232
    from . import obj as _obj
    return Mesh(_faces_of(src, (src.Faces, None) if isinstance(src.Faces, _obj.FaceTable) else src.Faces))


def ConvertObjectToMaterial(src, asset_fs) -> dict:  # node.go:266-309
    materials = {}
    for name, mat in src.Materials.items():
        if mat.Texture != "":
            try:
                tex = NewTextureImageFS(asset_fs, mat.Texture, "bilinear", "repeat")
            except Exception as e:
                raise ValueError(f"failed to load texture '{mat.Texture}' for material '{name}': {e}")
        else:
            tex = TextureUniform(tuple(mat.Diffuse))
        m = Emitter(tex)
        m.Validate()
        materials[name] = m
    if not materials:
        materials["default"] = Emitter(TextureUniform((0.8, 0.8, 0.8)))
    return materials


def ConvertObjectToNodes(src, asset_fs) -> list:  # node.go:101-169
    """One Mesh node per material name.  The reference iterates a Go map, so its node order is unspecified;
    here nodes come in order of first use, which is stable."""
    materials = ConvertObjectToMaterial(src, asset_fs)
    from . import obj as _obj
    groups: dict = {}
    if isinstance(src.Faces, _obj.FaceTable):
        groups = {name: (src.Faces, sel) for name, sel in src.Faces.groups().items()}
    else:
        for face in src.Faces:
            groups.setdefault(face.Material or "default", []).append(face)
    nodes = []
    for name, faces in groups.items():
        mesh = NewMesh(_faces_of(src, faces))
        material = materials.get(name, materials.get("default"))
        node = Node(name, mesh, material)
        try:
            node.Validate()
        except (ValueError, AttributeError) as e:
            raise ValueError(f"invalid node for material '{name}': {e}")
        nodes.append(node)
    return nodes

"""The reference's JSON scene format: writer and reader for `phys.Scene`.

The format is what `encoding/json` produces from the reference's marshallers:

* every interface value held by the scene goes through `marshalInterface` (phys/json.go:52-79), which wraps
  the value's own encoding as `{"Data": ..., "Type": "<Go type name>"}`;
* `Scene.MarshalJSON` (scene.go:82-114) writes `Camera`, `Light`, `Node`, `RenderOptions`; `Node.MarshalJSON`
  (node.go:48-63) writes `Material`, `Name`, `Shape` (the node's Transform is not written);
* `BVH`, `Group`, `TextureCheckerboard`, `TextureUniform` and `Emitter` have marshallers of their own that put a
  second `"Type"` inside the data and wrap their interface fields again (shape_bvh.go:275-372,
  texture_checkerboard.go:77-135, texture_uniform.go:27-54, material_emitter.go:33-68);
* every other type is encoded field by field in declaration order; an interface field of such a type
  (`Lambertian.Texture`, `TransformedShape.Shape`, `TextureImage.Image`) is written as the bare encoding of the
  value it holds, without a type tag;
* maps are written with sorted keys, struct fields in declaration order, floats in the shortest form that
  round-trips ('f' notation, 'e' below 1e-6 and from 1e21, exponent without leading zero).

`MarshalScene` writes that text.  Go calls a marshaller with a pointer receiver (`BVH`, `Group`,
`TextureCheckerboard`, `Emitter`) only for values it reaches through a pointer; this mirror has no such
distinction and always writes the marshaller's form, the only one `UnmarshalJSON` of those types accepts.

`UnmarshalScene` reads it.  The reference's own reader (scene.go:117-159, json.go:82-100) resolves type names
through a registry (`RegisterInterfaceType`); `PinholeCamera`, `FocusableCamera` and `Triangle` are never
registered and an untagged interface field cannot be decoded by `encoding/json` at all, so the reference cannot
read back several things it writes.  `strict=True` reproduces that reader: the same registry, the same
messages for an unknown or mismatched type, an error for an untagged interface field.  The default reader
accepts everything the writer produces: all types, and untagged fields, whose type is recognised from their keys.

A mesh's `BVH` field is written from the tree the library builds for its faces (`rt_bvh_build`: the reference's
`NewBVH`); on reading, the faces are taken and the field is skipped, since the library builds that tree again
when the scene is uploaded.

Nothing in the reference's tests holds a scene file, so the format is taken from the marshallers alone.
"""
import base64
import json
import math
from decimal import Decimal

import numpy as np

from . import abi, phys

# Types the reference registers (the init functions of phys/*.go).
REGISTERED = frozenset([
    "Cylinder", "TextureImage", "TransformedShape", "CalibratedCamera", "TriangleUV", "Dielectric", "BVH", "Group",
    "DebugNormal", "Metal", "PointLight", "Sphere", "TextureUniform", "Quad", "Lambertian", "OrthographicCamera",
    "TextureCheckerboard", "Mesh", "Face", "DebugUV", "Emitter",
])
UNREGISTERED = frozenset(["PinholeCamera", "FocusableCamera", "Triangle"])


# ---- writing ------------------------------------------------------------------------------------------------------


def go_float(x) -> str:
    """A float64 as encoding/json writes it (floatEncoder in encode.go)."""
    f = float(x)
    if math.isnan(f) or math.isinf(f):
        raise ValueError(f"json: unsupported value: {'NaN' if math.isnan(f) else ('+Inf' if f > 0 else '-Inf')}")
    a = abs(f)
    if a != 0 and (a < 1e-6 or a >= 1e21):
        s = repr(f)  # exponent notation in both ranges, at least two exponent digits, like strconv's 'e'
        if len(s) >= 4 and s[-4:-1] == "e-0":  # encode.go cleans e-09 up to e-9
            s = s[:-2] + s[-1]
        return s
    s = format(Decimal(repr(f)), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    return s


def go_string(s: str) -> str:
    out = json.dumps(s, ensure_ascii=False)
    return (out.replace("<", "\\u003c").replace(">", "\\u003e").replace("&", "\\u0026")
            .replace("\u2028", "\\u2028").replace("\u2029", "\\u2029"))


def _obj(pairs) -> str:
    return "{" + ",".join(f'"{k}":{v}' for k, v in pairs) + "}"


def _arr(items) -> str:
    return "[" + ",".join(items) + "]"


def _xyz(v) -> str:
    return _obj(zip("XYZ", (go_float(c) for c in v)))


def _xy(v) -> str:
    return _obj(zip("XY", (go_float(c) for c in v)))


def _wrap(name: str, data: str) -> str:  # json.go:74-78, a map: keys sorted
    return _obj([("Data", data), ("Type", go_string(name))])


def _image(img) -> str:
    """image.Image as the struct behind it: *image.NRGBA for 8-bit arrays, *image.RGBA64 for 16-bit ones."""
    if img is None:
        return "null"
    a = np.asarray(img)
    if a.ndim != 3 or a.shape[2] != 4 or a.dtype not in (np.uint8, np.uint16):
        raise ValueError(f"TextureImage.Image: expected (H, W, 4) uint8 or uint16, got {a.dtype} {a.shape}")
    h, w = a.shape[:2]
    if a.dtype == np.uint8:
        pix, stride = np.ascontiguousarray(a).tobytes(), 4 * w
    else:
        pix, stride = np.ascontiguousarray(a.astype(">u2")).tobytes(), 8 * w
    rect = _obj([("Min", _obj([("X", "0"), ("Y", "0")])), ("Max", _obj([("X", str(w)), ("Y", str(h))]))])
    return _obj([("Pix", '"' + base64.b64encode(pix).decode() + '"'), ("Stride", str(stride)), ("Rect", rect)])


def _texture_data(t) -> str:
    if isinstance(t, phys.TextureUniform):
        return _obj([("Type", '"TextureUniform"'), ("Color", _xyz(t.Color))])
    if isinstance(t, phys.TextureCheckerboard):
        return _obj([("Type", '"TextureCheckerboard"'), ("Odd", _texture(t.Odd)), ("Even", _texture(t.Even)),
                     ("Frequency", go_float(t.Frequency))])
    if isinstance(t, phys.TextureImage):
        return _obj([("Image", _image(t.Image)), ("FilePath", go_string(t.FilePath)), ("Interp", go_string(t.Interp)),
                     ("WrapMode", go_string(t.WrapMode))])
    raise ValueError(f"cannot marshal texture {type(t).__name__}")


def _texture(t) -> str:
    if t is None:
        raise ValueError("cannot marshal a nil value")
    return _wrap(type(t).__name__, _texture_data(t))


def _material_data(m) -> str:
    if isinstance(m, phys.Emitter):
        return _obj([("Type", '"Emitter"'), ("Texture", _texture(m.Texture))])
    if isinstance(m, phys.Lambertian):
        return _obj([("Texture", "null" if m.Texture is None else _texture_data(m.Texture))])
    if isinstance(m, phys.Metal):
        return _obj([("Albedo", _xyz(m.Albedo)), ("Fuzz", go_float(m.Fuzz))])
    if isinstance(m, phys.Dielectric):
        return _obj([("RefractiveIndexInterior", go_float(m.RefractiveIndexInterior)),
                     ("RefractiveIndexExterior", go_float(m.RefractiveIndexExterior)), ("Roughness", go_float(m.Roughness))])
    if isinstance(m, (phys.DebugUV, phys.DebugNormal)):
        return "{}"
    raise ValueError(f"cannot marshal material {type(m).__name__}")


def _vertex(v) -> str:
    return _obj([("Position", _xyz(v[:3])), ("UV", _xy(v[3:5]))])


def _face_data(f) -> str:
    return _obj([("Vertex", _arr(_vertex(v) for v in f))])


def mesh_bvh(faces: np.ndarray):
    """(nodes, leaf_faces) of NewBVH over `faces` ((n, 3, 5) float64), from the library's host-side builder."""
    from . import _lib
    lib = _lib.host_library()
    if lib is None:
        raise RuntimeError("librt_hip.so is needed to write a mesh's BVH and could not be opened")
    faces = np.ascontiguousarray(faces, dtype=np.float64)
    n = len(faces)
    nodes = np.zeros(2 * n, dtype=abi.BVH_NODE_DTYPE)
    leaf = np.zeros(n, dtype=np.int32)
    import ctypes as C
    count = C.c_uint64(0)
    _lib.check(lib.rt_bvh_build(faces.ctypes.data, n, nodes.ctypes.data, len(nodes), leaf.ctypes.data, C.byref(count)), lib)
    return nodes[:count.value], leaf


def _bvh(faces, nodes, leaf, out: list) -> None:
    """(*BVH).MarshalJSON over the flat tree, written without recursion (trees of scans are deep and wide)."""
    face_json = {}

    def face(i):
        s = face_json.get(i)
        if s is None:
            s = face_json[i] = _wrap("Face", _face_data(faces[i]))
        return s

    # Work list of text pieces and node indices to expand.
    stack = [0]
    while stack:
        item = stack.pop()
        if isinstance(item, str):
            out.append(item)
            continue
        nd = nodes[item]
        bounds = _obj([("Min", _xyz(nd["bounds_min"])), ("Max", _xyz(nd["bounds_max"]))])
        if nd["right"] < 0:  # leaf: Left and Right are the same shape, written twice
            first, count = int(nd["first"]), int(nd["count"])
            if count == 1:
                child = face(int(leaf[first]))
            else:
                shapes = _arr(face(int(leaf[first + k])) for k in range(count))
                child = _wrap("Group", _obj([("Type", '"Group"'), ("Shapes", shapes)]))
            out.append('{"Type":"BVH","Left":' + child + ',"Right":' + child + ',"Bounds":' + bounds + "}")
            continue
        out.append('{"Type":"BVH","Left":{"Data":')
        stack.append("}")
        stack.append(',"Type":"BVH"},"Bounds":' + bounds)
        stack.append(int(nd["right"]))
        stack.append(',"Type":"BVH"},"Right":{"Data":')
        stack.append(item + 1)


def _shape_data(s) -> str:
    if isinstance(s, phys.Sphere):
        pairs = [("Center", _xyz(s.Center)), ("Radius", go_float(s.Radius))]
        if int(s.UVMap) != 0:  # `json:"UVMap,omitempty"`
            pairs.append(("UVMap", str(int(s.UVMap))))
        return _obj(pairs)
    if isinstance(s, phys.Quad):
        return _obj([("Center", _xyz(s.Center)), ("Normal", _xyz(s.Normal)), ("Width", go_float(s.Width)),
                     ("Height", go_float(s.Height))])
    if isinstance(s, phys.Cylinder):
        return _obj([("Origin", _xyz(s.Origin)), ("Direction", _xyz(s.Direction)), ("Radius", go_float(s.Radius)),
                     ("Height", go_float(s.Height))])
    if isinstance(s, phys.TriangleUV):
        return _obj([("P0", _xyz(s.P0)), ("P1", _xyz(s.P1)), ("P2", _xyz(s.P2)), ("UV0", _xy(s.UV0)), ("UV1", _xy(s.UV1)),
                     ("UV2", _xy(s.UV2)), ("Normal", _xyz(s.Normal))])
    if isinstance(s, phys.Triangle):
        return _obj([("P0", _xyz(s.P0)), ("P1", _xyz(s.P1)), ("P2", _xyz(s.P2))])
    if isinstance(s, phys.Face):
        return _face_data(np.asarray(s.Vertex, dtype=np.float64).reshape(3, 5))
    if isinstance(s, phys.Mesh):
        faces = s.Face
        out = ['{"Face":', _arr(_face_data(f) for f in faces), ',"BVH":']
        if len(faces) == 0:
            out.append("null")  # NewBVH returns nil for no shapes (shape_bvh.go:86-88)
        else:
            nodes, leaf = mesh_bvh(faces)
            _bvh(faces, nodes, leaf, out)
        out.append("}")
        return "".join(out)
    if isinstance(s, phys.TransformedShape):
        t = s.Transform
        rot = _obj([("M", _arr(_arr(go_float(c) for c in row) for row in t.Rotation))])
        tr = _obj([("Translation", _xyz(t.Translation)), ("Rotation", rot), ("Scale", _xyz(t.Scale))])
        return _obj([("Shape", "null" if s.Shape is None else _shape_data(s.Shape)), ("Transform", tr)])
    raise ValueError(f"cannot marshal shape {type(s).__name__}")


def _camera_data(c) -> str:
    if isinstance(c, phys.PinholeCamera):
        return _obj([("LowerLeftCorner", _xyz(c.LowerLeftCorner)), ("Origin", _xyz(c.Origin)),
                     ("Horizontal", _xyz(c.Horizontal)), ("Vertical", _xyz(c.Vertical))])
    if isinstance(c, phys.FocusableCamera):
        return _obj([("LookFrom", _xyz(c.LookFrom)), ("LookAt", _xyz(c.LookAt)), ("VUp", _xyz(c.VUp)),
                     ("FOVHeight", go_float(c.FOVHeight)), ("FOVWidth", go_float(c.FOVWidth)),
                     ("Aperture", go_float(c.Aperture)), ("WorkingDistance", go_float(c.WorkingDistance))])
    if isinstance(c, phys.OrthographicCamera):
        return _obj([("LookFrom", _xyz(c.LookFrom)), ("LookAt", _xyz(c.LookAt)), ("VUp", _xyz(c.VUp)),
                     ("FOVHeight", go_float(c.FOVHeight)), ("FOVWidth", go_float(c.FOVWidth))])
    if isinstance(c, phys.CalibratedCamera):
        i, e = c.Intrinsics, c.Extrinsics
        intr = _obj([("Width", str(int(i.Width))), ("Height", str(int(i.Height)))] +
                    [(k, go_float(getattr(i, k))) for k in ("Fx", "Fy", "Cx", "Cy", "K1", "K2", "P1", "P2", "K3", "K4", "K5", "K6")])
        extr = _obj([("LookFrom", _xyz(e.LookFrom)), ("LookAt", _xyz(e.LookAt)), ("VUp", _xyz(e.VUp))])
        return _obj([("Intrinsics", intr), ("Extrinsics", extr)])
    if isinstance(c, phys.AnimatedCamera):
        raise ValueError("json: unsupported type: func(float64) phys.Camera")  # AnimatedCamera.Build is a closure
    raise ValueError(f"cannot marshal camera {type(c).__name__}")


def _iface(v, data_of) -> str:
    if v is None:
        raise ValueError("cannot marshal a nil value")  # json.go:53-55
    return _wrap(type(v).__name__, data_of(v))


def MarshalNode(n) -> str:  # node.go:48-63
    return _obj([("Material", _iface(n.Material, _material_data)), ("Name", go_string(n.Name)), ("Shape", _iface(n.Shape, _shape_data))])


def MarshalScene(scene) -> str:  # scene.go:82-114
    o = scene.RenderOptions
    opts = _obj([("Seed", str(int(o.Seed))), ("RaysPerPixel", str(int(o.RaysPerPixel))), ("MaxRayDepth", str(int(o.MaxRayDepth))),
                 ("Dx", str(int(o.Dx))), ("Dy", str(int(o.Dy)))])
    light = lambda l: _obj([("Position", _xyz(l.Position)), ("RadiantIntensity", _xyz(l.RadiantIntensity))])
    return _obj([
        ("Camera", _arr(_iface(c, _camera_data) for c in scene.Camera)),
        ("Light", _arr(_iface(l, light) for l in scene.Light)),
        ("Node", _arr(MarshalNode(n) for n in scene.Node)),
        ("RenderOptions", opts),
    ])


# ---- reading ------------------------------------------------------------------------------------------------------


class _Reader:
    def __init__(self, strict: bool, asset_dir):
        self.strict = strict
        self.asset_dir = asset_dir

    # -- helpers --
    @staticmethod
    def num(d, key, default=0.0):
        v = d.get(key, default) if isinstance(d, dict) else default
        if v is None:
            return default
        if isinstance(v, bool) or not isinstance(v, (int, float)):
            raise ValueError(f"json: cannot unmarshal {type(v).__name__} into Go struct field .{key} of type float64")
        return float(v)

    @staticmethod
    def integer(d, key, default=0):
        v = d.get(key, default)
        if v is None:
            return default
        if isinstance(v, bool) or not isinstance(v, (int, float)) or int(v) != v:
            raise ValueError(f"json: cannot unmarshal {v!r} into Go struct field .{key} of type int")
        return int(v)

    def xyz(self, d, key):
        v = d.get(key) or {}
        return (self.num(v, "X"), self.num(v, "Y"), self.num(v, "Z"))

    def xy(self, d, key):
        v = d.get(key) or {}
        return (self.num(v, "X"), self.num(v, "Y"))

    @staticmethod
    def is_wrapper(raw) -> bool:
        return isinstance(raw, dict) and set(raw) <= {"Type", "Data"} and "Data" in raw

    def unwrap(self, raw):
        """unmarshalInterface (json.go:82-100): (type name, data)."""
        if not isinstance(raw, dict):
            raise ValueError(f"json: cannot unmarshal {type(raw).__name__} into Go value of type struct {{ Type string; Data json.RawMessage }}")
        name = raw.get("Type", "")
        if name not in REGISTERED and (self.strict or name not in UNREGISTERED):
            raise ValueError(f"unsupported type: `{name}`; has it been registered?")
        return name, raw.get("Data")

    def field(self, raw, guess, what):
        """An interface field: wrapped by a marshaller, or bare (default struct encoding), then typed by its keys."""
        if self.is_wrapper(raw):
            return self.unwrap(raw)
        if self.strict:
            raise ValueError(f"json: cannot unmarshal object into Go struct field of type phys.{what}")
        if not isinstance(raw, dict):
            raise ValueError(f"{what}: expected an object")
        if isinstance(raw.get("Type"), str):
            return raw["Type"], raw
        name = guess(raw)
        if name is None:
            raise ValueError(f"{what}: cannot tell the type of an untagged value with keys {sorted(raw)}")
        return name, raw

    @staticmethod
    def expect(data, name):
        got = data.get("Type", "") if isinstance(data, dict) else ""
        if got != name:
            raise ValueError(f"invalid type: expected {name}, got {got}")

    # -- textures --
    @staticmethod
    def guess_texture(d):
        if "Color" in d:
            return "TextureUniform"
        if "Odd" in d or "Even" in d:
            return "TextureCheckerboard"
        if "Image" in d or "FilePath" in d:
            return "TextureImage"
        return None

    def image(self, d):
        if d is None:
            return None
        if self.strict:
            raise ValueError("json: cannot unmarshal object into Go struct field TextureImage.Image of type image.Image")
        rect = d.get("Rect") or {}
        mn, mx = rect.get("Min") or {}, rect.get("Max") or {}
        w = self.integer(mx, "X") - self.integer(mn, "X")
        h = self.integer(mx, "Y") - self.integer(mn, "Y")
        stride = self.integer(d, "Stride")
        pix = base64.b64decode(d.get("Pix") or "")
        if w <= 0 or h <= 0 or stride < 4 * w or len(pix) < stride * (h - 1) + 4 * w:
            raise ValueError(f"TextureImage.Image: {len(pix)} bytes of pixels for {w}x{h}, stride {stride}")
        if stride >= 8 * w and len(pix) >= stride * (h - 1) + 8 * w:  # 16 bits per channel, big-endian
            rows = np.frombuffer(pix, dtype=np.uint8)[:stride * (h - 1) + 8 * w]
            a = np.lib.stride_tricks.as_strided(rows, shape=(h, 8 * w), strides=(stride, 1))
            return np.ascontiguousarray(a).view(">u2").astype(np.uint16).reshape(h, w, 4)
        rows = np.frombuffer(pix, dtype=np.uint8)[:stride * (h - 1) + 4 * w]
        a = np.lib.stride_tricks.as_strided(rows, shape=(h, 4 * w), strides=(stride, 1))
        return np.ascontiguousarray(a).reshape(h, w, 4)

    def texture(self, name, d):
        if name == "TextureUniform":
            self.expect(d, "TextureUniform")  # texture_uniform.go:49-51
            return phys.TextureUniform(self.xyz(d, "Color"))
        if name == "TextureCheckerboard":
            self.expect(d, "TextureCheckerboard")  # texture_checkerboard.go:120-122
            odd = self.texture(*self.unwrap(d.get("Odd")))
            even = self.texture(*self.unwrap(d.get("Even")))
            return phys.TextureCheckerboard(odd, even, self.num(d, "Frequency"))
        if name == "TextureImage":
            img = self.image(d.get("Image"))
            path = d.get("FilePath") or ""
            if img is None and path and not self.strict:  # the pixels were not carried: FilePath is there "for serialization"
                import os
                full = path if self.asset_dir is None else os.path.join(self.asset_dir, path)
                t = phys.NewTextureImage(full, d.get("Interp") or "", d.get("WrapMode") or "")
                t.FilePath = path
                return t
            return phys.TextureImage(img, d.get("Interp") or "", d.get("WrapMode") or "", path)
        raise ValueError(f"expected Texture, got {name}")

    # -- materials --
    @staticmethod
    def guess_material(d):
        if "Albedo" in d:
            return "Metal"
        if "RefractiveIndexInterior" in d:
            return "Dielectric"
        if "Texture" in d:
            return "Lambertian"
        return None

    def material(self, name, d):
        d = d if isinstance(d, dict) else {}
        if name == "Emitter":
            self.expect(d, "Emitter")  # material_emitter.go:59-61
            return phys.Emitter(self.texture(*self.unwrap(d.get("Texture"))))
        if name == "Lambertian":
            raw = d.get("Texture")
            if raw is None:
                return phys.Lambertian(None)
            return phys.Lambertian(self.texture(*self.field(raw, self.guess_texture, "Texture")))
        if name == "Metal":
            return phys.Metal(self.xyz(d, "Albedo"), self.num(d, "Fuzz"))
        if name == "Dielectric":
            return phys.Dielectric(self.num(d, "RefractiveIndexInterior"), self.num(d, "RefractiveIndexExterior"),
                                   self.num(d, "Roughness"))
        if name == "DebugUV":
            return phys.DebugUV()
        if name == "DebugNormal":
            return phys.DebugNormal()
        raise ValueError(f"expected Material, got *phys.{name}")  # node.go:89-92

    # -- shapes --
    @staticmethod
    def guess_shape(d):
        if "Face" in d:
            return "Mesh"
        if "Vertex" in d:
            return "Face"
        if "Shape" in d:
            return "TransformedShape"
        if "UV0" in d:
            return "TriangleUV"
        if "P0" in d:
            return "Triangle"
        if "Normal" in d:
            return "Quad"
        if "Direction" in d:
            return "Cylinder"
        if "Center" in d:
            return "Sphere"
        return None

    def face(self, d):
        vs = d.get("Vertex") or []
        out = np.zeros((3, 5))
        for i, v in enumerate(vs[:3]):
            out[i, :3] = self.xyz(v, "Position")
            out[i, 3:] = self.xy(v, "UV")
        return out

    def shape(self, name, d):
        d = d if isinstance(d, dict) else {}
        if name == "Sphere":
            return phys.Sphere(self.xyz(d, "Center"), self.num(d, "Radius"), self.integer(d, "UVMap"))
        if name == "Quad":
            return phys.Quad(self.xyz(d, "Center"), self.xyz(d, "Normal"), self.num(d, "Width"), self.num(d, "Height"))
        if name == "Cylinder":
            return phys.Cylinder(self.xyz(d, "Origin"), self.xyz(d, "Direction"), self.num(d, "Radius"), self.num(d, "Height"))
        if name == "Triangle":
            return phys.Triangle(self.xyz(d, "P0"), self.xyz(d, "P1"), self.xyz(d, "P2"))
        if name == "TriangleUV":
            return phys.TriangleUV(self.xyz(d, "P0"), self.xyz(d, "P1"), self.xyz(d, "P2"), self.xy(d, "UV0"), self.xy(d, "UV1"),
                                   self.xy(d, "UV2"), self.xyz(d, "Normal"))
        if name == "Face":
            return phys.Face(self.face(d))
        if name == "Mesh":
            faces = d.get("Face") or []
            arr = np.zeros((len(faces), 3, 5))
            for i, f in enumerate(faces):
                arr[i] = self.face(f)
            bvh = d.get("BVH")
            if bvh is not None:
                self.expect(bvh, "BVH")  # shape_bvh.go:311-313; the tree itself is rebuilt from the faces
            return phys.Mesh(arr)
        if name == "TransformedShape":
            raw = d.get("Shape")
            child = None if raw is None else self.shape(*self.field(raw, self.guess_shape, "Shape"))
            t = d.get("Transform") or {}
            m = (t.get("Rotation") or {}).get("M") or [[0.0] * 3] * 3
            rot = tuple(tuple(float(c) for c in row) for row in m)
            return phys.TransformedShape(child, phys.Transform(self.xyz(t, "Translation"), rot, self.xyz(t, "Scale")))
        if name in ("BVH", "Group"):
            raise ValueError(f"{name} as a node's shape is not supported: it is read only inside a Mesh")
        raise ValueError(f"expected Shape, got *phys.{name}")  # node.go:80-83

    # -- cameras, lights --
    def camera(self, name, d):
        d = d if isinstance(d, dict) else {}
        if name == "OrthographicCamera":
            return phys.OrthographicCamera(self.xyz(d, "LookFrom"), self.xyz(d, "LookAt"), self.xyz(d, "VUp"),
                                           self.num(d, "FOVHeight"), self.num(d, "FOVWidth"))
        if name == "CalibratedCamera":
            i, e = d.get("Intrinsics") or {}, d.get("Extrinsics") or {}
            intr = phys.CameraIntrinsics(self.integer(i, "Width"), self.integer(i, "Height"),
                                         *[self.num(i, k) for k in ("Fx", "Fy", "Cx", "Cy", "K1", "K2", "P1", "P2", "K3", "K4", "K5", "K6")])
            return phys.CalibratedCamera(intr, phys.CameraExtrinsics(self.xyz(e, "LookFrom"), self.xyz(e, "LookAt"), self.xyz(e, "VUp")))
        if name == "PinholeCamera":
            return phys.PinholeCamera(self.xyz(d, "LowerLeftCorner"), self.xyz(d, "Origin"), self.xyz(d, "Horizontal"),
                                      self.xyz(d, "Vertical"))
        if name == "FocusableCamera":
            return phys.FocusableCamera(self.xyz(d, "LookFrom"), self.xyz(d, "LookAt"), self.xyz(d, "VUp"), self.num(d, "FOVHeight"),
                                        self.num(d, "FOVWidth"), self.num(d, "Aperture"), self.num(d, "WorkingDistance"))
        raise ValueError(f"expected Camera, got *phys.{name}")

    def light(self, name, d):
        if name == "PointLight":
            return phys.PointLight(self.xyz(d or {}, "Position"), self.xyz(d or {}, "RadiantIntensity"))
        raise ValueError(f"expected Light, got *phys.{name}")

    def node(self, raw):  # node.go:66-97
        if not isinstance(raw, dict):
            raise ValueError("json: cannot unmarshal a node that is not an object")
        shape = self.shape(*self.unwrap(raw.get("Shape")))
        material = self.material(*self.unwrap(raw.get("Material")))
        name = raw.get("Name", "")
        if not isinstance(name, str):
            raise ValueError("json: cannot unmarshal Node.Name: not a string")
        return phys.Node(name, shape, material)

    def scene(self, raw):  # scene.go:117-159
        if not isinstance(raw, dict):
            raise ValueError("json: cannot unmarshal a scene that is not an object")
        s = phys.Scene()
        for c in raw.get("Camera") or []:
            s.Camera.append(self.camera(*self.unwrap(c)))
        for n in raw.get("Node") or []:
            s.Node.append(self.node(n))
        for l in raw.get("Light") or []:
            s.Light.append(self.light(*self.unwrap(l)))
        o = raw.get("RenderOptions") or {}
        s.RenderOptions = phys.RenderOptions(self.integer(o, "Seed"), self.integer(o, "RaysPerPixel"), self.integer(o, "MaxRayDepth"),
                                             self.integer(o, "Dx"), self.integer(o, "Dy"))
        return s


def _loads(text):
    # "-0" is a float64 negative zero written without a fraction; json would make it the integer 0
    return json.loads(text, parse_int=lambda s: -0.0 if s == "-0" else int(s))


def UnmarshalScene(text, strict: bool = False, asset_dir=None):
    """A `phys.Scene` from the reference's JSON (str or bytes).  Like Go's json.Unmarshal it does not validate the
    scene; `phys.Render` does (render.go:351-358).  `asset_dir`: where a TextureImage written without pixels
    finds its FilePath."""
    try:
        raw = _loads(text)
    except json.JSONDecodeError as e:
        raise ValueError(f"invalid character in scene JSON: {e}") from None
    return _Reader(strict, asset_dir).scene(raw)


def UnmarshalNode(text, strict: bool = False, asset_dir=None):
    return _Reader(strict, asset_dir).node(_loads(text))

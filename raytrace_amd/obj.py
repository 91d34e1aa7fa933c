"""Wavefront OBJ / MTL reader with the semantics of the reference's obj package (obj/obj.go), the converters of
phys/node.go:101-309 that turn a parsed object into Mesh nodes, and a PNG decoder for `map_Kd` textures.

A "file system" is anything with `read(name) -> bytes`: `DirFS(path)` for a directory, `MapFS({...})` for
in-memory files (what the reference's tests use, obj_test.go:12-190).  As in the reference, only the base name
of a path is looked up (obj.go:174-176, 458-461).
"""
from __future__ import annotations

import os
import posixpath
import struct
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


# ---- file systems -------------------------------------------------------------------------


class MapFS:
    def __init__(self, files: Dict[str, bytes]):
        self.files = {k: (v.encode() if isinstance(v, str) else bytes(v)) for k, v in files.items()}

    def read(self, name: str) -> bytes:
        if name not in self.files:
            raise FileNotFoundError(name)
        return self.files[name]


class DirFS:
    def __init__(self, root: str):
        self.root = root

    def read(self, name: str) -> bytes:
        with open(os.path.join(self.root, name), "rb") as f:
            return f.read()


# ---- parsed object (obj.go:109-155) -----------------------------------------------------------


@dataclass
class Material:  # obj.go:138-145
    Name: str
    Diffuse: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    Ambient: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    Specular: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    Shininess: float = 0.0
    Texture: str = ""


@dataclass
class Face:  # obj.go:132-135; Indices holds (vertex, texcoord, normal), 1-based, 0 = not given
    Indices: List[Tuple[int, int, int]]
    Material: str = ""


@dataclass
class Object:  # obj.go:148-154
    Vertices: List[Tuple[float, float, float]] = field(default_factory=list)
    TexCoords: List[Tuple[float, float]] = field(default_factory=list)
    Normals: List[Tuple[float, float, float]] = field(default_factory=list)
    Faces: List[Face] = field(default_factory=list)
    Materials: Dict[str, Material] = field(default_factory=dict)


class FaceTable:
    """The faces of a large object as arrays instead of a list of Face objects: what the native scanner
    (rt_obj_scan_text) returns.  Reads like the list it stands for (len, indexing, iteration give Face objects, made on
    demand); the converters of phys.py use the arrays directly."""

    def __init__(self, corners: np.ndarray, start: np.ndarray, material_id: np.ndarray, material_names: List[str]):
        self.corners = corners          # (n_corners, 3) int32: vertex, texcoord, normal index, 1-based, 0 = not given
        self.start = start              # (n_faces + 1,) int64 offsets into corners
        self.material_id = material_id  # (n_faces,) int32 index into material_names, -1 = ""
        self.material_names = material_names

    def __len__(self):
        return len(self.material_id)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        c = self.corners[self.start[i]:self.start[i + 1]]
        m = int(self.material_id[i])
        return Face([(int(a), int(b), int(d)) for a, b, d in c], self.material_names[m] if m >= 0 else "")

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __eq__(self, other):
        return list(self) == list(other)

    def groups(self) -> Dict[str, np.ndarray]:
        """Material name ("default" for none) -> indices of its faces, names in order of first use."""
        out: Dict[str, np.ndarray] = {}
        if len(self) == 0:
            return out
        ids, first = np.unique(self.material_id, return_index=True)
        for m in ids[np.argsort(first)]:
            name = (self.material_names[int(m)] if m >= 0 else "") or "default"
            sel = np.nonzero(self.material_id == m)[0]
            out[name] = np.concatenate([out[name], sel]) if name in out else sel
        for name in out:
            out[name] = np.sort(out[name])
        return out

    def fan(self, faces: np.ndarray = None) -> np.ndarray:
        """(n_triangles, 3, 3) indices: every face (or the selected ones) cut into a fan around its first corner."""
        start = self.start[:-1] if faces is None else self.start[faces]
        count = (self.start[1:] - self.start[:-1]) if faces is None else (self.start[faces + 1] - self.start[faces])
        ntri = np.maximum(count - 2, 0)
        total = int(ntri.sum())
        if total == 0:
            return np.zeros((0, 3, 3), dtype=np.int64)
        face_of = np.repeat(np.arange(len(start)), ntri)
        k = np.arange(total) - np.repeat(np.cumsum(ntri) - ntri, ntri)  # triangle number inside its face
        base = start[face_of]
        idx = np.stack([base, base + k + 1, base + k + 2], axis=1)
        return self.corners[idx].astype(np.int64)


class ParseError(ValueError):  # obj.go:157-171
    def __init__(self, filename, line, line_text, msg):
        self.Filename, self.Line, self.LineText, self.Msg = filename, line, line_text, msg
        where = f"{filename}:{line}" if filename else f"line {line}"
        super().__init__(f"{where}: {msg}\n    {line_text}")


def _split_fields(s: str, limit: int) -> List[str]:  # obj.go:664-669
    return s.split(" ", limit - 1) if limit > 0 else s.split()


def _resolve_index(val: int, size: int) -> int:  # obj.go:431-439
    if val < 0:
        val = size + val + 1
    if val < 1 or val > size:
        raise ValueError(f"index {val} out of range (1 to {size})")
    return val


def ParseFS(fsys, pattern: str) -> Object:  # obj.go:174-195
    name = posixpath.basename(pattern)
    try:
        data = fsys.read(name)
    except OSError as e:
        raise ParseError(name, 0, "", f"failed to read file '{pattern}': {e}")
    fast = _scan_native(data, fsys, name)
    if fast is not None:
        return fast
    obj = Object()
    current_material = ""
    text = data.decode("utf-8", errors="replace")
    for line_no, raw in enumerate(text.split("\n"), start=1):
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        sp = line.find(" ")
        if sp < 0:
            continue  # a lone token: ignored (obj.go:242-246)
        directive, rest = line[:sp], line[sp + 1:]

        def err(msg):
            return ParseError(name, line_no, line, msg)

        if directive == "v" or directive == "vn":  # obj.go:270-290, 312-332
            what = "vertex" if directive == "v" else "normal"
            f = _split_fields(rest, 3)
            if len(f) < 3:
                raise err(f"invalid {what} data: expected at least 3 components, got {len(f)}")
            try:
                xyz = (float(f[0]), float(f[1]), float(f[2]))
            except ValueError as e:
                raise err(f"invalid {what} coordinate: {e}")
            (obj.Vertices if directive == "v" else obj.Normals).append(xyz)
        elif directive == "vt":  # obj.go:293-309
            f = _split_fields(rest, 2)
            if len(f) < 2:
                raise err(f"invalid texture coordinate data: expected at least 2 components, got {len(f)}")
            try:
                obj.TexCoords.append((float(f[0]), float(f[1])))  # a third component is an error, as in the reference
            except ValueError as e:
                raise err(f"invalid texture coordinate: {e}")
        elif directive == "f":  # obj.go:335-356
            parts = rest.split()
            if len(parts) < 3:
                raise err(f"face definition error: a face must have at least 3 vertices, got {len(parts)}")
            idx = []
            for part in parts:
                try:
                    idx.append(_parse_index(part, obj))
                except ValueError as e:
                    raise err(f"invalid face index '{part}': {e}")
            obj.Faces.append(Face(idx, current_material))
        elif directive == "usemtl":  # obj.go:442-448
            if rest == "":
                raise err("usemtl directive error: material name is missing")
            current_material = rest
        elif directive == "mtllib":  # obj.go:451-468
            if rest == "":
                raise err("mtllib directive error: filename is missing")
            for filename in rest.split():
                base = posixpath.basename(filename)
                try:
                    mtl = fsys.read(base)
                except OSError as e:
                    raise err(f"failed to read material library '{base}': {e}")
                _parse_mtl(mtl.decode("utf-8", errors="replace"), base, obj)
    return obj


def _scan_native(data: bytes, fsys, name: str):
    """The same object through the library's scanner (rt_obj_scan_text), arrays instead of lists; None when the library is
    not available or the scanner met a line it leaves to the line parser above (then that parser reads the whole file,
    so every error and corner case is its own).  RT_OBJ_NATIVE=0 switches the scanner off."""
    if os.environ.get("RT_OBJ_NATIVE", "1") == "0" or len(data) < _NATIVE_MIN_BYTES:
        return None
    from . import _lib, abi
    import ctypes as C
    lib = _lib.host_library()
    if lib is None:
        return None
    scan = abi.ObjScan()
    if lib.rt_obj_scan_text(data, len(data), C.byref(scan)) != abi.RT_OK:
        return None
    try:
        if scan.fallback_line != 0:
            return None

        def arr(ptr, n, width, dtype):
            if n == 0:
                return np.zeros((0, width) if width > 1 else (0,), dtype=dtype)
            a = np.ctypeslib.as_array(ptr, shape=(n * width,)).astype(dtype, copy=True)
            return a.reshape(n, width) if width > 1 else a

        obj = Object()
        obj.Vertices = arr(scan.vertices, scan.n_vertices, 3, np.float64)
        obj.TexCoords = arr(scan.texcoords, scan.n_texcoords, 2, np.float64)
        obj.Normals = arr(scan.normals, scan.n_normals, 3, np.float64)
        corners = arr(scan.corners, scan.n_corners, 3, np.int32)
        start = arr(scan.face_start, scan.n_faces + 1, 1, np.int64)
        face_event = arr(scan.face_event, scan.n_faces, 1, np.int32)
        names: List[str] = []
        event_name = np.full(scan.n_events + 1, -1, dtype=np.int32)  # last slot: faces before any usemtl
        libs = []
        for i in range(scan.n_events):
            ev = scan.events[i]
            arg = data[ev.arg_off:ev.arg_off + ev.arg_len].decode("utf-8", errors="replace")
            if ev.kind == abi.OBJ_USEMTL:
                if arg not in names:
                    names.append(arg)
                event_name[i] = names.index(arg)
            else:
                libs.append((int(ev.line), data[ev.line_off:ev.line_off + ev.line_len].decode("utf-8", errors="replace"), arg))
        obj.Faces = FaceTable(corners, start, event_name[face_event], names)
    finally:
        lib.rt_obj_scan_free(C.byref(scan))
    for line_no, line, rest in libs:  # obj.go:451-468
        for filename in rest.split():
            base = posixpath.basename(filename)
            try:
                mtl = fsys.read(base)
            except OSError as e:
                raise ParseError(name, line_no, line, f"failed to read material library '{base}': {e}")
            _parse_mtl(mtl.decode("utf-8", errors="replace"), base, obj)
    return obj


# Files smaller than this are read by the line parser: its lists of tuples are what small callers and the reference's
# own test tables expect, and the scanner only pays off on large files.
_NATIVE_MIN_BYTES = 1 << 16


def _parse_index(s: str, obj: Object) -> Tuple[int, int, int]:  # obj.go:359-428
    if s == "":
        raise ValueError("empty face index")
    parts = s.split("/")
    if len(parts) > 3:
        raise ValueError("invalid face index format")
    v = _resolve_index(int(parts[0]), len(obj.Vertices))
    vt = vn = 0
    if len(parts) == 2:
        vt = _resolve_index(int(parts[1]), len(obj.TexCoords))
    elif len(parts) == 3:
        if parts[1] != "":
            vt = _resolve_index(int(parts[1]), len(obj.TexCoords))
        if parts[2] != "":
            vn = _resolve_index(int(parts[2]), len(obj.Normals))
    return (v, vt, vn)


def _parse_mtl(text: str, filename: str, obj: Object) -> None:  # obj.go:471-661
    current = None
    for line_no, raw in enumerate(text.split("\n"), start=1):
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        sp = line.find(" ")
        if sp < 0:
            continue
        directive, rest = line[:sp], line[sp + 1:]

        def err(msg):
            return ParseError(filename, line_no, line, msg)

        if directive == "newmtl":
            if rest == "":
                raise err("newmtl directive error: material name is missing")
            current = Material(rest)
            obj.Materials[rest] = current
        elif directive in ("Kd", "Ka", "Ks"):
            if current is None:
                raise err(f"{directive} directive error: defined before any newmtl")
            f = _split_fields(rest, 3)
            if len(f) < 3:
                raise err(f"{directive} directive error: expected 3 components")
            try:
                rgb = (float(f[0]), float(f[1]), float(f[2]))
            except ValueError as e:
                raise err(f"invalid {directive} value: {e}")
            setattr(current, {"Kd": "Diffuse", "Ka": "Ambient", "Ks": "Specular"}[directive], rgb)
        elif directive == "Ns":
            if current is None:
                raise err("Ns directive error: defined before any newmtl")
            try:
                current.Shininess = float(_split_fields(rest, 1)[0])
            except ValueError as e:
                raise err(f"invalid Ns value: {e}")
        elif directive == "map_Kd":
            if current is None:
                raise err("map_Kd directive error: defined before any newmtl")
            if rest == "":
                raise err("map_Kd directive error: expected a filename")
            current.Texture = rest


# ---- PNG (what image/png.Decode gives NewTextureImageFS, texture_image.go:33-49) --------------


def decode_png(data: bytes) -> np.ndarray:
    """Non-interlaced PNG of colour type 0, 2, 3, 4 or 6 and bit depth 8 or 16 (1/2/4 for palettes and gray).
    Returns (H, W, 4) uint8 non-premultiplied NRGBA for depths <= 8 and (H, W, 4) uint16 alpha-premultiplied
    RGBA for 16-bit files: the two layouts phys.TextureImage accepts (what Go's Color.RGBA() reports)."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("png: invalid format: not a PNG file")
    pos, idat, palette, trns, hdr = 8, [], None, None, None
    while pos + 8 <= len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            palette = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = body
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    if hdr is None:
        raise ValueError("png: missing IHDR")
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace:
        raise ValueError("png: interlaced files are not supported")
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}.get(ctype)
    if channels is None or depth not in (1, 2, 4, 8, 16) or (depth < 8 and ctype not in (0, 3)):
        raise ValueError("png: unsupported colour type / bit depth")
    raw = zlib.decompress(b"".join(idat))
    bpp = max(1, channels * depth // 8)  # bytes per complete pixel, for the filters
    stride = (w * channels * depth + 7) // 8
    rows = np.zeros((h, stride), dtype=np.uint8)
    prev = bytes(stride)
    lanes = -(-stride // bpp) * bpp
    for y in range(h):
        ft = raw[y * (stride + 1)]
        line = raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)]
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = ((np.frombuffer(line, dtype=np.uint8).astype(np.int32) + np.frombuffer(prev, dtype=np.uint8)) & 255).astype(np.uint8).tobytes()
        elif ft == 1:  # Sub: a running sum per byte lane of the pixel
            lane = np.zeros(lanes, dtype=np.int64)
            lane[:stride] = np.frombuffer(line, dtype=np.uint8)
            cur = (np.cumsum(lane.reshape(-1, bpp), axis=0) & 255).astype(np.uint8).reshape(-1)[:stride].tobytes()
        elif ft in (3, 4):  # Average and Paeth depend on the byte to the left and the row above: byte-serial
            cur = bytearray(line)
            for i in range(min(bpp, stride)):
                b = prev[i]
                cur[i] = (cur[i] + ((b >> 1) if ft == 3 else b)) & 255
            if ft == 3:
                for i in range(bpp, stride):
                    cur[i] = (cur[i] + ((cur[i - bpp] + prev[i]) >> 1)) & 255
            else:
                for i in range(bpp, stride):
                    a, b, c = cur[i - bpp], prev[i], prev[i - bpp]
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    cur[i] = (cur[i] + (a if (pa <= pb and pa <= pc) else (b if pb <= pc else c))) & 255
            cur = bytes(cur)
        else:
            raise ValueError("png: bad filter type")
        rows[y] = np.frombuffer(cur, dtype=np.uint8)
        prev = cur
    if depth == 16:
        px = rows.reshape(h, -1, 2).astype(np.uint32)
        px = ((px[..., 0] << 8) | px[..., 1]).reshape(h, w, channels)
        full = 65535
    elif depth == 8:
        px = rows.reshape(h, w, channels).astype(np.uint32)
        full = 255
    else:  # packed gray / palette indices
        bits = np.unpackbits(rows, axis=1)[:, :w * depth].reshape(h, w, depth)
        px = bits.dot(1 << np.arange(depth - 1, -1, -1)).astype(np.uint32)[..., None]
        full = (1 << depth) - 1
    out = np.empty((h, w, 4), dtype=np.uint32)
    if ctype == 3:
        if palette is None:
            raise ValueError("png: palette missing")
        idx = px[..., 0]
        out[..., :3] = palette[idx]
        alpha = np.full(256, 255, dtype=np.uint32)
        if trns is not None:
            alpha[:len(trns)] = np.frombuffer(trns, dtype=np.uint8)
        out[..., 3] = alpha[idx]
        return out.astype(np.uint8)
    if ctype in (0, 4):
        g = px[..., 0] * (255 // full) if depth < 8 else px[..., 0]
        out[..., 0] = out[..., 1] = out[..., 2] = g
        out[..., 3] = px[..., 1] if ctype == 4 else (65535 if depth == 16 else 255)
    else:
        out[..., :3] = px[..., :3]
        out[..., 3] = px[..., 3] if ctype == 6 else (65535 if depth == 16 else 255)
    if depth == 16:  # NRGBA64 -> premultiplied, as color.NRGBA64.RGBA() does
        a = out[..., 3:4].astype(np.uint64)
        out[..., :3] = (out[..., :3].astype(np.uint64) * a // 0xffff).astype(np.uint32)
        return out.astype(np.uint16)
    return out.astype(np.uint8)


def encode_png_rgba8(img: np.ndarray) -> bytes:
    """Minimal writer (filter 0, one IDAT) for (H, W, 4) uint8 images; used by tests and for saving renders."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xffffffff)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + \
        chunk(b"IEND", b"")

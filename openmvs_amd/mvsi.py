"""MVSI scene archive (`*.mvs`) reader / writer — the data format upstream of the dense path.

The layout is `MVS::Interface`'s own binary archive (reference `libs/MVS/Interface.h:215-275` header,
`:360-760` records; the reference's Python reader is `scripts/python/MvsUtils.py:72-187`):

    "MVSI" u32 version u32 reserved
    platforms[]: name, cameras[]{name, (v>3) bandName, (v>0) u32 width,height, f64 K[9], f64 R[9], f64 C[3]}, poses[]{f64 R[9], f64 C[3]}
    images[]:    name, (v>4) maskName, u32 platformID,cameraID,poseID, (v>2) u32 ID,
                 (v>6) f32 minDepth,avgDepth,maxDepth, viewScores[]{u32 ID,points, f32 scale,angle,area,score}
    vertices[]:  f32 X[3], views[]{u32 imageID, f32 confidence}
    verticesNormal[] f32[3]; verticesColor[] u8[3]
    (v>0) lines[]{f32 pt1[3], pt2[3], views[]}, linesNormal[], linesColor[]
    (v>1) f64 transform[16]; (v>5) f64 obb.rot[9], ptMin[3], ptMax[3]

Every array is a u64 count followed by its elements; strings are u64 length + bytes.  All little-endian.

This module keeps the point cloud as flat numpy arrays (CSR views), not one dict per vertex, so a scene with millions of sparse
points loads in one pass.  `Scene.cameras()` composes the per-image pinhole camera exactly as the reference does
(`Platform::GetCamera`, `libs/MVS/Platform.cpp:44-54`; `Image::GetCamera`, `libs/MVS/Image.cpp:190-203`;
`Scene::LoadInterface`, `libs/MVS/Scene.cpp:95-110`).
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field

import numpy as np

MVSI_PROJECT_VER = 7            # Interface.h:16
NO_ID = 0xFFFFFFFF


class _Reader:
    def __init__(self, buf: bytes):
        self.b = memoryview(buf)
        self.o = 0

    def take(self, n: int) -> memoryview:
        if self.o + n > len(self.b):
            raise ValueError("MVSI: truncated archive")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def u32(self) -> int:
        return struct.unpack_from("<I", self.take(4))[0]

    def u64(self) -> int:
        return struct.unpack_from("<Q", self.take(8))[0]

    def string(self) -> str:
        return bytes(self.take(self.u64())).decode("utf-8", "replace")

    def arr(self, dtype, n: int) -> np.ndarray:
        dt = np.dtype(dtype)
        return np.frombuffer(self.take(dt.itemsize * n), dtype=dt, count=n).copy()


@dataclass
class Camera:
    name: str = ""
    band_name: str = ""
    width: int = 0
    height: int = 0
    K: np.ndarray = field(default_factory=lambda: np.eye(3))
    R: np.ndarray = field(default_factory=lambda: np.eye(3))
    C: np.ndarray = field(default_factory=lambda: np.zeros(3))

    def is_normalized(self) -> bool:          # Interface.h:392
        return not (self.width > 0 and self.height > 0)


@dataclass
class Platform:
    name: str = ""
    cameras: list = field(default_factory=list)
    poses_R: np.ndarray = field(default_factory=lambda: np.zeros((0, 3, 3)))
    poses_C: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))


@dataclass
class Image:
    name: str = ""
    mask_name: str = ""
    platform_id: int = NO_ID
    camera_id: int = NO_ID
    pose_id: int = NO_ID
    id: int = NO_ID
    min_depth: float = 0.0
    avg_depth: float = 0.0
    max_depth: float = 0.0
    view_scores: np.ndarray = field(default_factory=lambda: np.zeros(0, VIEW_SCORE_DTYPE))

    def is_valid(self) -> bool:               # Interface.h:560
        return self.pose_id != NO_ID


VIEW_SCORE_DTYPE = np.dtype([("ID", "<u4"), ("points", "<u4"), ("scale", "<f4"), ("angle", "<f4"), ("area", "<f4"), ("score", "<f4")])
VIEW_DTYPE = np.dtype([("image_id", "<u4"), ("confidence", "<f4")])


@dataclass
class Scene:
    version: int = MVSI_PROJECT_VER
    platforms: list = field(default_factory=list)
    images: list = field(default_factory=list)
    vertices: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    vertex_view_start: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int64))   # CSR offsets, len = nVertices+1
    vertex_views: np.ndarray = field(default_factory=lambda: np.zeros(0, VIEW_DTYPE))
    vertices_normal: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    vertices_color: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.uint8))
    lines: np.ndarray = field(default_factory=lambda: np.zeros((0, 2, 3), np.float32))
    line_view_start: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int64))
    line_views: np.ndarray = field(default_factory=lambda: np.zeros(0, VIEW_DTYPE))
    lines_normal: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    lines_color: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.uint8))
    transform: np.ndarray = field(default_factory=lambda: np.eye(4))
    obb_rot: np.ndarray = field(default_factory=lambda: np.eye(3))
    obb_min: np.ndarray = field(default_factory=lambda: np.zeros(3))
    obb_max: np.ndarray = field(default_factory=lambda: np.zeros(3))

    # ---- derived -------------------------------------------------------------------------------------------------
    def _obb(self):
        """`TOBB::Set(rot, ptMin, ptMax)` in float (libs/Common/OBB.inl:64-69, called at libs/MVS/Scene.cpp:208)."""
        rot = self.obb_rot.astype(np.float32)
        mn, mx = self.obb_min.astype(np.float32), self.obb_max.astype(np.float32)
        return rot, (mx + mn) * np.float32(0.5), (mx - mn) * np.float32(0.5)

    def is_bounded(self) -> bool:
        """`Scene::IsBounded` (libs/MVS/Scene.h:73 -> OBB.inl:272-275): every half-extent is positive."""
        return bool(self._obb()[2].min() > 0)

    def roi_contains(self, pts: np.ndarray) -> np.ndarray:
        """`TOBB::Intersects(point)` (libs/Common/OBB.inl:388-400) for an (n,3) float32 array."""
        rot, pos, ext = self._obb()
        d = (np.asarray(pts, np.float32) - pos) @ rot.T
        return np.all(np.abs(d) <= ext, axis=1)

    def camera(self, idx: int, size: tuple | None = None):
        """(K, R, C, width, height) of image `idx` in pixels at `size` = (width, height) (default: the camera's resolution).

        K goes through the reference's normalise -> de-normalise round trip (Scene.cpp:100-104 then Camera.h:190-200 with the
        half-pixel convention of `ScaleK`, Camera.h:146-152) so it carries the same last-bit rounding."""
        im = self.images[idx]
        pl = self.platforms[im.platform_id]
        cam = pl.cameras[im.camera_id]
        K = np.array(cam.K, np.float64)
        if not cam.is_normalized():
            K = _scale_k(K, 1.0 / float(np.float32(max(cam.width, cam.height))))
        w, h = size if size is not None else (cam.width, cam.height)
        if w <= 0 or h <= 0:
            raise ValueError("MVSI: image %d has no resolution; pass size=(w,h)" % idx)
        s = float(np.float32(max(w, h)))
        if K[0, 2] != 0 or K[1, 2] != 0:
            Kp = _scale_k(K, s)
        else:                                   # ComposeK: principal point at the image centre
            Kp = np.array([[K[0, 0] * s, 0, 0.5 * (w - 1)], [0, K[1, 1] * s, 0.5 * (h - 1)], [0, 0, 1]], np.float64)
        Rp, Cp = pl.poses_R[im.pose_id], pl.poses_C[im.pose_id]
        R = matx_mul(cam.R, Rp)                                           # Platform::GetCamera, libs/MVS/Platform.cpp:44-54
        C = matx_mul(np.asarray(Rp).T, cam.C) + np.asarray(Cp, np.float64)
        return Kp, R, C, int(w), int(h)

    def views_of(self, v: int) -> np.ndarray:
        return self.vertex_views[self.vertex_view_start[v]:self.vertex_view_start[v + 1]]


def matx_mul(a, b) -> np.ndarray:
    """cv::Matx product as the reference's small matrices multiply: c(i, j) = sum_k a(i, k) b(k, j), accumulated left to right from 0 in double.  (numpy's `@` goes through
    BLAS, whose blocked / fused summation differs from this in the last bit for almost every product.)"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    vec = b.ndim == 1
    if vec:
        b = b[:, None]
    c = np.zeros((a.shape[0], b.shape[1]))
    for i in range(a.shape[0]):
        for j in range(b.shape[1]):
            s = 0.0
            for k in range(a.shape[1]):
                s += a[i, k] * b[k, j]
            c[i, j] = s
    return c[:, 0] if vec else c


def _scale_k(K: np.ndarray, s: float) -> np.ndarray:
    """`Camera::ScaleK` (libs/MVS/Camera.h:146-152)."""
    return np.array([[K[0, 0] * s, K[0, 1] * s, (K[0, 2] + 0.5) * s - 0.5],
                     [0.0, K[1, 1] * s, (K[1, 2] + 0.5) * s - 0.5],
                     [0.0, 0.0, 1.0]], np.float64)


def _read_view_lists(r: _Reader, n: int, point_floats: int):
    """n records of {f32[point_floats], u64 count, count x (u32,f32)} -> (points, csr offsets, views)."""
    pts = np.empty((n, point_floats), np.float32)
    start = np.zeros(n + 1, np.int64)
    chunks = []
    b, o = r.b, r.o
    pf = point_floats * 4
    for i in range(n):
        if o + pf + 8 > len(b):
            raise ValueError("MVSI: truncated archive")
        pts[i] = np.frombuffer(b[o:o + pf], "<f4")
        (m,) = struct.unpack_from("<Q", b, o + pf)
        o += pf + 8
        if o + 8 * m > len(b):
            raise ValueError("MVSI: truncated archive")
        chunks.append(b[o:o + 8 * m])
        o += 8 * m
        start[i + 1] = start[i] + m
    r.o = o
    views = np.frombuffer(b"".join(bytes(c) for c in chunks), VIEW_DTYPE).copy() if n else np.zeros(0, VIEW_DTYPE)
    return pts, start, views


def load(path: str) -> Scene:
    with open(path, "rb") as f:
        buf = f.read()
    r = _Reader(buf)
    sc = Scene()
    if bytes(r.take(4)) == b"MVSI":
        sc.version = r.u32()
        if sc.version > MVSI_PROJECT_VER:
            raise ValueError("MVSI: version %d is newer than %d" % (sc.version, MVSI_PROJECT_VER))
        r.u32()
    else:                                       # head-less first version, only for *.mvs (Interface.h:252-262)
        if not path.lower().endswith(".mvs"):
            raise ValueError("MVSI: not a scene archive: %s" % path)
        sc.version = 0
        r.o = 0
    v = sc.version
    for _ in range(r.u64()):
        pl = Platform(name=r.string())
        for _ in range(r.u64()):
            cam = Camera(name=r.string())
            if v > 3:
                cam.band_name = r.string()
            if v > 0:
                cam.width, cam.height = r.u32(), r.u32()
            cam.K = r.arr("<f8", 9).reshape(3, 3)
            cam.R = r.arr("<f8", 9).reshape(3, 3)
            cam.C = r.arr("<f8", 3)
            pl.cameras.append(cam)
        npose = r.u64()
        poses = r.arr("<f8", 12 * npose).reshape(npose, 12)
        pl.poses_R = poses[:, :9].reshape(npose, 3, 3).copy()
        pl.poses_C = poses[:, 9:].copy()
        sc.platforms.append(pl)
    for _ in range(r.u64()):
        im = Image(name=r.string())
        if v > 4:
            im.mask_name = r.string()
        im.platform_id, im.camera_id, im.pose_id = r.u32(), r.u32(), r.u32()
        if v > 2:
            im.id = r.u32()
        if v > 6:
            im.min_depth, im.avg_depth, im.max_depth = (float(x) for x in r.arr("<f4", 3))
            im.view_scores = r.arr(VIEW_SCORE_DTYPE, r.u64())
        sc.images.append(im)
    sc.vertices, sc.vertex_view_start, sc.vertex_views = _read_view_lists(r, r.u64(), 3)
    sc.vertices_normal = r.arr("<f4", 3 * r.u64()).reshape(-1, 3)
    sc.vertices_color = r.arr("u1", 3 * r.u64()).reshape(-1, 3)
    if v > 0:
        pts, sc.line_view_start, sc.line_views = _read_view_lists(r, r.u64(), 6)
        sc.lines = pts.reshape(-1, 2, 3)
        sc.lines_normal = r.arr("<f4", 3 * r.u64()).reshape(-1, 3)
        sc.lines_color = r.arr("u1", 3 * r.u64()).reshape(-1, 3)
        if v > 1:
            sc.transform = r.arr("<f8", 16).reshape(4, 4)
            if v > 5:
                sc.obb_rot = r.arr("<f8", 9).reshape(3, 3)
                sc.obb_min = r.arr("<f8", 3)
                sc.obb_max = r.arr("<f8", 3)
    return sc


def save(path: str, sc: Scene, version: int | None = None) -> None:
    """Write `sc` as an MVSI archive of `version` (default `sc.version`); the inverse of `load`, byte for byte."""
    v = sc.version if version is None else version
    out = []
    w = out.append

    def s(x: str):
        e = x.encode("utf-8")
        w(struct.pack("<Q", len(e)))
        w(e)

    def n(x: int):
        w(struct.pack("<Q", x))

    def a(x, dt):
        w(np.ascontiguousarray(x, dtype=dt).tobytes())

    if v > 0:
        w(b"MVSI")
        w(struct.pack("<II", v, 0))
    n(len(sc.platforms))
    for pl in sc.platforms:
        s(pl.name)
        n(len(pl.cameras))
        for cam in pl.cameras:
            s(cam.name)
            if v > 3:
                s(cam.band_name)
            if v > 0:
                w(struct.pack("<II", cam.width, cam.height))
            a(cam.K, "<f8"); a(cam.R, "<f8"); a(cam.C, "<f8")
        n(len(pl.poses_R))
        for R, C in zip(pl.poses_R, pl.poses_C):
            a(R, "<f8"); a(C, "<f8")
    n(len(sc.images))
    for im in sc.images:
        s(im.name)
        if v > 4:
            s(im.mask_name)
        w(struct.pack("<III", im.platform_id, im.camera_id, im.pose_id))
        if v > 2:
            w(struct.pack("<I", im.id))
        if v > 6:
            w(struct.pack("<fff", im.min_depth, im.avg_depth, im.max_depth))
            n(len(im.view_scores))
            a(im.view_scores, VIEW_SCORE_DTYPE)

    def view_lists(pts, start, views):
        n(len(pts))
        for i in range(len(pts)):
            a(pts[i], "<f4")
            n(int(start[i + 1] - start[i]))
            a(views[start[i]:start[i + 1]], VIEW_DTYPE)

    view_lists(sc.vertices.reshape(len(sc.vertices), 3), sc.vertex_view_start, sc.vertex_views)
    n(len(sc.vertices_normal)); a(sc.vertices_normal, "<f4")
    n(len(sc.vertices_color)); a(sc.vertices_color, "u1")
    if v > 0:
        view_lists(sc.lines.reshape(len(sc.lines), 6), sc.line_view_start, sc.line_views)
        n(len(sc.lines_normal)); a(sc.lines_normal, "<f4")
        n(len(sc.lines_color)); a(sc.lines_color, "u1")
        if v > 1:
            a(sc.transform, "<f8")
            if v > 5:
                a(sc.obb_rot, "<f8"); a(sc.obb_min, "<f8"); a(sc.obb_max, "<f8")
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(b"".join(out))
    os.replace(tmp, path)


def _sml_entries(chunk: str, out):
    blank = "\n\r\t "
    for line in chunk.split("\n"):
        line = line.strip(blank)
        if not line:
            continue
        name, eq, val = line.partition("=")
        name = name.strip(blank) if eq else ""
        out.append((name, val.strip(blank)) if name else ("", val.strip(blank) if eq else line))


def _sml_section(text: str, pos: int, out, depth: int = 0):
    """One section of an SML text from text[pos] (SML::ParseSection, libs/Common/SML.cpp:103-160) -> (ok, position behind it); same rule as csrc/sml_text.h."""
    while True:
        open_ = text.find("[", pos)
        last = open_ < 0
        if last:
            open_ = len(text)
        close = text.find("}", pos)
        if 0 <= close < open_:
            if out is not None:
                _sml_entries(text[pos:close], out)
            return True, close + 1
        if out is not None:
            _sml_entries(text[pos:open_], out)
        if last:
            return True, len(text)
        name_end = text.find("]", open_ + 1)
        if name_end < 0:                                               # a name that is never closed swallows the rest of the text
            return True, len(text)
        if name_end == open_ + 1:                                      # "[]": a parse error (blanks count as a name)
            return False, len(text)
        body = text.find("{", name_end + 1)
        if body < 0:
            return True, len(text)
        ok, pos = _sml_section(text, body + 1, None, depth + 1) if depth <= 64 else (False, len(text))
        if not ok:
            return False, pos


def _sml_root_values(text: str):
    """(name, value) of the ROOT entries of an SML text (libs/Common/SML.cpp:94-227): the text outside child sections ("[name]" + "{ ... }"), one entry per line,
    "name = value" or -- without a '=' or with nothing in front of it -- an unnamed entry whose value is the line (SML_AUTOVALUES).  A malformed section ends the reading;
    what stood in front of it counts."""
    out = []
    _sml_section(text, 0, out)
    return out


def _split_words(line: str):
    """Util::CommandLineToArgvA (libs/Common/Util.cpp:740-803): blanks separate, double quotes group."""
    words, quoted, in_space = [], False, True
    for a in line:
        if quoted:
            if a == '"':
                quoted = False
            else:
                words[-1] += a
        elif a == '"':
            quoted = True
            if in_space:
                words.append("")
            in_space = False
        elif a in " \t\n\r":
            in_space = True
        else:
            if in_space:
                words.append("")
            words[-1] += a
            in_space = False
    return words


def load_view_neighbors(sc: Scene, path: str) -> None:
    """`Scene::LoadViewNeighbors` (libs/MVS/Scene.cpp:413-457; `DensifyPointCloud --view-neighbors-file`): one line per image, "<id> <neighbour-0> <neighbour-1> ...",
    best first; lines starting with '#' and lines with fewer than two words are skipped.  Every listed neighbour becomes ViewScore{ID, 0, 1, 15 deg, 0.5, 3} (`:451`) in
    `images[id].view_scores`, which `views.select_views` then uses instead of scoring the views (SceneDensify.cpp:278-281).  An id outside the scene raises (the
    reference asserts) and leaves the scene untouched."""
    with open(path, "rb") as f:
        text = f.read().decode("latin-1")

    def index(w):
        if not (0 < len(w) <= 10 and w.isascii() and w.isdigit() and int(w) < len(sc.images)):
            raise ValueError("%s: %r is not an image of this scene" % (path, w))
        return int(w)

    lists = []
    for _, val in _sml_root_values(text):
        words = _split_words(val)
        if (words and words[0][:1] == "#") or len(words) < 2:
            continue
        nb = np.zeros(len(words) - 1, VIEW_SCORE_DTYPE)
        nb["ID"] = [index(w) for w in words[1:]]
        nb["scale"], nb["angle"], nb["area"], nb["score"] = 1.0, np.float32(15.0) * (np.float32(3.14159265358979323846) / np.float32(180.0)), 0.5, 3.0
        lists.append((index(words[0]), nb))
    for i, nb in lists:
        sc.images[i].view_scores = nb


def save_view_neighbors(sc: Scene, path: str) -> None:
    """`Scene::SaveViewNeighbors` (libs/MVS/Scene.cpp:458-480): every image, "<id> <n0> <n1> ...\\n"."""
    with open(path, "wb") as f:
        for i, im in enumerate(sc.images):
            f.write((" ".join([str(i)] + [str(int(n)) for n in im.view_scores["ID"]]) + "\n").encode())

"""Host-side callers in front of the depth-map estimator: neighbour-view selection and depth-map initialisation.

These are the steps `Scene::ComputeDepthMaps` runs between loading the scene and the first `EstimateDepthMap` call
(reference `libs/MVS/SceneDensify.cpp:1754-1870`): per image it scores every other image by the sparse points they share
(`Scene::SelectNeighborViews`, `libs/MVS/Scene.cpp:801-934`), drops the bad ones (`Scene::FilterNeighborViews`, `:953-968`),
cuts the list at a score ratio (`DepthMapsData::InitViews`, `SceneDensify.cpp:333-340`) and seeds the reference depth map from the
sparse points (`:418-460`, `TriangulatePoints2DepthMap`, `libs/MVS/DepthMap.cpp:1117-1192`).

They operate on a few thousand sparse points per image and stay on the host, as in the reference; numpy in float32/float64 follows
the reference's mixed precision term by term (the order of the float sums included) so that the ranking it produces is the
reference's.  Parity status: unpinned — the reference ships no golden neighbour lists (its `scene.mvs` test fixture is archive
version 6, which predates stored view scores), so tests check invariants and the end-to-end acceptance of
`apps/Tests/Tests.cpp:78-105` instead.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import mvsi

VIEW_SCORE_DTYPE = mvsi.VIEW_SCORE_DTYPE
f32 = np.float32


@dataclass
class DenseOptions:
    """The `OPTDENSE` values read by view selection and initialisation (defaults: `libs/MVS/DepthMap.cpp:69-90`)."""
    nResolutionLevel: int = 1
    nMaxResolution: int = 3200
    nMinResolution: int = 640
    nMinViews: int = 2
    nMaxViews: int = 12
    nMinViewsTrustPoint: int = 2
    nNumViews: int = 0
    nPointInsideROI: int = 1
    bAddCorners: bool = False
    bInitSparse: bool = True
    fViewMinScore: float = 2.0
    fViewMinScoreRatio: float = 0.03
    fMinArea: float = 0.05
    fMinAngle: float = 3.0
    fOptimAngle: float = 12.0
    fMaxAngle: float = 65.0


def compute_max_resolution(width: int, height: int, level: int, min_size: int, max_size: int):
    """`TImage::computeMaxResolution` (libs/Common/Types.inl:2459-2477) -> (max resolution, effective level)."""
    size0 = max(width, height)
    if level == 0:
        return min(size0, max_size), 0
    size = size0 >> level
    if size < min_size:
        level = 0
        while (size0 >> (level + 1)) >= min_size:
            level += 1
        size = size0 >> level
    return min(size, max_size), level


def resized_size(width: int, height: int, max_resolution: int):
    """`Image::ResizeImage` size rule (libs/MVS/Image.cpp:139-150; `computeResize`, Types.inl:2440-2445)."""
    if max_resolution == 0 or max(width, height) <= max_resolution:
        return width, height
    scale = max_resolution / width if width > height else max_resolution / height
    return int(np.rint(width * scale)), int(np.rint(height * scale))


def to_gray(rgb: np.ndarray) -> np.ndarray:
    """`TImage::toGray(out, COLOR_BGR2GRAY, bNormalize=true)` (libs/Common/Types.inl:2377-2425) for an (h,w,3) uint8 RGB array:
    float32 0.114*B + 0.587*G + 0.299*R of the channels times (1.f / 255.f) -- `CONVERT::NormRGB_t`, Types.inl:1610-1615: a product, which differs from a division
    by 255 in the last bit for half of the byte values -- summed in that (B, G, R) order.  Pinned to the reference's helpers (tests/test_ref_text.py)."""
    c = rgb.astype(f32) * (f32(1) / f32(255))
    return (f32(0.114) * c[..., 2] + f32(0.587) * c[..., 1]) + f32(0.299) * c[..., 0]


class Cameras:
    """Pixel cameras of every valid image of a scene at the working resolution (K, R, C, P in float64 like `MVS::Camera`)."""

    def __init__(self, scene: mvsi.Scene, sizes=None):
        n = len(scene.images)
        self.valid = np.array([im.is_valid() for im in scene.images])
        self.K = np.zeros((n, 3, 3)); self.R = np.zeros((n, 3, 3)); self.C = np.zeros((n, 3)); self.P = np.zeros((n, 3, 4))
        self.size = np.zeros((n, 2), np.int64)
        for i in range(n):
            if not self.valid[i]:
                continue
            K, R, C, w, h = scene.camera(i, None if sizes is None else sizes[i])
            self.K[i], self.R[i], self.C[i], self.size[i] = K, R, C, (w, h)
            M = mvsi.matx_mul(K, R)                     # Camera::ComposeP -> AssembleProjectionMatrix (libs/MVS/Camera.cpp:173-180): P = [K R | K R (-C)]
            self.P[i, :, :3] = M
            self.P[i, :, 3] = mvsi.matx_mul(M, -np.asarray(C, np.float64))

    def point_depth(self, i: int, X: np.ndarray) -> np.ndarray:
        """`Camera::PointDepth` (libs/Common/Util.inl:462-464): R[2].(X-C) in double."""
        return (np.asarray(X, np.float64) - self.C[i]) @ self.R[i, 2]

    def project_p(self, i: int, X: np.ndarray) -> np.ndarray:
        """`Camera::ProjectPointP<float>` (libs/MVS/Camera.h:308-320): rows of P in double, each cast to float, then x*(1/z)."""
        q = (np.asarray(X, np.float64) @ self.P[i, :, :3].T + self.P[i, :, 3]).astype(f32)
        inv = f32(1) / q[:, 2]
        return np.stack([q[:, 0] * inv, q[:, 1] * inv], 1)


def _seq_sum_f32(x: np.ndarray) -> np.float32:
    """Left-to-right float32 sum (the reference accumulates in a scalar loop; numpy's `sum` is pairwise)."""
    return np.add.accumulate(x.astype(f32), dtype=f32)[-1] if len(x) else f32(0)


def select_neighbor_views(scene: mvsi.Scene, cams: Cameras, ID: int, nMinViews=2, nMinPointViews=2,
                          fOptimAngle=np.deg2rad(12.0), nInsideROI=1):
    """`Scene::SelectNeighborViews(ID, points, ...)` (libs/MVS/Scene.cpp:801-934).

    Returns (ok, neighbors, points, avgDepth): `neighbors` a VIEW_SCORE_DTYPE array sorted by decreasing score, `points` the indices of
    the sparse points seen by `ID` in at least `nMinPointViews` views, `avgDepth` the mean depth of all points seen by `ID`."""
    nImages = len(scene.images)
    nCalibrated = int(cams.valid.sum())
    nMinPointViews = min(nMinPointViews, nCalibrated)
    fOptimAngle = f32(fOptimAngle)
    sigmaSmall = f32(-1) / (f32(2) * (fOptimAngle * f32(0.38)) ** 2)
    sigmaLarge = f32(-1) / (f32(2) * (fOptimAngle * f32(0.7)) ** 2)
    start, views = scene.vertex_view_start, scene.vertex_views["image_id"]
    nviews = np.diff(start)
    owner = np.repeat(np.arange(len(scene.vertices)), nviews)            # vertex of every (vertex, view) record
    seen = np.zeros(len(scene.vertices), bool)
    seen[owner[views == ID]] = True
    X = scene.vertices
    wROI = np.ones(len(X), f32)
    if nInsideROI > 0 and scene.is_bounded():
        inside = scene.roi_contains(X)
        if nInsideROI > 1:
            seen &= inside
        wROI[~inside] = f32(0.7)
    depth = cams.point_depth(ID, X).astype(f32)
    seen &= depth > 0
    idxSeen = np.nonzero(seen)[0]
    points = idxSeen[nviews[idxSeen] >= nMinPointViews].astype(np.uint32)
    nPoints = len(idxSeen)
    avgDepth = _seq_sum_f32(depth[idxSeen])
    if nPoints > 3:
        avgDepth = avgDepth / f32(nPoints)
    # score shared views: one record per (seen vertex, other view), in the reference's vertex-major order
    rec = np.nonzero(seen[owner] & (views != ID))[0]
    pv, vw = owner[rec], views[rec].astype(np.int64)
    Xd = X[pv].astype(np.float64)
    V1 = (cams.C[ID] - Xd).astype(f32)
    V2 = (cams.C[vw] - Xd).astype(f32)
    dot = (V1[:, 0] * V2[:, 0] + V1[:, 1] * V2[:, 1]) + V1[:, 2] * V2[:, 2]
    n1 = (V1[:, 0] * V1[:, 0] + V1[:, 1] * V1[:, 1]) + V1[:, 2] * V1[:, 2]
    n2 = (V2[:, 0] * V2[:, 0] + V2[:, 1] * V2[:, 1]) + V2[:, 2] * V2[:, 2]
    fAngle = np.arccos(np.clip(dot / np.sqrt(n1 * n2), f32(-1), f32(1))).astype(f32)
    dA = fAngle - fOptimAngle
    wAngle = np.exp((dA * dA) * np.where(fAngle < fOptimAngle, sigmaSmall, sigmaLarge)).astype(f32)
    foot1 = (cams.K[ID, 0, 0] / cams.point_depth(ID, Xd)).astype(f32)
    foot2 = (cams.K[vw, 0, 0] / np.einsum("ij,ij->i", Xd - cams.C[vw], cams.R[vw, 2])).astype(f32)
    ratio = foot1 / foot2
    wScale = np.where(ratio > f32(1.6), (f32(1.6) / ratio) ** 2, np.where(ratio >= f32(1), f32(1), ratio * ratio)).astype(f32)
    contrib = (np.maximum(wAngle, f32(0.1)) * wScale) * wROI[pv]
    score = np.zeros(nImages, f32); sumScale = np.zeros(nImages, f32); sumAngle = np.zeros(nImages, f32)
    count = np.zeros(nImages, np.uint32)
    np.add.at(score, vw, contrib)                    # ufunc.at applies the records one after the other, like the reference loop
    np.add.at(sumScale, vw, ratio)
    np.add.at(sumAngle, vw, fAngle)
    np.add.at(count, vw, 1)
    # covered area of the shared projections, per candidate
    out = []
    boundsA = cams.size[ID].astype(f32)
    inPoints = np.zeros(len(X), bool)
    inPoints[points] = True
    projA = cams.project_p(ID, X[points])
    insideA = (projA[:, 0] >= 0) & (projA[:, 1] >= 0) & (projA[:, 0] < boundsA[0]) & (projA[:, 1] < boundsA[1])
    pos = np.full(len(X), -1, np.int64)
    pos[points] = np.arange(len(points))
    for IDB in range(nImages):
        if not cams.valid[IDB] or count[IDB] < 3 or IDB == ID:
            continue
        shared = owner[(views == IDB) & inPoints[owner]]
        if len(shared) == 0:
            continue
        k = pos[shared]
        projB = cams.project_p(IDB, X[shared])
        bB = cams.size[IDB].astype(f32)
        ok = insideA[k] & (projB[:, 0] >= 0) & (projB[:, 1] >= 0) & (projB[:, 0] < bB[0]) & (projB[:, 1] < bB[1])
        if not ok.any():
            continue
        area = covered_area(projA[k][ok], boundsA)
        out.append((IDB, count[IDB], sumScale[IDB] / f32(count[IDB]), sumAngle[IDB] / f32(count[IDB]), area,
                    score[IDB] * max(area, f32(0.01))))
    neighbors = np.array(out, VIEW_SCORE_DTYPE) if out else np.zeros(0, VIEW_SCORE_DTYPE)
    neighbors = neighbors[np.argsort(-neighbors["score"], kind="stable")]
    ok = len(points) > 3 and len(neighbors) >= min(nMinViews, nCalibrated - 1)
    return ok, neighbors, points, float(avgDepth)


def covered_area(projs: np.ndarray, bounds: np.ndarray, s: int = 16) -> np.float32:
    """`ComputeCoveredArea<float,2,16,false>` (libs/Common/Util.inl:846-866): fraction of the s x s grid cells hit."""
    cell = np.floor((projs / bounds) * f32(s)).astype(np.int64)
    return f32(len(np.unique(cell[:, 0] * s + cell[:, 1]))) / f32(s * s)


def filter_neighbor_views(neighbors: np.ndarray, fMinArea=0.05, fMinScale=0.2, fMaxScale=3.2,
                          fMinAngle=np.deg2rad(3.0), fMaxAngle=np.deg2rad(65.0), nMaxViews=12) -> np.ndarray:
    """`Scene::FilterNeighborViews` (libs/MVS/Scene.cpp:953-968): scan from the worst neighbour up, dropping invalid ones while more
    than max(4, 3/4 nMaxViews) remain, then keep the best nMaxViews."""
    keep = list(range(len(neighbors)))
    nMin = max(4, nMaxViews * 3 // 4)
    for n in range(len(neighbors) - 1, -1, -1):
        nb = neighbors[n]
        bad = (nb["area"] < f32(fMinArea) or not (f32(fMinScale) <= nb["scale"] < f32(fMaxScale))      # ISINSIDE is half-open,
               or not (f32(fMinAngle) <= nb["angle"] < f32(fMaxAngle)))                                  # libs/Common/Types.h:1193
        if len(keep) > nMin and bad:
            keep.remove(n)
    return neighbors[keep][:nMaxViews]


def select_views(scene: mvsi.Scene, cams: Cameras, ID: int, opt: DenseOptions = DenseOptions(), all_neighbors: list | None = None):
    """`DepthMapsData::SelectViews(DepthData&)` (libs/MVS/SceneDensify.cpp:273-293) followed by the score cut of `InitViews`
    (`:333-340`).  Returns (neighbors, points, avgDepth) or None when the image cannot be densified.  `all_neighbors` (a list) receives the image's WHOLE scored list --
    `Image::neighbors`, which the reference keeps on the scene (it is what `scene_dense.mvs` stores as view scores and what orders the fusion, SceneDensify.cpp:1423)."""
    stored = scene.images[ID].view_scores
    if len(stored):            # a list the scene already carries -- the archive's view scores (Scene.cpp:158) or a view-neighbours file (mvsi.load_view_neighbors) -- is
        nb, points, avg = stored.copy(), np.zeros(0, np.int64), float(scene.images[ID].avg_depth)   # taken as it is (SceneDensify.cpp:278-281); no seed points then
    else:
        ok, nb, points, avg = select_neighbor_views(scene, cams, ID, opt.nMinViews,
                                                    opt.nMinViewsTrustPoint if opt.nMinViewsTrustPoint > 1 else 2,
                                                    np.deg2rad(opt.fOptimAngle), opt.nPointInsideROI)
        if not ok:
            return None
    if all_neighbors is not None:
        all_neighbors.append(nb.copy())
    nb = filter_neighbor_views(nb, opt.fMinArea, 0.2, 3.2, np.deg2rad(opt.fMinAngle), np.deg2rad(opt.fMaxAngle), opt.nMaxViews)
    if len(nb) == 0:
        return None
    fMinScore = max(nb[0]["score"] * f32(opt.fViewMinScoreRatio), f32(opt.fViewMinScore))
    cut = len(nb)
    for i in range(len(nb)):
        if (opt.nNumViews and i + 1 > opt.nNumViews) or nb[i]["score"] < fMinScore:
            cut = i
            break
    nb = nb[:cut]
    if len(nb) == 0:
        return None
    return nb, points, avg


def init_depth_map(scene: mvsi.Scene, cams: Cameras, ID: int, points: np.ndarray, opt: DenseOptions = DenseOptions(), avg_depth=None):
    """The `loadDepthMaps == 0` branch of `DepthMapsData::InitViews` (libs/MVS/SceneDensify.cpp:418-460).

    Returns (depthMap, normalMap, dMin, dMax).  Three cases, as in the reference:
      * no points: empty maps, range [0.1, 100];
      * `nMinViewsTrustPoint < 2`: each point's depth splatted on a 5x5 block with a zero normal (`:428-452`);
      * otherwise `TriangulatePoints2DepthMap` with `bInitSparse` (DepthMap.cpp:1117-1157): each point written to the 2x2 block around
        its projection with the area-weighted vertex normal of the Delaunay mesh of the projections (`Mesh::ComputeNormalVertices`,
        libs/MVS/Mesh.cpp:356-371).  The reference triangulates with CGAL; scipy's Qhull Delaunay gives the same triangulation except
        for co-circular point sets, so normals may differ at such vertices (the estimator treats them as initial guesses only).
    With `bInitSparse=0` the faces of that mesh are rasterised instead (DepthMap.cpp:1158-1190, `_raster_face`).  `bAddCorners` (the SGM path's
    variant with the image corners as extra support points) is not implemented."""
    w, h = (int(v) for v in cams.size[ID])
    depthMap = np.zeros((h, w), f32)
    normalMap = np.zeros((h, w, 3), f32)
    if len(points) == 0:
        return depthMap, normalMap, 1e-1, 1e+2
    K = cams.K[ID]
    X = scene.vertices[points]
    if opt.nMinViewsTrustPoint < 2:
        camX = (X.astype(np.float64) - cams.C[ID]) @ cams.R[ID].T
        px = np.floor(K[0, 0] * camX[:, 0] / camX[:, 2] + K[0, 2] + 0.5).astype(np.int64)    # ROUND2INT
        py = np.floor(K[1, 1] * camX[:, 1] / camX[:, 2] + K[1, 2] + 0.5).astype(np.int64)
        d = camX[:, 2].astype(f32)
        for x, y, z in zip(px, py, d):                                   # later points overwrite earlier ones, as in the reference
            y0, y1, x0, x1 = max(y - 2, 0), min(y + 2, h - 1), max(x - 2, 0), min(x + 2, w - 1)
            if y1 >= y0 and x1 >= x0:                                    # (a projection outside the image splats nothing)
                depthMap[y0:y1 + 1, x0:x1 + 1] = z
        return depthMap, normalMap, float(d.min() * f32(0.9)), float(d.max() * f32(1.1))
    if opt.bAddCorners and avg_depth is None:
        raise ValueError("bAddCorners needs the image's average depth (select_neighbor_views returns it)")
    proj, z, vert, faces = triangulate_points(K, cams.P[ID], X, w, h, avg_depth if opt.bAddCorners else None)
    normals = np.zeros_like(vert)
    if len(faces):
        f0, f1, f2 = faces[:, 0], faces[:, 1], faces[:, 2]
        t = np.cross(vert[f1] - vert[f0], vert[f2] - vert[f0]).astype(f32)               # Mesh::ComputeNormalVertices, Mesh.cpp:356-371
        for f in (f0, f1, f2):
            np.add.at(normals, f, t)
        nrm = np.sqrt((normals.astype(np.float64) ** 2).sum(1))
        inv = np.where(nrm > 0, 1.0 / np.maximum(nrm, 1e-300), 0.0)          # cv::normalize: v * (1/|v|), the norm in double
        normals = (normals.astype(np.float64) * inv[:, None]).astype(f32)
    dMin, dMax = float(z.min() * f32(0.9)), float(z.max() * f32(1.1))            # the corners do not count (depthBounds, DepthMap.cpp:1043-1046)
    if not opt.bInitSparse:
        if not len(faces):                                                                 # no face: nothing is rasterised
            return depthMap, normalMap, dMin, dMax
        # dense interpolation, DepthMap.cpp:1158-1190: rasterise every mesh face (TImage::RasterizeTriangleBary, libs/Common/Types.inl:2629-2669) with
        # perspective-correct barycentric depth and normal.  Faces in a canonical order (the reference's is CGAL's; only pixels exactly on a shared edge see it).
        for fa in faces:
            _raster_face(proj[fa], vert[fa, 2], normals[fa], depthMap, normalMap)
        return depthMap, normalMap, dMin, dMax
    ix = np.floor(proj).astype(np.int64)
    for (x, y), zz, n in zip(ix, vert[:, 2], normals):
        for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
            ax, ay = x + dx, y + dy
            if 0 <= ax < w and 0 <= ay < h:
                depthMap[ay, ax] = zz
                normalMap[ay, ax] = n
    return depthMap, normalMap, dMin, dMax


def triangulate_points(K, P, X, w, h, avg_depth=None):
    """`TriangulatePointsDelaunay` (libs/MVS/DepthMap.cpp:1019-1115): the 2-D Delaunay mesh of the projections of the sparse points `X`.

    -> (proj (n,2) f32, z (n_points,) f32 depths of the points, vert (n,3) f32 camera-space vertices, faces (m,3) in a canonical order).
    With `avg_depth` (the reference's bAddCorners) the four image corners are appended as vertices, each at the depth where its viewing ray meets
    the planes of the (up to 3, nearest first) faces across the edges opposite to it, weighted by inverse distance (`:1050-1107`).
    The reference triangulates with CGAL; Qhull gives the same triangulation except for co-circular point sets."""
    q = (np.asarray(X, np.float64) @ P[:, :3].T + P[:, 3]).astype(f32)                   # ProjectPointP3<float>
    z = q[:, 2]
    proj = np.stack([q[:, 0] / z, q[:, 1] / z], 1)
    i2c = lambda x, y, d: np.array([f32((x - K[0, 2]) * d / K[0, 0]), f32((y - K[1, 2]) * d / K[1, 1]), f32(d)], f32)   # TransformPointI2C, Camera.h:338-344
    vert = np.stack([((proj[:, 0] - K[0, 2]) * z / K[0, 0]).astype(f32), ((proj[:, 1] - K[1, 2]) * z / K[1, 1]).astype(f32), z], 1)
    n = len(z)
    corners = avg_depth is not None and n >= 3
    if corners:
        cxy = np.array([(0, 0), (w - 1, 0), (0, h - 1), (w - 1, h - 1)], f32)
        proj = np.concatenate([proj, cxy]); vert = np.concatenate([vert, np.stack([i2c(x, y, f32(avg_depth)) for x, y in cxy])])
    if len(proj) < 3:
        return proj, z, vert, np.zeros((0, 3), np.int64)
    from scipy.spatial import Delaunay
    tri = Delaunay(proj.astype(np.float64)).simplices.astype(np.int64)
    a, b, c = proj[tri[:, 0]].astype(np.float64), proj[tri[:, 1]].astype(np.float64), proj[tri[:, 2]].astype(np.float64)
    ccw = ((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])) > 0
    tri = np.where(ccw[:, None], tri, tri[:, ::-1])                 # CGAL faces are counter-clockwise ...
    faces = tri[:, ::-1]                                            # ... and the mesh stores them reversed (DepthMap.cpp:1110)
    rot = np.argmin(faces, 1)                                       # canonical order: smallest index first (keeps the winding), then sorted
    faces = np.stack([faces[np.arange(len(faces)), (rot + k) % 3] for k in range(3)], 1)
    faces = faces[np.lexsort((faces[:, 2], faces[:, 1], faces[:, 0]))]
    if corners:
        lo, hi = z.min(), z.max()
        edges = {}
        for fi, fa in enumerate(faces):
            for k in range(3):
                edges.setdefault(frozenset((int(fa[k]), int(fa[(k + 1) % 3]))), []).append(fi)
        for ci in range(n, n + 4):
            posA = proj[ci].astype(np.float64)
            d = vert[ci].astype(np.float64); d = d / np.linalg.norm(d)                       # Ray3d(0, normalized(vertex))
            top = []                                                                           # (score, depth), best three
            for fi in np.nonzero((faces == ci).any(1))[0]:
                opp = frozenset(int(v) for v in faces[fi] if v != ci)
                nbs = [g for g in edges[opp] if g != fi]
                if not nbs:                                                                    # hull edge: the neighbour is the infinite face
                    continue
                fb = faces[nbs[0]]
                p0, p1, p2 = vert[fb].astype(np.float64)
                nrm = np.cross(p1 - p0, p2 - p0); nn = np.linalg.norm(nrm)
                if nn == 0:
                    continue
                nrm = nrm / nn
                Vd = nrm @ d
                t = 0.0 if Vd == 0 else (nrm @ p0) / Vd                                        # TRay::IntersectsDist, libs/Common/Ray.inl:600-610
                zB = d[2] * t
                if zB <= 0:
                    continue
                posB = (proj[fb[0]].astype(np.float64) + proj[fb[1]].astype(np.float64) + proj[fb[2]].astype(np.float64)) / 3.0
                dist = np.linalg.norm(posB - posA)
                top.append((f32(1.0) / f32(dist), min(max(f32(zB), lo), hi)))
            top.sort(key=lambda e: -e[0]); top = top[:3]
            if top:                                                 # (the reference asserts three; fewer can only happen in degenerate meshes)
                sc = np.array([e[0] for e in top], f32); dp = np.array([e[1] for e in top], f32)
                sc = sc * (f32(1) / sc.sum(dtype=f32))
                depth = f32(0)
                for s_, d_ in zip(sc, dp):
                    depth = f32(depth + f32(s_ * d_))
                vert[ci] = i2c(proj[ci, 0], proj[ci, 1], depth)
    return proj, z, vert, faces


def triangulate_points_depth_map(K, P, X, w, h, avg_depth=None, sparse=False):
    """The depth-only `TriangulatePoints2DepthMap` (libs/MVS/DepthMap.cpp:1194-1251) that seeds the SGM path (SemiGlobalMatcher.cpp:608-618 calls
    it with corners, dense).  -> (depthMap, dMin, dMax): dMin/dMax the depth bounds of the points themselves."""
    proj, z, vert, faces = triangulate_points(K, P, X, w, h, avg_depth)
    depthMap = np.zeros((h, w), f32)
    if sparse:
        ix = np.floor(proj).astype(np.int64)
        for (x, y), zz in zip(ix, vert[:, 2]):
            for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
                if 0 <= x + dx < w and 0 <= y + dy < h:
                    depthMap[y + dy, x + dx] = zz
    else:
        for fa in faces:
            _raster_face(proj[fa], vert[fa, 2], None, depthMap, None)
    return depthMap, (float(z.min()) if len(z) else 3.4028234663852886e+38), (float(z.max()) if len(z) else 0.0)


def _raster_face(v, zs, ns, depthMap, normalMap):
    """One face of the dense initialisation: RasterizeTriangleBary + the RasterDepth functor of TriangulatePoints2DepthMap (DepthMap.cpp:1159-1187)."""
    h, w = depthMap.shape
    v1, v2, v3 = v
    mnx, mxx = min(v1[0], v2[0], v3[0]), max(v1[0], v2[0], v3[0]); mny, mxy = min(v1[1], v2[1], v3[1]), max(v1[1], v2[1], v3[1])
    if mxx < 0 or mnx > f32(w - 1) or mxy < 0 or mny > f32(h - 1):
        return
    x0, x1 = max(int(np.floor(mnx)), 0), min(int(np.ceil(mxx)), w - 1)
    y0, y1 = max(int(np.floor(mny)), 0), min(int(np.ceil(mxy)), h - 1)
    # EdgeFunction, Util.inl:602-604: (x2 - x0).cross(x1 - x0) -- the differences in float, the cross product in DOUBLE (cv::Point_::cross returns double), then back to float
    edge = lambda a, b, c: f32(float(f32(c[0] - a[0])) * float(f32(b[1] - a[1])) - float(f32(c[1] - a[1])) * float(f32(b[0] - a[0])))
    area = edge(v1, v2, v3)
    if area <= 0:
        return
    inv = f32(f32(1) / area)
    z0, z1, z2 = zs
    for y in range(y0, y1 + 1):
        for x in range(x0, x1 + 1):
            p = (f32(x), f32(y))
            b1 = f32(edge(v2, v3, p) * inv)
            if b1 < 0:
                continue
            b2 = f32(edge(v3, v1, p) * inv)
            if b2 < 0:
                continue
            b3 = f32(edge(v1, v2, p) * inv)
            if b3 < 0:
                continue
            pb = (f32(f32(b1 * z1) * z2), f32(f32(b2 * z0) * z2), f32(f32(b3 * z0) * z1))                # PerspectiveCorrectBarycentricCoordinates, Util.inl:746-749
            sden = f32(f32(pb[0] + pb[1]) + pb[2]); invd = f32(f32(1) / sden)                                  # TPoint3 / scalar multiplies by the reciprocal
            pb = (f32(invd * pb[0]), f32(invd * pb[1]), f32(invd * pb[2]))
            depthMap[y, x] = f32(f32(f32(pb[0] * z0) + f32(pb[1] * z1)) + f32(pb[2] * z2))
            if ns is None:
                continue
            n = (ns[0] * pb[0]).astype(f32) + (ns[1] * pb[1]).astype(f32)
            n = (n.astype(f32) + (ns[2] * pb[2]).astype(f32)).astype(f32)
            nd = n.astype(np.float64)                                    # cv::normalize: the norm and the product in double (np.float64 operands: a Python float would
            nn = np.sqrt((nd[0] * nd[0] + nd[1] * nd[1]) + nd[2] * nd[2])  # take the float32 operand's type under numpy 2's promotion rules)
            normalMap[y, x] = (nd * (1.0 / nn if nn else 0.0)).astype(f32)


# ---- ignore masks (`--ignore-mask-label`, `--mask-path`) ------------------------------------------------------------------------------------------

def mask_file_name(image_name: str, mask_name: str = "", mask_path: str | None = None) -> str:
    """Where `DepthEstimator::ImportIgnoreMask` looks for an image's segmentation mask (libs/MVS/DepthMap.cpp:301): the name stored in the scene, else the image's path
    with its extension replaced by ".mask.png" (`Util::getFileFullName`: everything before the LAST '.', libs/Common/Util.h:466-469).  `mask_path` is
    `DensifyPointCloud --mask-path` (apps/DensifyPointCloud/DensifyPointCloud.cpp:307-320): <mask_path><file name without directory and extension>.mask.png for every
    image; an image that already names a mask is an error there."""
    if mask_path:
        if mask_name:
            raise ValueError("image %s has non-empty maskName %s" % (image_name, mask_name))
        base = image_name[image_name.rfind("/") + 1:]                      # Util::getFileName, Util.h:476-482
        j = image_name.rfind(".")
        base = image_name[image_name.rfind("/") + 1:j] if j >= 0 else base
        return (mask_path if mask_path.endswith("/") else mask_path + "/") + base + ".mask.png"      # (ensureValidFolderPath appends the separator)
    if mask_name:
        return mask_name
    i = image_name.rfind(".")
    return (image_name[:i] if i >= 0 else image_name) + ".mask.png"


def import_ignore_mask(mask: np.ndarray, size, label: int) -> np.ndarray:
    """`DepthEstimator::ImportIgnoreMask` after the file is read (libs/MVS/DepthMap.cpp:307-320): the 16-bit label image resized to the depth map's `size` = (w, h) with
    INTER_NEAREST, then 1 where the label differs from `label` (pixels to process), 0 where it equals it (ignored) -- the array `PatchMatchHIP.scene_set_mask` takes.
    cv::resize's nearest rule: sx = min(floor(dx * ifx), sw - 1) with ifx = 1 / (dw / sw) in double (resizeNN)."""
    m = np.asarray(mask)
    if m.ndim != 2:
        raise ValueError("a label image is (h, w)")
    m = m.astype(np.uint16)                                              # Image16U
    w, h = int(size[0]), int(size[1])
    sh, sw = m.shape
    ifx, ify = 1.0 / (w / float(sw)), 1.0 / (h / float(sh))
    sx = np.minimum(np.floor(np.arange(w) * ifx).astype(np.int64), sw - 1)
    sy = np.minimum(np.floor(np.arange(h) * ify).astype(np.int64), sh - 1)
    return (m[sy][:, sx] != np.uint16(label & 0xFFFF)).astype(np.uint8)


# ---- normals of a depth map that has none (a .dmap written without them, the SGM fuse mode) --------------------------------------------------------

def estimate_normal_map(K, depth):
    """`MVS::EstimateNormalMap(K, depthMap, normalMap)` (libs/MVS/DepthMap.cpp:1522-1613, the active least-squares branch): per pixel the depth gradient from the
    8-neighbourhood -- neighbours inside the map, with a depth > 0 that differs by less than 3 % of the pixel's (IsDepthSimilar, libs/Common/Util.inl:797-809), at
    least three of them, non-singular 2x2 system in integers -- then the normal normalized((K00*dx, K11*dy, (K02 - x)*dx + (K12 - y)*dy - d)); zero where that fails.
    Float sums in the reference's order (rows of the neighbourhood top to bottom, left to right); cv::normalize = v * (1 / |v|) with the norm and the product in double."""
    d = np.ascontiguousarray(depth, f32)
    H, W = d.shape
    Kf = np.asarray(K, np.float64).astype(f32)
    whxx = np.zeros((H, W), np.int64); whxy = np.zeros((H, W), np.int64); whyy = np.zeros((H, W), np.int64); n = np.zeros((H, W), np.int64)
    wgx = np.zeros((H, W), f32); wgy = np.zeros((H, W), f32)
    pad = np.zeros((H + 2, W + 2), f32); pad[1:-1, 1:-1] = d
    inside = np.zeros((H + 2, W + 2), bool); inside[1:-1, 1:-1] = True
    pos = d > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        for y in (-1, 0, 1):
            for x in (-1, 0, 1):
                if x == 0 and y == 0:
                    continue
                wi = pad[1 + y:1 + y + H, 1 + x:1 + x + W]
                ok = inside[1 + y:1 + y + H, 1 + x:1 + x + W] & pos & (wi > 0) & ((np.abs(d - wi) / d) < f32(0.03))
                whxx += ok * (x * x); whxy += ok * (x * y); whyy += ok * (y * y); n += ok
                diff = (wi - d).astype(f32)
                wgx = np.where(ok, (wgx + diff * f32(x)).astype(f32), wgx)
                wgy = np.where(ok, (wgy + diff * f32(y)).astype(f32), wgy)
        det = whxx * whyy - whxy * whxy
        good = pos & (n >= 3) & (det != 0)
        inv = (f32(1) / np.where(good, det, 1).astype(f32)).astype(f32)
        wx = ((whyy.astype(f32) * wgx - whxy.astype(f32) * wgy) * inv).astype(f32)
        wy = (((-whxy).astype(f32) * wgx + whxx.astype(f32) * wgy) * inv).astype(f32)
        xs = np.arange(W, dtype=f32)[None, :]; ys = np.arange(H, dtype=f32)[:, None]
        nx = (Kf[0, 0] * wx).astype(f32); ny = (Kf[1, 1] * wy).astype(f32)
        nz = ((((Kf[0, 2] - xs) * wx).astype(f32) + ((Kf[1, 2] - ys) * wy).astype(f32)).astype(f32) - d).astype(f32)
        v = np.stack([nx, ny, nz], -1).astype(np.float64)
        nv = np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2])
        a = np.where(nv != 0, 1.0 / np.where(nv != 0, nv, 1.0), 0.0)
        out = (v * a[..., None]).astype(f32)
    out[~good] = 0
    return out

"""Depth maps through the SGM path (`DensifyPointCloud --fusion-mode -1/-2`): for a reference image and each of its best neighbours
`SemiGlobalMatcher::Match(scene, idxImage, numNeighbors, minResolution)` (reference `libs/MVS/SemiGlobalMatcher.cpp:529-736`) -- rectify the pair,
run the tSGM loop, keep the disparity data -- then `SemiGlobalMatcher::Fuse` (`:738-859`): project every pair's disparities into the reference
image and fuse the pair depth maps per pixel.  Orchestration only: the steps are `rectify.py`, `tsgm.py` and the calls of include/sgmhip.h
behind the backend object (see tsgm.py)."""
from __future__ import annotations

import numpy as np

from . import rectify, tsgm


_SRGB = None


def _srgb_table():
    """g_ptrsRGB82RGBf (libs/Common/Types.inl:1595-1607): sRGB2RGB(i / 255.f) in float, with the C library's powf as the reference uses."""
    global _SRGB
    if _SRGB is None:
        import ctypes, ctypes.util
        libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        libm.powf.restype = ctypes.c_float; libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
        f = np.float32
        t = np.zeros(256, f)
        for i in range(256):
            x = f(i) / f(255)
            t[i] = x * (f(1) / f(12.92)) if x <= f(0.04045) else f(libm.powf(f((x + f(0.055)) * (f(1) / f(1.055))), f(2.4)))
        _SRGB = t
    return _SRGB


def to_gray_linear(bgr):
    """`imageColor.toGray(imageGray, COLOR_BGR2GRAY, bNormalize = true, bSRGB = true)` of the SGM path (libs/MVS/SemiGlobalMatcher.cpp:579-582;
    TImage::toGray, libs/Common/Types.inl:2377-2425): every 8-bit channel through the sRGB -> linear table, then 0.114 B + 0.587 G + 0.299 R."""
    f = np.float32
    t = _srgb_table()
    return (f(0.114) * t[bgr[..., 0]] + f(0.587) * t[bgr[..., 1]]) + f(0.299) * t[bgr[..., 2]]


def world_to_image3(K, R, C, X):
    """Camera::TransformPointW2I3 (libs/MVS/Camera.h:396-399): (x, y, depth) as float32."""
    cx = (np.asarray(X, np.float64) - C) @ R.T
    return np.stack([K[0, 2] + K[0, 0] * cx[:, 0] / cx[:, 2], K[1, 2] + K[1, 1] * cx[:, 1] / cx[:, 2], cx[:, 2]], 1).astype(np.float32)


def compute_resize(size, scale):
    """Image8U::computeResize(size, scale) (libs/Common/Types.inl:2446-2449): cvRound of the scaled extents."""
    return int(np.rint(size[0] * scale)), int(np.rint(size[1] * scale))


def match_pair(be, bgrA, camA, bgrB, camB, shared_points, min_resolution=320, subpixel_steps=4, seed_depth=None):
    """One pair of `Match(scene, ...)`: cam = (K, R, C).  Returns the `.dimap` content: dict(disparity, cost, H, Q, image_size, subpixel_steps)
    or None if the pair cannot be rectified.

    seed_depth: callable (w, h) -> depth map of image A at that size, the rough estimate from the sparse points that the reference makes with
    `TriangulatePoints2DepthMap(..., bAddCorners = true)` (SemiGlobalMatcher.cpp:608-618; `views.triangulate_points_depth_map` or
    `mvsf_triangulate_depth_map`); it becomes the first level's initial disparities through Depth2DisparityMap (`:619-625`).  Without it the
    first level searches the default range around zero."""
    p1 = world_to_image3(*camA, shared_points); p2 = world_to_image3(*camB, shared_points)
    r = rectify.stereo_rectify_images(bgrA, *camA, bgrB, *camB, p1, p2)
    if r is None:
        return None
    k = tsgm.compute_scale(r["size"][0], r["size"][1], min_resolution)
    f = 1 << k
    w, h = r["size"][0] // f * f, r["size"][1] // f * f           # the loop's 8-bit resampler wants multiples of 2^levels: crop right / bottom
    lb, rb = r["rect1"][:h, :w].copy(), r["rect2"][:h, :w].copy()
    init = None
    if seed_depth is not None:
        s = 0.5 / f                                               # scale * 0.5
        depth = seed_depth(*compute_resize((bgrA.shape[1], bgrA.shape[0]), s))
        H2, Q2 = rectify.scale_stereo_rectification(r["H"], r["Q"], s)
        hw, hh = compute_resize((w // f, h // f), 0.5)
        init = be.Depth2DisparityMap(depth, np.linalg.inv(H2), np.linalg.inv(Q2), 1, (hw - 2 * tsgm.HW, hh - 2 * tsgm.HW))
    args = (lb, to_gray_linear(lb), rb, to_gray_linear(rb), r["mask1"][:h, :w].copy(), r["mask2"][:h, :w].copy())
    if hasattr(be, "tsgm_match"):                                 # the device runs the whole loop in one resident call (sgmhip_tsgm_match)
        disp, cost, _ = be.tsgm_match(args[0], args[2], args[1], args[3], args[4], args[5], min_resolution=min_resolution, init_left_disparity=init,
                                      subpixel_steps=subpixel_steps)
    else:
        disp, cost, _ = tsgm.tsgm_match(be, *args, min_resolution=min_resolution, init_left_disparity=init, subpixel_steps=subpixel_steps)
    return dict(disparity=disp, cost=cost, H=r["H"], Q=r["Q"], image_size=(bgrA.shape[1], bgrA.shape[0]), subpixel_steps=subpixel_steps, seeded=init is not None)


def fuse_pairs(be, pairs, min_views=2):
    """`SemiGlobalMatcher::Fuse` for pairs that all have the reference image on the left: ProjectDisparity2DepthMap per pair, then the per-pixel
    cluster fusion.  -> (depth map, confidence map) of the reference image."""
    if hasattr(be, "fuse_disparities") and pairs:                 # the device does it in one resident call (sgmhip_fuse_disparities)
        dep, cf, _ = be.fuse_disparities(pairs, pairs[0]["image_size"], min_views)
        return dep, cf
    deps, rgs, cfs = [], [], []
    for p in pairs:
        ok, dep, rg, cf = be.ProjectDisparity2DepthMap(p["disparity"], p["cost"], p["Q"], p["subpixel_steps"], p["image_size"])
        if ok:
            deps.append(dep); rgs.append(rg); cfs.append(cf)
    if not deps:
        w, h = pairs[0]["image_size"] if pairs else (0, 0)
        return np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    return be.FusePairs(deps, rgs, cfs, min_views)


# ---- scene level: `DensifyPointCloud --fusion-mode -1` (disparity maps of every pair) and `-2` (fuse them into depth maps) ------------------------------------------

def _matx(a, b):
    """cv::Matx product: c(i, j) = sum_k a(i, k) b(k, j), accumulated left to right from 0 in double."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    c = np.zeros((a.shape[0], b.shape[1]))
    for i in range(a.shape[0]):
        for j in range(b.shape[1]):
            s = 0.0
            for k in range(a.shape[1]):
                s += a[i, k] * b[k, j]
            c[i, j] = s
    return c


def swapped_pair_q(Q, cam_left, cam_right):
    """`SemiGlobalMatcher::Fuse` when only the pair (right, left) is on disk (libs/MVS/SemiGlobalMatcher.cpp:767-777): that file's Q puts a disparity of the rectified RIGHT
    image into the right camera's image space; `P * invK * Q` carries it on into the LEFT image -- P = K_left [R | -R C] with (R, C) the pose of the left camera relative to
    the right one (ComputeRelativePose, libs/Common/Util.inl:39-42), invK = the right camera's inverse intrinsics (Camera::InvK)."""
    (Kl, Rl, Cl), (Kr, Rr, Cr) = [tuple(np.asarray(x, np.float64) for x in cam) for cam in (cam_left, cam_right)]
    poseR = _matx(Rl, Rr.T)
    poseC = _matx(Rr, (Cl - Cr)[:, None])[:, 0]
    M = _matx(Kl, poseR)
    P = np.eye(4); P[:3, :3] = M; P[:3, 3] = _matx(M, (-poseC)[:, None])[:, 0]          # AssembleProjectionMatrix (libs/MVS/Camera.cpp:173-180)
    invK = np.eye(4)
    invK[0, 0] = 1.0 / Kr[0, 0]; invK[1, 1] = 1.0 / Kr[1, 1]; invK[0, 2] = -Kr[0, 2] * invK[0, 0]; invK[1, 2] = -Kr[1, 2] * invK[1, 1]   # Camera::InvK, libs/MVS/Camera.h:175-188
    return _matx(_matx(P, invK), Q)


def pair_file_name(a: int, b: int) -> str:
    return "%04u_%04u.dimap" % (a, b)


def _listed(neighbors, n_views, f_min_score_ratio, f_min_score):
    """The neighbours `Match` / `Fuse` walk (:532-540, :742-759): best first, cut at numNeighbors and at the score threshold."""
    if not len(neighbors):
        return []
    f_min = max(np.float32(neighbors[0]["score"]) * np.float32(f_min_score_ratio), np.float32(f_min_score))
    out = []
    for k, nb in enumerate(neighbors):
        if (n_views and k >= n_views) or nb["score"] < f_min:
            break
        out.append(int(nb["ID"]))
    return out


def match_scene(be, sc, cams, bgr, neighbors, out_dir, n_views=0, f_min_score_ratio=0.03, f_min_score=2.0, min_resolution=320, subpixel_steps=4, avg_depth=None):
    """`SemiGlobalMatcher::Match(scene, idxImage, numNeighbors, minResolution)` for every image (DenseReconstruction with nFusionMode == -1, SceneDensify.cpp:2046-2048):
    each image against its listed neighbours -- a pair whose disparity file exists already, in either order, is skipped (:543-545) -- rectified, matched through the tSGM
    loop seeded from the image's sparse points (`match_pair`), written as `<left>_<right>.dimap`.
    sc: mvsi.Scene; cams: views.Cameras; bgr: the colour images; neighbors: {image: its whole scored list (Image::neighbors)}; avg_depth: {image: Image::avgDepth} for the
    seed's corner points.  Returns the pairs written."""
    import os
    from . import dmap, views
    os.makedirs(out_dir, exist_ok=True)
    own = np.repeat(np.arange(len(sc.vertices)), np.diff(sc.vertex_view_start)); ids = sc.vertex_views["image_id"]
    seen = {}

    def sees(i):
        if i not in seen:
            s = np.zeros(len(sc.vertices), bool); s[own[ids == i]] = True; seen[i] = s
        return seen[i]

    cam = lambda i: (cams.K[i], cams.R[i], cams.C[i])
    done = []
    for i in sorted(neighbors):
        for j in _listed(neighbors[i], n_views, f_min_score_ratio, f_min_score):
            if os.path.exists(os.path.join(out_dir, pair_file_name(i, j))) or os.path.exists(os.path.join(out_dir, pair_file_name(j, i))):
                continue
            pts = np.nonzero(sees(i))[0]

            def seed(w, h, i=i, pts=pts):
                K, R, C, _, _ = sc.camera(i, (w, h))
                P = K @ np.hstack([R, -(R @ C)[:, None]])
                return views.triangulate_points_depth_map(K, P, sc.vertices[pts], w, h, avg_depth=None if avg_depth is None else avg_depth[i])[0]

            p = match_pair(be, bgr[i], cam(i), bgr[j], cam(j), sc.vertices[sees(i) & sees(j)], min_resolution=min_resolution, subpixel_steps=subpixel_steps,
                           seed_depth=seed if (avg_depth is not None and len(pts) >= 3) else None)
            if p is None:
                continue                                         # the pair cannot be rectified (:566-567)
            dmap.save_dimap(os.path.join(out_dir, pair_file_name(i, j)), p["image_size"], p["H"], p["Q"], p["subpixel_steps"], p["disparity"], p["cost"])
            done.append((i, j))
    return done


def fuse_scene(be, cams, sizes, neighbors, pair_dir, n_views=0, f_min_score_ratio=0.03, f_min_score=2.0, min_views=2, estimate_normals=True):
    """`SemiGlobalMatcher::Fuse` for every image (DenseReconstruction with nFusionMode == -2, SceneDensify.cpp:2049-2057): the disparity files of the image's listed
    neighbours -- stored as (image, neighbour), or as (neighbour, image) with Q carried over (`swapped_pair_q`) -- projected into the image and fused per pixel; normals from
    the depths (`views.estimate_normal_map`) when `estimate_normals` (nEstimateNormals == 2).  sizes: {image: (w, h)}.
    Returns {image: (depth, normal or None, conf)}; the depth range of such a map is (ZEROTOLERANCE, FLT_MAX), :2056."""
    import os
    from . import dmap, views
    cam = lambda i: (cams.K[i], cams.R[i], cams.C[i])
    out = {}
    for i in sorted(neighbors):
        pairs = []
        for j in _listed(neighbors[i], n_views, f_min_score_ratio, f_min_score):
            direct, swapped = os.path.join(pair_dir, pair_file_name(i, j)), os.path.join(pair_dir, pair_file_name(j, i))
            if os.path.exists(direct):
                g = dmap.load_dimap(direct)
            elif os.path.exists(swapped):
                g = dmap.load_dimap(swapped)
                g["Q"] = swapped_pair_q(g["Q"], cam(i), cam(j))
            else:
                continue                                         # "no disparity-data file found for image pair"
            pairs.append(dict(disparity=g["disparity"], cost=g["cost"], Q=g["Q"], subpixel_steps=g["subpixel_steps"], image_size=tuple(sizes[i])))
        if not pairs:
            w, h = sizes[i]
            out[i] = (np.zeros((h, w), np.float32), None, np.zeros((h, w), np.float32))
            continue
        depth, conf = fuse_pairs(be, pairs, min_views)
        out[i] = (depth, views.estimate_normal_map(cams.K[i], depth) if estimate_normals else None, conf)
    return out


class DeviceBackend:
    """The device behind `match_pair` / `fuse_pairs`: `sgm.SemiGlobalMatcherHIP` has every step of the interface (and the resident loop and fusion, `tsgm_match`,
    `fuse_disparities`) except the float area resampler of the image pyramid, which the PatchMatch library provides (`PatchMatchHIP.resize`)."""

    def __init__(self, matcher, engine):
        self.m, self.e = matcher, engine

    def resize_area_f32(self, img, f):
        return self.e.resize(0, img, f)

    def __getattr__(self, name):
        return getattr(self.m, name)


def dense_reconstruction(be, mvs_in, out_dir, fusion_mode, opt=None, image_loader=None, min_resolution=320):
    """The SGM modes of `Scene::DenseReconstruction` (libs/MVS/SceneDensify.cpp:1655-1750, :2042-2058): `fusion_mode` -1 writes the disparity files of every image's
    pairs into `out_dir`; -2 fuses the files found there into depth maps and writes `depthNNNN.dmap` (depth, normals when nEstimateNormals == 2, confidence; depth range
    (ZEROTOLERANCE, FLT_MAX) as at :2056) -- what a later `--fusion-mode 0` run fuses into the cloud.  be: a backend (`DeviceBackend(SemiGlobalMatcherHIP(dev),
    PatchMatchHIP(dev))`); opt: an `optdense.OptDense` (nNumViews, view-score cuts, resolution level); image_loader as in `densify.load_scene`.
    Returns the pairs written (-1) or {image: (depth, normal, conf)} (-2)."""
    import os
    from . import densify, dmap, mvsi, optdense, views
    if fusion_mode not in (-1, -2):
        raise ValueError("the SGM path is fusion modes -1 and -2")
    opt = opt or optdense.defaults()
    sv = densify.load_scene(mvs_in, opt=opt, image_loader=image_loader)
    if sv.alias_of or len(set(map(tuple, sv.sizes))) > 1:
        raise NotImplementedError("the SGM driver takes scenes whose images share one size")
    sc = mvsi.load(mvs_in)
    cams = views.Cameras(sc, sv.sizes)
    nbs = {i: sv.all_view_scores[i] for i in sv.ids if len(sv.all_view_scores.get(i, ()))}
    args = dict(n_views=int(opt.nNumViews), f_min_score_ratio=float(opt.fViewMinScoreRatio), f_min_score=float(opt.fViewMinScore))
    if fusion_mode == -1:
        return match_scene(be, sc, cams, sv.bgr, nbs, out_dir, min_resolution=min_resolution, avg_depth=sv.avg_depth, **args)
    sizes = {i: tuple(sv.sizes[i]) for i in nbs}
    fused = fuse_scene(be, cams, sizes, nbs, out_dir, min_views=2, estimate_normals=int(opt.nEstimateNormals) == 2, **args)
    for i, (depth, normal, conf) in fused.items():
        ids = [i] + [int(v) for v in sv.neighbors[i]]
        dmap.save(os.path.join(out_dir, dmap.depth_file_name(i)), sv.names[i], ids, sizes[i], sv.K[i], sv.R[i], sv.C[i], 1e-4, float(np.finfo(np.float32).max),      # ZEROTOLERANCE<float>() = FZERO_TOLERANCE (libs/Common/Types.h:579), FLT_MAX
                  depth, normal, conf)
    return fused

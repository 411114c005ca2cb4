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

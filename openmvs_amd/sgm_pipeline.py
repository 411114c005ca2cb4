"""Depth maps through the SGM path (`DensifyPointCloud --fusion-mode -1/-2`): for a reference image and each of its best neighbours
`SemiGlobalMatcher::Match(scene, idxImage, numNeighbors, minResolution)` (reference `libs/MVS/SemiGlobalMatcher.cpp:529-736`) -- rectify the pair,
run the tSGM loop, keep the disparity data -- then `SemiGlobalMatcher::Fuse` (`:738-859`): project every pair's disparities into the reference
image and fuse the pair depth maps per pixel.  Orchestration only: the steps are `rectify.py`, `tsgm.py` and the calls of include/sgmhip.h
behind the backend object (see tsgm.py)."""
from __future__ import annotations

import numpy as np

from . import rectify, tsgm


def to_gray_linear(bgr):
    """0.114 B + 0.587 G + 0.299 R of the normalised channels.  (The reference converts with bSRGB = true for this path,
    libs/MVS/SemiGlobalMatcher.cpp:571-576: an sRGB -> linear transfer before the weighted sum; not applied here.)"""
    f = np.float32
    return (f(0.114) * (bgr[..., 0].astype(f) / f(255)) + f(0.587) * (bgr[..., 1].astype(f) / f(255))) + f(0.299) * (bgr[..., 2].astype(f) / f(255))


def world_to_image3(K, R, C, X):
    """Camera::TransformPointW2I3 (libs/MVS/Camera.h:396-399): (x, y, depth) as float32."""
    cx = (np.asarray(X, np.float64) - C) @ R.T
    return np.stack([K[0, 2] + K[0, 0] * cx[:, 0] / cx[:, 2], K[1, 2] + K[1, 1] * cx[:, 1] / cx[:, 2], cx[:, 2]], 1).astype(np.float32)


def match_pair(be, bgrA, camA, bgrB, camB, shared_points, min_resolution=320, subpixel_steps=4):
    """One pair of `Match(scene, ...)`: cam = (K, R, C).  Returns the `.dimap` content: dict(disparity, cost, H, Q, image_size, subpixel_steps)
    or None if the pair cannot be rectified."""
    p1 = world_to_image3(*camA, shared_points); p2 = world_to_image3(*camB, shared_points)
    r = rectify.stereo_rectify_images(bgrA, *camA, bgrB, *camB, p1, p2)
    if r is None:
        return None
    k = tsgm.compute_scale(r["size"][0], r["size"][1], min_resolution)
    f = 1 << k
    w, h = r["size"][0] // f * f, r["size"][1] // f * f           # the loop's 8-bit resampler wants multiples of 2^levels: crop right / bottom
    lb, rb = r["rect1"][:h, :w].copy(), r["rect2"][:h, :w].copy()
    disp, cost, _ = tsgm.tsgm_match(be, lb, to_gray_linear(lb), rb, to_gray_linear(rb), r["mask1"][:h, :w].copy(), r["mask2"][:h, :w].copy(),
                                    min_resolution=min_resolution, subpixel_steps=subpixel_steps)
    return dict(disparity=disp, cost=cost, H=r["H"], Q=r["Q"], image_size=(bgrA.shape[1], bgrA.shape[0]), subpixel_steps=subpixel_steps)


def fuse_pairs(be, pairs, min_views=2):
    """`SemiGlobalMatcher::Fuse` for pairs that all have the reference image on the left: ProjectDisparity2DepthMap per pair, then the per-pixel
    cluster fusion.  -> (depth map, confidence map) of the reference image."""
    deps, rgs, cfs = [], [], []
    for p in pairs:
        ok, dep, rg, cf = be.ProjectDisparity2DepthMap(p["disparity"], p["cost"], p["Q"], p["subpixel_steps"], p["image_size"])
        if ok:
            deps.append(dep); rgs.append(rg); cfs.append(cf)
    if not deps:
        w, h = pairs[0]["image_size"] if pairs else (0, 0)
        return np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    return be.FusePairs(deps, rgs, cfs, min_views)

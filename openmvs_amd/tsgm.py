"""The tSGM coarse-to-fine loop of `SemiGlobalMatcher::Match(scene, ...)` for one rectified image pair
(reference `libs/MVS/SemiGlobalMatcher.cpp:577-706`), on top of the C-ABI steps of include/sgmhip.h.

What is here is the loop body between rectification and export: per pyramid level the two `Match` calls (right->left with ranges from
the flipped previous disparities, then left->right), the consistency checks, the first-level speckle filter and mask extraction, the mask
up-scaling, and the final sub-pixel refinement.  What is not: `Image::StereoRectifyImages` (OpenCV `stereoRectify` + remap) and the initial
disparity map from the sparse points (`TriangulatePoints2DepthMap` with corners + `Depth2DisparityMap`) -- the caller passes rectified images,
their masks, and optionally that initial half-resolution disparity map (without it the first level searches the default range).

The loop is written against a small backend interface so that the very same code runs on the device (`SemiGlobalMatcherHIP`, plus a float
INTER_AREA resampler) and, in the tests, on the CPU oracle.
"""
from __future__ import annotations

import numpy as np

NO_DISP = 32767
HW = 3                       # halfWindowSizeX / Y


def compute_scale(width: int, height: int, min_resolution: int) -> int:
    """The pyramid depth of :577-583: returns k with scale = 1 / 2^k (k = 0: plain SGM)."""
    if not min_resolution:
        return 0
    size0 = max(width, height)
    level = 8                                    # computeMaxResolution(w, h, level = 8, minResolution), libs/Common/Types.inl:2459-2477
    if (size0 >> level) < min_resolution:
        level = 0
        while (size0 >> (level + 1)) >= min_resolution:
            level += 1
    return max(1, level)                         # scale = 1 / max(2, 2^level)


def resize_area_u8(img: np.ndarray, f: int) -> np.ndarray:
    """cv::resize(img, Size(), 1/f, 1/f, INTER_AREA) for an 8-bit image whose size is a multiple of f: OpenCV's integer-factor path -- for
    f = 2 the rounded mean (a+b+c+d+2)>>2 of ResizeAreaFastVec, otherwise saturate_cast<uchar>(sum * (1/f^2)) (round half to even)."""
    if f == 1:
        return img
    H, W = img.shape[:2]
    if W % f or H % f:
        raise NotImplementedError("tsgm: image size %dx%d is not a multiple of %d" % (W, H, f))
    s = img.reshape(H // f, f, W // f, f, -1).astype(np.int32).sum(axis=(1, 3))
    out = ((s + 2) >> 2) if f == 2 else np.rint(s.astype(np.float32) * np.float32(1.0 / (f * f)))
    return out.astype(np.uint8).reshape((H // f, W // f) + img.shape[2:])


def resize_nearest_u8(img: np.ndarray, w: int, h: int) -> np.ndarray:
    """cv::resize(mask, size, INTER_NEAREST)."""
    H, W = img.shape
    ys = np.minimum(np.floor(np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return img[ys][:, xs]


def tsgm_match(be, left_bgr, left_gray, right_bgr, right_gray, left_mask, right_mask, min_resolution=320, init_left_disparity=None,
               n_speckle_size=100, subpixel_mode=6, subpixel_steps=4):
    """Runs the loop; returns (left disparity map at full resolution, refined to `subpixel_steps`, its cost map, the number of levels).

    `be` provides: resize_area_f32(img, f); set_problem / Match / results (as SemiGlobalMatcherHIP); ConsistencyCrossCheck, FilterSpeckles,
    ExtractMask, UpscaleMask, FlipDirection, Disparity2RangeMap, RefineDisparityMap."""
    H, W = left_gray.shape
    k = compute_scale(W, H, min_resolution)
    if k == 0:
        raise NotImplementedError("plain SGM (minResolution = 0) needs the global range of an initial disparity map; only tSGM is driven here")
    left_disp = right_disp = None
    cost = None
    levels = 0
    for lvl in range(k, -1, -1):
        f = 1 << lvl
        lb, rb = resize_area_u8(left_bgr, f), resize_area_u8(right_bgr, f)
        lg = left_gray if f == 1 else be.resize_area_f32(left_gray, f)
        rg = right_gray if f == 1 else be.resize_area_f32(right_gray, f)
        h, w = lg.shape
        vw, vh = w - 2 * HW, h - 2 * HW
        first = left_disp is None
        if first:
            hw2, hh2 = int(np.rint(w * 0.5)), int(np.rint(h * 0.5))                      # Image8U::computeResize(size, 0.5)
            if init_left_disparity is not None:
                left_disp = np.ascontiguousarray(init_left_disparity, np.int16)
                if left_disp.shape != (hh2 - 2 * HW, hw2 - 2 * HW):
                    raise ValueError("initial disparity map must be %dx%d" % (hw2 - 2 * HW, hh2 - 2 * HW))
            else:
                left_disp = np.full((hh2 - 2 * HW, hw2 - 2 * HW), NO_DISP, np.int16)
            lm = resize_nearest_u8(left_mask, w, h)[HW:HW + vh, HW:HW + vw].copy()        # :612-617
            rm = resize_nearest_u8(right_mask, w, h)[HW:HW + vh, HW:HW + vw].copy()
        else:
            lm = be.UpscaleMask(lm, (vw, vh)); rm = be.UpscaleMask(rm, (vw, vh))          # :619-621
        a, b = (11, 33) if first else (5, 7)
        right_disp = be.FlipDirection(left_disp)                                           # :626-627
        px, n, mx = be.Disparity2RangeMap(right_disp, rm, a, b)
        be.set_problem(rb, rg, lg, px, n, mx); be.Match()                                  # Match(rightDataLevel, leftDataLevel, ...), :654
        right_disp, _ = be.results()
        px, n, mx = be.Disparity2RangeMap(left_disp, lm, a, b)                             # :657
        be.set_problem(lb, lg, rg, px, n, mx); be.Match()                                  # :667
        left_disp, cost = be.results()
        if first:                                                                          # :680-690
            left_disp = be.ConsistencyCrossCheck(left_disp, right_disp)
            right_disp = be.ConsistencyCrossCheck(right_disp, left_disp)
            left_disp = be.FilterSpeckles(left_disp, n_speckle_size, 5)
            right_disp = be.FilterSpeckles(right_disp, n_speckle_size, 5)
            lm = be.ExtractMask(left_disp, lm); rm = be.ExtractMask(right_disp, rm)
        else:
            left_disp = be.ConsistencyCrossCheck(left_disp, right_disp)                    # :693
        levels += 1
    # RefineDisparityMap works on the resident sums of the last Match with the cross-checked map (:699): push the checked map back first
    be.set_disparity(left_disp)
    be.RefineDisparityMap(subpixel_mode, subpixel_steps)
    refined, cost = be.results()
    return refined, cost, levels

"""Stereo rectification of an image pair for the SGM path: `Image::StereoRectifyImages` (reference `libs/MVS/Image.cpp:224-345`) with
`Camera::StereoRectifyFusiello` (`libs/MVS/Camera.cpp:359-405`), `SetStereoRectificationROI` (`:414-436`) and the `RECTIFY` helpers (`:236-278`).

The geometry (R1, R2, K1, K2, baseline t, the rectifying homographies H1, H2, the rectified size and the disparity-to-depth matrix Q) is the
reference's arithmetic in double.  The two pixel operations are OpenCV calls in the reference and are only approximated here:
`cv::warpPerspective` (bilinear, constant border) is a plain float bilinear resample rounded to 8 bits (OpenCV interpolates in fixed point with
1/32-pixel coordinates), and the validity mask that the reference draws with `cv::drawContours` is the exact set of pixels whose centre maps
inside the source image.  Neither affects the geometry; both change the rectified pixel values by at most a grey level or a border pixel.
"""
from __future__ import annotations

import numpy as np


def _inv_k(K):
    """Camera::InvK (libs/MVS/Camera.h:176-185)."""
    o = np.eye(3)
    o[0, 0] = 1.0 / K[0, 0]; o[1, 1] = 1.0 / K[1, 1]; o[0, 2] = -K[0, 2] * o[0, 0]; o[1, 2] = -K[1, 2] * o[1, 1]
    return o


def stereo_rectify_fusiello(K1, R1w, C1, K2, R2w, C2):
    """-> (R1, R2, K1r, K2r, t): rotations from each camera to the common rectified frame, new camera matrices, signed baseline."""
    K1 = np.asarray(K1, np.float64); K2 = np.asarray(K2, np.float64); R1w = np.asarray(R1w, np.float64); R2w = np.asarray(R2w, np.float64)
    C1 = np.asarray(C1, np.float64); C2 = np.asarray(C2, np.float64)
    poseR = R2w @ R1w.T; poseC = R1w @ (C2 - C1)                    # ComputeRelativePose, libs/Common/Util.inl:39-42
    v1 = C2 - C1                                                      # new x axis: the baseline
    v2 = np.cross(R1w[2], v1)                                         # new y: orthogonal to the old z and the new x
    v3 = np.cross(v1, v2)
    R = np.stack([v / np.linalg.norm(v) for v in (v1, v2, v3)])
    K1r = K1.copy(); K1r[0, 1] = 0; K2r = K2.copy(); K2r[0, 1] = 0
    K1r[1, 1] = K2r[1, 1] = (K1[1, 1] + K2[1, 1]) / 2
    R1 = R @ R1w.T; R2 = R @ R2w.T
    t = (R2 @ (poseR @ (-poseC)))[0]
    return R1, R2, K1r, K2r, float(t)


def _project_h(H, xy):
    """ProjectVertex_3x3_2_2 (libs/Common/Util.inl:389-393) for an (n,2) float32 array -> float32."""
    x = xy[:, 0].astype(np.float64); y = xy[:, 1].astype(np.float64)
    z = H[2, 0] * x + H[2, 1] * y + H[2, 2]
    inv = np.where(z == 0, 1e14, 1.0 / np.where(z == 0, 1, z))
    return np.stack([((H[0, 0] * x + H[0, 1] * y + H[0, 2]) * inv).astype(np.float32), ((H[1, 0] * x + H[1, 1] * y + H[1, 2]) * inv).astype(np.float32)], 1)


def set_rectification_roi(points1, points2, size1, size2, K1o, K2o, R1, R2, K1, K2):
    """SetStereoRectificationROI: equal x focal lengths, then centre the rectified images on the area covered by the shared points.
    points: (n,3) float32 (x, y, depth) image projections of the shared sparse points.  -> (K1, K2, (w, h))."""
    K1 = K1.copy(); K2 = K2.copy()
    K1[0, 1] = K2[0, 1] = 0
    K1[0, 0] = K2[0, 0] = (K1[0, 0] + K2[0, 0]) / 2
    H1 = K1 @ R1 @ _inv_k(K1o); H2 = K2 @ R2 @ _inv_k(K2o)
    a = _project_h(H1, points1[:, :2]); b = _project_h(H2, points2[:, :2])
    s1 = a.max(0) - a.min(0); s2 = b.max(0) - b.min(0)                # AABB GetSize
    maxSize = max(size1[0] + size2[0], size1[1] + size2[1]) // 2
    rnd = lambda v: int(np.floor(np.float32(v) + np.float32(0.5)))
    w = min(rnd(max(s1[0], s2[0])), maxSize); h = min(rnd(max(s1[1], s2[1])), maxSize)
    c1 = (a.min(0) + a.max(0)) * np.float32(0.5); c2 = (b.min(0) + b.max(0)) * np.float32(0.5)     # GetCenter
    K1[0, 2] += w // 2 - c1[0]; K1[1, 2] += h // 2 - c1[1]
    K2[0, 2] += w // 2 - c2[0]; K2[1, 2] += h // 2 - c2[1]
    return K1, K2, (w, h)


def warp_perspective_u8(img, H, size):
    """dst(x, y) = bilinear src(H^-1 (x, y)), 0 outside (the approximation of cv::warpPerspective described in the module docstring)."""
    w, h = size
    Hi = np.linalg.inv(H)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    Z = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    X = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / Z; Y = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / Z
    H0, W0 = img.shape[:2]
    x0 = np.floor(X).astype(np.int64); y0 = np.floor(Y).astype(np.int64)
    fx = (X - x0).astype(np.float32)[..., None]; fy = (Y - y0).astype(np.float32)[..., None]
    src = img.astype(np.float32).reshape(H0, W0, -1)

    def tap(yy, xx):
        ok = (xx >= 0) & (yy >= 0) & (xx < W0) & (yy < H0)
        v = src[np.clip(yy, 0, H0 - 1), np.clip(xx, 0, W0 - 1)]
        return np.where(ok[..., None], v, np.float32(0))
    out = (tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx) * (1 - fy) + (tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx) * fy
    out = np.clip(np.floor(out + np.float32(0.5)), 0, 255).astype(np.uint8)
    inside = (X >= 0) & (Y >= 0) & (X <= W0) & (Y <= H0)
    return out.reshape((h, w) + img.shape[2:]), np.where(inside, 255, 0).astype(np.uint8)


def stereo_rectify_images(img1, K1, R1w, C1, img2, K2, R2w, C2, points1, points2):
    """Image::StereoRectifyImages.  img*: (h, w, 3) uint8 BGR; points*: (n,3) projections (x, y, depth) of the sparse points both images see.
    -> dict(rect1, rect2, mask1, mask2, H (3x3, original -> rectified left), Q (4x4), size) or None if the baseline vanishes."""
    R1, R2, K1r, K2r, t = stereo_rectify_fusiello(K1, R1w, C1, K2, R2w, C2)
    if abs(t) < 1e-7:
        return None
    size1 = (img1.shape[1], img1.shape[0]); size2 = (img2.shape[1], img2.shape[0])
    size = size1
    if len(points1):
        K1r, K2r, size = set_rectification_roi(np.asarray(points1, np.float32), np.asarray(points2, np.float32), size1, size2,
                                               np.asarray(K1, np.float64), np.asarray(K2, np.float64), R1, R2, K1r, K2r)
    H1 = K1r @ R1 @ _inv_k(np.asarray(K1, np.float64)); H2 = K2r @ R2 @ _inv_k(np.asarray(K2, np.float64))
    rect1, mask1 = warp_perspective_u8(img1, H1, size); rect2, mask2 = warp_perspective_u8(img2, H2, size)
    Q = np.zeros((4, 4))                                               # Q * [x, y, disparity, 1] = [X, Y, Z, 1] * w in camera-1 coordinates (:326-337)
    Q[0, 0] = Q[1, 1] = 1; Q[0, 3] = -K1r[0, 2]; Q[1, 3] = -K1r[1, 2]; Q[2, 3] = K1r[0, 0]; Q[3, 2] = -1.0 / t; Q[3, 3] = (K1r[0, 2] - K2r[0, 2]) / t
    P = np.eye(4); P[:3, :3] = np.asarray(K1, np.float64) @ R1.T       # ... then into the original image 1 (:339-342)
    return dict(rect1=rect1, rect2=rect2, mask1=mask1, mask2=mask2, H=H1, Q=P @ Q, size=size, t=t, K1=K1r, K2=K2r, R1=R1, R2=R2)


def scale_stereo_rectification(H, Q, scale):
    """Image::ScaleStereoRectification (libs/MVS/Image.cpp:349-364): H and Q of the same pair with both images resized by `scale`."""
    S = np.diag([scale, scale, 1.0])
    H2 = S @ np.asarray(H, np.float64) @ np.linalg.inv(S)
    Sinv = np.diag([1.0 / scale, 1.0 / scale, 1.0 / scale, 1.0])
    S4 = np.diag([scale * scale, scale * scale, scale, scale])
    return H2, S4 @ np.asarray(Q, np.float64) @ Sinv

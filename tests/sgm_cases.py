"""Seeded SGM problems shared by the CPU and GPU tests."""
import numpy as np

from openmvs_amd import sgm


def stereo_pair(w, h, shift, seed=0):
    """Left/right images with right(x + shift) == left(x): the true disparity is `shift` everywhere."""
    r = np.random.RandomState(seed)
    base = r.rand(h, w + 2 * abs(shift) + 8, 3)
    # smooth a little so ZNCC is well behaved
    for _ in range(2):
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, -1, 0) + np.roll(base, -1, 1)) / 5
    base = (base - base.min()) / (base.max() - base.min())
    o = abs(shift) + 4
    left = base[:, o:o + w]
    right = base[:, o - shift:o - shift + w]
    lb = np.clip(np.round(left * 255), 0, 255).astype(np.uint8)
    rb = np.clip(np.round(right * 255), 0, 255).astype(np.uint8)
    gray = lambda b: (np.float32(0.114) * (b[..., 0].astype(np.float32) / np.float32(255)) + np.float32(0.587) * (b[..., 1].astype(np.float32) / np.float32(255))
                      + np.float32(0.299) * (b[..., 2].astype(np.float32) / np.float32(255))).astype(np.float32)
    return lb, gray(lb), gray(rb)


def ranges(w, h, kind, dmin, dmax, seed=1):
    """Per-pixel disparity ranges on the valid grid (h-6, w-6)."""
    vh, vw = h - 6, w - 6
    r = np.random.RandomState(seed)
    if kind == "uniform":
        mn = np.full((vh, vw), dmin, np.int16); mx = np.full((vh, vw), dmax, np.int16)
    elif kind == "ragged":       # per-pixel D in {3..dmax-dmin}, random placement, ~5% invalid pixels
        nd = r.randint(3, dmax - dmin + 1, (vh, vw))
        mn = (dmin + r.randint(0, (dmax - dmin) - nd + 1)).astype(np.int16)
        mx = (mn + nd).astype(np.int16)
        inv = r.rand(vh, vw) < 0.05
        mx[inv] = mn[inv]
    elif kind == "holes":        # what Disparity2RangeMap produces under a mask: long runs of invalid pixels (NO_DISP, NO_DISP) with ragged borders,
        nd = r.randint(3, dmax - dmin + 1, (vh, vw))                 # longer than the path kernel's 64-pixel table chunk in every direction
        mn = (dmin + r.randint(0, (dmax - dmin) - nd + 1)).astype(np.int16)
        mx = (mn + nd).astype(np.int16)
        inv = np.zeros((vh, vw), bool)
        for y in range(vh):
            a = vw // 8 + r.randint(0, 5); inv[y, a:a + 70 + r.randint(0, 9)] = True
        for x in range(vw - vw // 6, vw):
            a = vh // 8 + r.randint(0, 5); inv[a:a + 70 + r.randint(0, 9), x] = True
        mn[inv] = 32767; mx[inv] = 32767
    else:
        raise ValueError(kind)
    return sgm.make_pixels(mn, mx)

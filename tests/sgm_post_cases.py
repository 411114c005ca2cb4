"""Inputs for the tSGM post-processing tests: disparity maps with holes, borders and inconsistencies; masks; cost maps."""
import numpy as np

NO_DISP = 32767


def disparity_pair(w, h, seed=0, true_d=6):
    """A left->right map around +true_d and a right->left map around -true_d with noise, holes and outliers."""
    r = np.random.RandomState(seed)
    l2r = (true_d + r.randint(-2, 3, (h, w))).astype(np.int16)
    r2l = (-true_d + r.randint(-2, 3, (h, w))).astype(np.int16)
    for m in (l2r, r2l):
        m[r.rand(h, w) < 0.15] = NO_DISP
        o = r.rand(h, w) < 0.05
        m[o] = r.randint(-w, w, int(o.sum())).astype(np.int16)
        m[:, :r.randint(0, 9)] = NO_DISP; m[:, w - r.randint(1, 9):] = NO_DISP
        m[r.randint(0, h)] = NO_DISP                      # an empty row
    return l2r, r2l


def cost_map(w, h, seed=0):
    return np.random.RandomState(seed + 100).randint(0, 2600, (h, w)).astype(np.uint16)


def mask_map(w, h, seed=0):
    r = np.random.RandomState(seed + 200)
    m = np.full((h, w), 255, np.uint8)
    m[r.rand(h, w) < 0.1] = 0
    m[:, :3] = 0
    return m


def smooth_disparity(w, h, seed=0):
    """A piecewise-smooth low-resolution disparity map with holes (what a tSGM level hands to Disparity2RangeMap)."""
    r = np.random.RandomState(seed + 300)
    ys, xs = np.mgrid[0:h, 0:w]
    d = (8 * np.sin(xs / 9.0) + 5 * np.cos(ys / 7.0) + (xs > w // 2) * 12).astype(np.int16)
    d[r.rand(h, w) < 0.2] = NO_DISP
    d[h // 3:h // 3 + 12, w // 4:w // 4 + 25] = NO_DISP        # a hole bigger than the 7x7 window
    d[:, :2] = NO_DISP
    return d


def rectification(seed=0):
    """A plausible stereo rectification: H (3x3 homography close to identity) and Q (4x4 disparity-to-depth) and their inverses."""
    r = np.random.RandomState(seed + 400)
    H = np.eye(3) + 0.02 * r.randn(3, 3); H[2, :2] *= 1e-3; H[2, 2] = 1
    f, cx, cy, B = 600.0, 80.0, 60.0, 0.3
    Qcv = np.array([[1, 0, 0, -cx], [0, 1, 0, -cy], [0, 0, 0, f], [0, 0, -1.0 / B, 0]], np.float64)
    # the reference's Q maps (u, v, -d, 1) to (x*z, y*z, z, 1)*w in ORIGINAL image coordinates (Image.h:87-93): reprojection, then the camera matrix
    K4 = np.array([[f, 0, cx, 0], [0, f, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
    Q = K4 @ Qcv
    return H, Q, np.linalg.inv(H), np.linalg.inv(Q + 1e-9 * np.eye(4))

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

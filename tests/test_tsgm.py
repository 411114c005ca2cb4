"""The tSGM coarse-to-fine loop (openmvs_amd/tsgm.py) driven on the CPU oracle: the loop's own logic (level sizes, mask handling, which map
feeds which step) on a rectified synthetic pair with a planted disparity."""
import numpy as np
import pytest

from openmvs_amd import tsgm
from tests import sgm_cases as sc
from tests.tsgm_backends import OracleBackend


def test_scale_and_resamplers():
    assert tsgm.compute_scale(256, 192, 64) == 2 and tsgm.compute_scale(256, 192, 0) == 0 and tsgm.compute_scale(4000, 3000, 320) == 3
    assert tsgm.compute_scale(100, 80, 640) == 1                    # never coarser than... at least one halving (scale = 1/max(2, 2^level))
    img = np.array([[0, 1, 2, 3], [1, 1, 3, 3], [255, 255, 0, 0], [255, 254, 0, 1]], np.uint8)
    assert tsgm.resize_area_u8(img, 2).tolist() == [[1, 3], [255, 0]]          # (0+1+1+1+2)>>2 = 1, (2+3+3+3+2)>>2 = 3, (1019+2)>>2 = 255, (1+2)>>2 = 0
    assert tsgm.resize_area_u8(img, 4).tolist() == [[65]]                      # 1034/16 = 64.6 -> 65
    m = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert tsgm.resize_nearest_u8(m, 2, 2).tolist() == [[0, 2], [4, 6]]
    with pytest.raises(NotImplementedError):
        tsgm.resize_area_u8(np.zeros((5, 4), np.uint8), 2)


def test_loop_recovers_a_planted_disparity_on_the_oracle():
    w, h, d0 = 256, 192, 12
    lb, lg, rg = sc.stereo_pair(w, h, d0, seed=4)
    rb = np.roll(lb, d0, axis=1)                                     # the right colour image of the same shift (only its gray image is matched)
    mask = np.full((h, w), 255, np.uint8)
    be = OracleBackend()
    disp, cost, levels = tsgm.tsgm_match(be, lb, lg, rb, rg, mask, mask, min_resolution=64)
    assert levels == 3 and disp.shape == (h - 6, w - 6) and cost.shape == disp.shape
    ok = disp != tsgm.NO_DISP
    core = ok[10:-10, 30:-30 - d0]
    assert core.mean() > 0.9
    got = disp[10:-10, 30:-30 - d0][core].astype(np.float64) / 4          # subpixelSteps = 4
    assert np.abs(np.median(got) - d0) < 0.26 and (np.abs(got - d0) < 1).mean() > 0.95
    # an initial disparity map of the right size is accepted, a wrong one rejected
    init = np.full((int(np.rint(h / 4 * 0.5)) - 6, int(np.rint(w / 4 * 0.5)) - 6), d0 // 8, np.int16)
    disp2, _, _ = tsgm.tsgm_match(OracleBackend(), lb, lg, rb, rg, mask, mask, min_resolution=64, init_left_disparity=init)
    assert (disp2 != tsgm.NO_DISP).mean() > 0.6
    with pytest.raises(ValueError):
        tsgm.tsgm_match(OracleBackend(), lb, lg, rb, rg, mask, mask, min_resolution=64, init_left_disparity=init[:-1])

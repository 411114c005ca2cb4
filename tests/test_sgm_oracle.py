"""CPU tests of the SGM oracle (known-answer tests; the reference holds no SGM vectors)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import sgm_cases as sc


def test_p2s_table():
    p = po.sgm_generate_p2s()
    assert p[0] == 60 and p[255] == 4 and (np.diff(p.astype(int)) <= 0).all()     # 4*(1+14) .. 4


def test_recurrence_forms_agree_bruteforce():
    r = np.random.RandomState(0)
    for _ in range(20000):
        pmin = r.randint(-20, 20); pmax = pmin + r.randint(0, 24)
        smin = r.randint(-20, 20); smax = smin + r.randint(1, 24)
        Lp = r.randint(0, 320, max(pmax - pmin, 1)); costs = r.randint(0, 256, smax - smin)
        assert po.sgm_step_forms_agree(Lp, pmin, pmax, costs, smin, smax, 3, int(r.randint(4, 61)))


def test_true_shift_wins_and_costs_are_zero_there():
    lb, lg, rg = sc.stereo_pair(96, 64, 5)
    px, n, mx = sc.ranges(96, 64, "uniform", 0, 16)
    d, c, costs, acc = po.sgm_match(lb, lg, rg, px, n, mx, 3, po.sgm_generate_p2s())
    inner = d[:, :96 - 6 - 16]
    assert (inner == 5).mean() > 0.99
    vol = costs.reshape(64 - 6, 96 - 6, 16)
    # identical patches: ncc = 1/sqrt(1 + eps/(n0*n1)) (eps = 1e-3 damps low-texture patches, :877,968) -> the smallest cost
    assert (vol[:, :70].argmin(-1) == 5).mean() > 0.99 and np.median(vol[:, :70, 5]) < 16
    assert (vol[:, -1, 10:] == 255).all()                    # taps leaving the right image -> 255 (:954-957)


def test_constant_cost_volume_gives_8C_in_the_interior():
    # textureless right image => ncc = 0/sqrt(eps) = 0 => cost 255 everywhere; every path then yields L = C
    lb, lg, rg = sc.stereo_pair(64, 48, 0)
    rg = np.full_like(rg, 0.5)
    px, n, mx = sc.ranges(64, 48, "uniform", -4, 4)
    d, c, costs, acc = po.sgm_match(lb, lg, rg, px, n, mx, 3, po.sgm_generate_p2s())
    inside = np.ones((42, 58), bool); inside[:, :4 + 3] = False; inside[:, -(4 + 3):] = False
    vol = acc.reshape(42, 58, 8)
    assert (costs == 255).all()
    assert (vol[1:-1, 8:-8] == 8 * 255).all()                # interior pixels are first on no path
    assert (vol[0, 0] > 8 * 255).all()                       # corner pixel starts several paths: + P2 each
    assert (d == -4).all()                                   # ties -> first disparity (:1281)


def test_invalid_pixels_are_skipped():
    lb, lg, rg = sc.stereo_pair(64, 48, 2)
    px, n, mx = sc.ranges(64, 48, "ragged", -2, 14)
    d, c, costs, acc = po.sgm_match(lb, lg, rg, px, n, mx, 3, po.sgm_generate_p2s())
    inv = (px["maxDisp"] <= px["minDisp"]).reshape(d.shape)
    assert inv.any() and (c[inv] == 0xFFFF).all() and (d[inv] == px["minDisp"].reshape(d.shape)[inv]).all()
    assert (c[~inv] < 0xFFFF).all()


def test_sgm_golden_fixture_pins_the_oracle():
    import os
    from openmvs_amd import sgm
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sgm_golden_80x60.npz"))
    px = np.zeros(g["idx"].size, sgm.PIXEL_DTYPE); px["idx"] = g["idx"]; px["minDisp"] = g["minDisp"]; px["maxDisp"] = g["maxDisp"]
    d, c, costs, acc = po.sgm_match(g["left_bgr"], g["left_gray"], g["right_gray"], px, int(g["num_costs"]), int(g["max_num_disp"]), 3, po.sgm_generate_p2s())
    assert np.array_equal(d, g["disparity"]) and np.array_equal(c, g["cost"]) and np.array_equal(costs, g["costs"]) and np.array_equal(acc, g["accums"])

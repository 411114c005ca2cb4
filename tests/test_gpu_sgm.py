"""-m gpu: integer-exact parity of the HIP SGM kernels (through the C ABI) with the CPU oracle."""
import os

import numpy as np
import pytest

from openmvs_amd import sgm
from oracle import pyoracle as po
from tests import sgm_cases as sc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def matcher():
    m = sgm.SemiGlobalMatcherHIP(0)
    yield m
    m.close()


def _check(matcher, lb, lg, rg, px, n, mx):
    matcher.set_problem(lb, lg, rg, px, n, mx)
    matcher.Match()
    d, c, costs, acc = matcher.results(volumes=True)
    od, oc, ocosts, oacc = po.sgm_match(lb, lg, rg, px, n, mx, matcher.P1, matcher.P2s)
    assert np.array_equal(costs, ocosts), "cost volume: %d of %d differ" % ((costs != ocosts).sum(), costs.size)
    assert np.array_equal(acc, oacc), "8-path sums: %d of %d differ" % ((acc != oacc).sum(), acc.size)
    assert np.array_equal(d, od) and np.array_equal(c, oc)
    return d, c


def test_p2s_match(matcher):
    assert np.array_equal(matcher.P2s, po.sgm_generate_p2s())


@pytest.mark.parametrize("w,h,kind,dmin,dmax", [(96, 64, "uniform", 0, 16), (96, 64, "uniform", -8, 56), (128, 80, "ragged", -4, 60),
                                               (70, 150, "ragged", -3, 30), (160, 96, "uniform", -10, 118), (120, 90, "ragged", 0, 200),
                                               (97, 65, "ragged", -5, 40), (203, 71, "uniform", 0, 70)])   # odd valid widths: the last pixel of a row has no pair partner
def test_match_parity(matcher, w, h, kind, dmin, dmax):
    lb, lg, rg = sc.stereo_pair(w, h, 5, seed=w)
    px, n, mx = sc.ranges(w, h, kind, dmin, dmax, seed=h)
    d, c = _check(matcher, lb, lg, rg, px, n, mx)
    if kind == "uniform" and dmin <= 5 < dmax:
        assert (d[:, 8:w - 6 - (dmax + 8)] == 5).mean() > 0.97


@pytest.mark.parametrize("w,h,dmin,dmax", [(96, 64, -8, 56), (150, 90, 0, 33), (90, 160, -3, 100), (214, 77, 0, 127), (80, 75, 2, 130), (71, 12, 0, 1), (12, 140, -1, 1)])
def test_uniform_range_path_kernel(matcher, w, h, dmin, dmax):
    """One range for every pixel (a plain Match and the first tSGM level): the engine aggregates with sgm_path_uniform_kernel (previous line of L in
    registers, neighbours by DPP wave shifts).  Full waves (64, 128), odd counts (33, 103, 127: pixels start on odd halves of the sum words), one and two
    entries per lane, lines longer than a 64-pixel chunk in every direction, one- and two-disparity ranges; the same integers as the oracle and as the
    general kernel."""
    lb, lg, rg = sc.stereo_pair(w, h, 5, seed=w + h)
    px, n, mx = sc.ranges(w, h, "uniform", dmin, dmax)
    _check(matcher, lb, lg, rg, px, n, mx)


def test_uniform_ranges_with_penalties_above_a_byte(matcher):
    """The atomic-free aggregation records best - min Lp <= P2 in a byte per direction; a penalty table with entries above 255 must take the u16 atomic sums instead
    (same integers as the oracle either way)."""
    w, h = 120, 70
    lb, lg, rg = sc.stereo_pair(w, h, 5, seed=77)
    px, n, mx = sc.ranges(w, h, "uniform", -2, 46)
    saved = matcher.P2s.copy()
    try:
        matcher.P2s = np.minimum(saved.astype(np.int64) * 40 + 200, 3000).astype(np.uint16)
        assert int(matcher.P2s.max()) > 255
        _check(matcher, lb, lg, rg, px, n, mx)
    finally:
        matcher.P2s = saved


def test_uniform_premise_is_checked(matcher):
    """One pixel with a different range, or with an idx that is not pixel * nD, sends the problem to the general path kernel: same results as the oracle."""
    w, h = 110, 80
    lb, lg, rg = sc.stereo_pair(w, h, 5, seed=21)
    mn = np.full((h - 6, w - 6), -4, np.int16); mx_ = np.full((h - 6, w - 6), 40, np.int16)
    mx_[37, 51] = 39
    px, n, mx = sgm.make_pixels(mn, mx_)
    _check(matcher, lb, lg, rg, px, n, mx)
    mn[0, 0] = 32767; mx_[0, 0] = 32767; mx_[37, 51] = 40           # the first pixel invalid: "every pixel like the first" must not pass either
    px, n, mx = sgm.make_pixels(mn, mx_)
    _check(matcher, lb, lg, rg, px, n, mx)


def test_pixel_table_with_entries_out_of_pixel_order(matcher):
    """PixelData::idx is whatever the caller says: a table whose pixels own their entries in reverse pixel order (still a partition of the volume) gives the
    same per-pixel results; the lane-per-pixel cost kernel's contiguous-tile write-out must notice and store per pixel."""
    w, h = 100, 40
    lb, lg, rg = sc.stereo_pair(w, h, 4, seed=5)
    px, n, mx = sc.ranges(w, h, "ragged", -3, 20, seed=2)
    nd = np.maximum(px["maxDisp"].astype(np.int64) - px["minDisp"].astype(np.int64), 0)
    rev = px.copy()
    rev["idx"] = (np.cumsum(nd[::-1])[::-1] - nd).astype(np.uint64)
    d0, c0 = _check(matcher, lb, lg, rg, px, n, mx)
    d1, c1 = _check(matcher, lb, lg, rg, rev, n, mx)
    assert np.array_equal(d0, d1) and np.array_equal(c0, c1)


def test_match_parity_across_long_invalid_runs(matcher):
    """Masked regions (ranges NO_DISP..NO_DISP) wider than the path kernel's 64-pixel table chunk: paths skip them without resetting their state
    (SemiGlobalMatcher.cpp:1071-1072), so a whole staged chunk can be invalid."""
    w, h = 230, 100                                             # lines of more than 3 chunks, so that chunk k+2 holds valid pixels where chunk k has a hole
    lb, lg, rg = sc.stereo_pair(w, h, 5, seed=8)
    px, n, mx = sc.ranges(w, h, "holes", -2, 12, seed=3)
    _check(matcher, lb, lg, rg, px, n, mx)


@pytest.mark.parametrize("w,h,kind,dmin,dmax", [(96, 64, "uniform", 0, 16), (97, 65, "ragged", -5, 40), (128, 80, "ragged", -4, 60), (70, 150, "ragged", -3, 30),
                                               (120, 90, "ragged", 0, 200), (203, 71, "uniform", 0, 70), (230, 100, "holes", -2, 12)])
def test_sub_group_kernels_match_parity(matcher, w, h, kind, dmin, dmax):
    """The 16-lanes-per-pixel mapping of cost volume, path aggregation and WTA (csrc/sgm_kernels_sub.hip, the one the resident tSGM loop uses for
    narrow ranges): the same integer-exact parity as the wide kernels, ranges narrower and wider than a sub-group."""
    lb, lg, rg = sc.stereo_pair(w, h, 5, seed=w)
    px, n, mx = sc.ranges(w, h, kind, dmin, dmax, seed=h)
    matcher.set_sub_group_kernels(True)
    try:
        _check(matcher, lb, lg, rg, px, n, mx)
    finally:
        matcher.set_sub_group_kernels(False)


@pytest.mark.parametrize("sub", [False, True])
def test_range_limits(matcher, sub):
    """The widest range the engine takes (256 disparities, most of them outside the right image: cost 255), one more is an argument error; a valid grid
    of a single row and of a single column; both kernel mappings."""
    matcher.set_sub_group_kernels(sub)
    try:
        lb, lg, rg = sc.stereo_pair(80, 20, 3, seed=2)
        px, n, mx = sc.ranges(80, 20, "uniform", -128, 128)
        assert mx == 256
        _check(matcher, lb, lg, rg, px, n, mx)
        px, n, mx = sc.ranges(80, 20, "uniform", -128, 129)
        with pytest.raises(sgm.SGMError):
            matcher.set_problem(lb, lg, rg, px, n, mx)
        for w, h in ((40, 7), (7, 40)):
            lb, lg, rg = sc.stereo_pair(w, h, 1, seed=5)
            px, n, mx = sc.ranges(w, h, "ragged", -2, 9, seed=6)
            _check(matcher, lb, lg, rg, px, n, mx)
    finally:
        matcher.set_sub_group_kernels(False)


@pytest.mark.parametrize("lanes", [8, 32])
def test_sub_group_widths(matcher, lanes):
    """The other two sub-group widths (8 and 32 lanes per pixel / pair / line; 16 is covered above)."""
    matcher.set_sub_group_kernels(lanes)
    try:
        for w, h, kind, dmin, dmax in ((97, 65, "ragged", -5, 40), (230, 100, "holes", -2, 12), (120, 90, "ragged", 0, 200)):
            lb, lg, rg = sc.stereo_pair(w, h, 5, seed=w)
            px, n, mx = sc.ranges(w, h, kind, dmin, dmax, seed=h)
            _check(matcher, lb, lg, rg, px, n, mx)
    finally:
        matcher.set_sub_group_kernels(False)
    with pytest.raises(sgm.SGMError):
        matcher.set_sub_group_kernels(12)


def test_tiny_and_degenerate(matcher):
    lb, lg, rg = sc.stereo_pair(8, 8, 0)
    px, n, mx = sc.ranges(8, 8, "uniform", -1, 2)
    _check(matcher, lb, lg, rg, px, n, mx)                      # 2x2 valid grid
    lb, lg, rg = sc.stereo_pair(40, 30, 1)
    px, n, mx = sc.ranges(40, 30, "ragged", 0, 8)
    px["maxDisp"][1:] = px["minDisp"][1:]                        # a single valid pixel
    px["idx"][:] = 0
    _check(matcher, lb, lg, rg, px, int(px["maxDisp"][0] - px["minDisp"][0]), int(px["maxDisp"][0] - px["minDisp"][0]))


def test_golden_fixture(matcher):
    g = np.load(os.path.join(GOLD, "sgm_golden_80x60.npz"))
    px = np.zeros(g["idx"].size, sgm.PIXEL_DTYPE); px["idx"] = g["idx"]; px["minDisp"] = g["minDisp"]; px["maxDisp"] = g["maxDisp"]
    matcher.set_problem(g["left_bgr"], g["left_gray"], g["right_gray"], px, int(g["num_costs"]), int(g["max_num_disp"]))
    matcher.Match()
    d, c, costs, acc = matcher.results(volumes=True)
    assert np.array_equal(d, g["disparity"]) and np.array_equal(c, g["cost"]) and np.array_equal(costs, g["costs"]) and np.array_equal(acc, g["accums"])


def test_config4_full_size_matches_golden(matcher):
    """BASELINE config 4 at its own size (2048x1536; D = 64, D = 128, ragged D <= 64) against the digests of the CPU oracle's Match
    (tests/golden/make_fullsize_golden.py; SemiGlobalMatcher.cpp:863-1302): cost volume, 8-path sums, disparities and costs, byte for byte."""
    from tests import golden_check as gc
    g = gc.load("sgm_config4_2048x1536.json")
    c = g["case"]
    lb, lg, rg = sc.stereo_pair(c["width"], c["height"], c["shift"], seed=c["seed"])
    for res in g["results"]:
        px, n, mx = sc.ranges(c["width"], c["height"], res["kind"], res["lo"], res["hi"])
        assert gc.sha(lb) == res["inputs"]["left_bgr"] and gc.sha(lg) == res["inputs"]["left_gray"] and gc.sha(rg) == res["inputs"]["right_gray"] \
            and gc.sha(px) == res["inputs"]["pixels"] and n == res["num_costs"], "the seeded SGM problem was not reproduced on this machine"
        matcher.set_problem(lb, lg, rg, px, n, mx)
        matcher.Match()
        d, cst, costs, acc = matcher.results(volumes=True)
        what = "%s D<=%d" % (res["kind"], res["hi"] - res["lo"])
        assert gc.sha(costs) == res["costs"], what + ": cost volume differs (sum %d vs %d)" % (int(costs.astype(np.uint64).sum()), res["costs_sum"])
        assert gc.sha(acc) == res["accums"], what + ": 8-path sums differ (sum %d vs %d)" % (int(acc.astype(np.uint64).sum()), res["accums_sum"])
        assert gc.sha(d) == res["disparity"], what + ": disparity rows %s differ" % gc.bad_rows(d, res["disparity_rows"])[:8]
        assert gc.sha(cst) == res["cost"], what + ": cost rows %s differ" % gc.bad_rows(cst, res["cost_rows"])[:8]


def test_full_size_properties(matcher):
    """BASELINE config 4 size (2048x1536, D = 64): the oracle takes minutes, so check size-independent
    properties: determinism, the planted disparity is recovered, sums bounded by 8*(255+60)."""
    w, h = 2048, 1536
    lb, lg, rg = sc.stereo_pair(w, h, 21, seed=9)
    px, n, mx = sc.ranges(w, h, "uniform", 0, 64)
    matcher.set_problem(lb, lg, rg, px, n, mx)
    matcher.Match(); d1, c1 = matcher.results()
    matcher.Match(); d2, c2 = matcher.results()
    assert np.array_equal(d1, d2) and np.array_equal(c1, c2)
    assert (d1[:, 8:w - 6 - 72] == 21).mean() > 0.98
    assert c1.max() <= 8 * (255 + 60)

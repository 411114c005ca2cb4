"""-m gpu: the tSGM steps around Match on the device (csrc/sgm_post.hip through the C ABI) against the sequential oracle; exact."""
import numpy as np
import pytest

from openmvs_amd import sgm
from oracle import pyoracle as po
from tests import sgm_cases as sc
from tests import sgm_post_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    m = sgm.SemiGlobalMatcherHIP(0)
    yield m
    m.close()


@pytest.mark.parametrize("w,h,seed", [(131, 77, 1), (640, 301, 2), (9, 5, 3)])
def test_map_steps_match_the_oracle(matcher, w, h, seed):
    l2r, r2l = pc.disparity_pair(w, h, seed)
    for th in (0, 1, 3):
        assert np.array_equal(matcher.ConsistencyCrossCheck(l2r, r2l, th), po.sgm_cross_check(l2r, r2l, th))
    narrow = r2l[:, :max(1, w - 5)].copy()
    assert np.array_equal(matcher.ConsistencyCrossCheck(l2r, narrow, 1), po.sgm_cross_check(l2r, narrow, 1))
    cost = pc.cost_map(w, h, seed)
    assert np.array_equal(matcher.FilterByCost(l2r, cost, 1300), po.sgm_filter_by_cost(l2r, cost, 1300))
    m0 = pc.mask_map(w, h, seed)
    for tv in (1, 3, 50):
        assert np.array_equal(matcher.ExtractMask(l2r, thValid=tv), po.sgm_extract_mask(l2r, thValid=tv))
        assert np.array_equal(matcher.ExtractMask(l2r, m0, tv), po.sgm_extract_mask(l2r, m0, tv))
    for size in ((2 * w + 6, 2 * h + 6), (2 * w + 5, 2 * h + 7), (2 * w + 1, 2 * h)):
        assert np.array_equal(matcher.UpscaleMask(m0, size), po.sgm_upscale_mask(m0, size))
    assert np.array_equal(matcher.FlipDirection(l2r), po.sgm_flip_direction(l2r))


def test_refine_on_the_resident_match(matcher):
    w, h = 80, 60
    lb, lg, rg = sc.stereo_pair(w, h, 5, seed=3)
    px, n, mx = sc.ranges(w, h, "ragged", -3, 20, seed=4)
    od, oc, ocosts, oacc = po.sgm_match(lb, lg, rg, px, n, mx, matcher.P1, matcher.P2s)
    for mode in range(7):
        for steps in (1, 4, 16):
            matcher.set_problem(lb, lg, rg, px, n, mx)
            matcher.Match()
            matcher.RefineDisparityMap(mode, steps)
            d, c = matcher.results()
            assert np.array_equal(d, po.sgm_refine(od, px, oacc, mode, steps)), (mode, steps)
            assert np.array_equal(c, oc)


# Rows f8-f9: first confirmed on a device by the round-1 driver run (GPUTEST_r01: all cases passed); strict since round 2.
@pytest.mark.parametrize("w,h,seed", [(64, 40, 0), (97, 53, 1)])
def test_range_map_matches_the_oracle(matcher, w, h, seed):
    d = pc.smooth_disparity(w, h, seed)
    mask = pc.mask_map(2 * w + 7, 2 * h + 6, seed)
    for a, b in ((11, 33), (5, 7)):
        px0, n0, m0 = po.sgm_disparity2range_map(d, mask, a, b)
        px1, n1, m1 = matcher.Disparity2RangeMap(d, mask, a, b)
        assert n0 == n1 and m0 == m1
        for k in ("idx", "minDisp", "maxDisp"):
            assert np.array_equal(px0[k], px1[k]), k


@pytest.mark.parametrize("w,h,seed", [(64, 40, 0), (97, 53, 1)])
def test_disparity_depth_conversions_match_the_oracle(matcher, w, h, seed):
    H, Q, iH, iQ = pc.rectification(seed)
    depth = (3.0 + 0.5 * np.sin(np.arange(h * w).reshape(h, w) / 50.0)).astype(np.float32)
    depth[np.random.RandomState(seed).rand(h, w) < 0.2] = 0
    cost = pc.cost_map(w - 6, h - 6, seed)
    for steps in (1, 4):
        a = po.sgm_depth2disparity_map(depth, iH, iQ, steps, (w - 6, h - 6))
        assert np.array_equal(matcher.Depth2DisparityMap(depth, iH, iQ, steps, (w - 6, h - 6)), a)
        for cst in (None, cost):
            da, ca = po.sgm_disparity2depth_map(a, cst, H, Q, steps, (w, h))
            db, cb = matcher.Disparity2DepthMap(a, cst, H, Q, steps, (w, h))
            assert np.array_equal(da.view(np.uint32), db.view(np.uint32))
            assert cst is None or np.array_equal(ca.view(np.uint32), cb.view(np.uint32))


@pytest.mark.parametrize("w,h,seed", [(64, 40, 0), (97, 53, 1)])
def test_projection_and_pair_fusion_match_the_oracle(matcher, w, h, seed):
    from tests.test_sgm_post import _pair_maps
    H, Q, iH, iQ = pc.rectification(seed)
    depth = (3.0 + 0.5 * np.sin(np.arange(h * w).reshape(h, w) / 50.0)).astype(np.float32)
    depth[np.random.RandomState(seed).rand(h, w) < 0.15] = 0
    cost = pc.cost_map(w - 6, h - 6, seed)
    for steps in (1, 4):
        disp = po.sgm_depth2disparity_map(depth, np.eye(3), iQ, steps, (w - 6, h - 6))
        for cst in (None, cost):
            ok0, d0, r0, c0 = po.sgm_project_disparity2depth_map(disp, cst, Q, steps, (w, h))
            ok1, d1, r1, c1 = matcher.ProjectDisparity2DepthMap(disp, cst, Q, steps, (w, h))
            assert ok0 == ok1 and np.array_equal(d0.view(np.uint32), d1.view(np.uint32)) and np.array_equal(r0.view(np.uint32), r1.view(np.uint32))
            assert cst is None or np.array_equal(c0.view(np.uint32), c1.view(np.uint32))
    base, deps, rgs, cfs = _pair_maps(w, h, seed)
    for mv in (1, 2, 3):
        a = po.sgm_fuse_pairs(deps, rgs, cfs, mv)
        b = matcher.FusePairs(deps, rgs, cfs, mv)
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_filter_speckles_matches_the_oracle(matcher):
    for w, h, seed in ((64, 40, 0), (131, 77, 1), (400, 300, 2)):
        base = pc.smooth_disparity(w, h, seed)
        r = np.random.RandomState(seed + 9)
        noisy = base.copy(); o = r.rand(h, w) < 0.08; noisy[o] = r.randint(-60, 60, int(o.sum())).astype(np.int16)
        for mx, df in ((100, 5), (10, 1), (0, 0), (5000, 50)):
            assert np.array_equal(matcher.FilterSpeckles(noisy, mx, df), po.sgm_filter_speckles(noisy, mx, df)), (w, h, mx, df)


def test_tsgm_loop_on_the_device_equals_the_loop_on_the_oracle(matcher):
    """openmvs_amd/tsgm.py: the same coarse-to-fine loop, every step on the device vs every step on the oracle."""
    from openmvs_amd import tsgm
    from openmvs_amd.patchmatch import PatchMatchHIP
    from tests.tsgm_backends import DeviceBackend, OracleBackend
    w, h, d0 = 256, 192, 12
    lb, lg, rg = sc.stereo_pair(w, h, d0, seed=4)
    rb = np.roll(lb, d0, axis=1)
    mask = np.full((h, w), 255, np.uint8); mask[:, :5] = 0
    e = PatchMatchHIP(0)
    dev = tsgm.tsgm_match(DeviceBackend(matcher, e), lb, lg, rb, rg, mask, mask, min_resolution=64)
    ref = tsgm.tsgm_match(OracleBackend(), lb, lg, rb, rg, mask, mask, min_resolution=64)
    e.close()
    assert dev[2] == ref[2] == 3 and np.array_equal(dev[0], ref[0]) and np.array_equal(dev[1], ref[1])
    ok = dev[0] != tsgm.NO_DISP
    assert ok.mean() > 0.6 and abs(np.median(dev[0][ok] / 4.0) - d0) < 0.5


@pytest.mark.parametrize("w,h,d0,min_res", [(96, 64, 6, 32), (200, 120, 7, 30), (256, 192, 12, 64)])
def test_resident_tsgm_loop_equals_the_stepwise_loop(matcher, w, h, d0, min_res):
    """sgmhip_tsgm_match (the whole loop in one call, resident in HBM) against openmvs_amd/tsgm.py on the oracle backend: with masks, with and
    without an initial disparity map, default and non-default speckle / sub-pixel options."""
    from openmvs_amd import tsgm
    from tests.tsgm_backends import OracleBackend
    lb, lg, rg = sc.stereo_pair(w, h, d0, seed=4)
    rb = np.roll(lb, d0, axis=1)
    mask = np.full((h, w), 255, np.uint8); mask[:, :5] = 0; mask[h // 3:h // 3 + 9, w // 2:w // 2 + 30] = 0
    k = tsgm.compute_scale(w, h, min_res); lw, lh = w >> k, h >> k
    init = np.full((int(np.rint(lh * 0.5)) - 6, int(np.rint(lw * 0.5)) - 6), d0 >> (k + 1), np.int16); init[0, :3] = tsgm.NO_DISP
    for kw in (dict(), dict(init_left_disparity=init, n_speckle_size=20, subpixel_mode=3, subpixel_steps=8)):
        dev = matcher.tsgm_match(lb, rb, lg, rg, mask, mask, min_resolution=min_res, **kw)
        ref = tsgm.tsgm_match(OracleBackend(), lb, lg, rb, rg, mask, mask, min_resolution=min_res, **kw)
        assert dev[2] == ref[2] == k + 1
        assert np.array_equal(dev[0], ref[0]) and np.array_equal(dev[1], ref[1])
        assert (dev[0] != tsgm.NO_DISP).mean() > 0.5
    with pytest.raises(sgm.SGMError):
        matcher.tsgm_match(lb[:, :-1], rb[:, :-1], lg[:, :-1], rg[:, :-1], mask[:, :-1], mask[:, :-1], min_resolution=min_res)      # not a multiple of 2^levels


@pytest.mark.parametrize("w,h,seed", [(64, 40, 0), (97, 53, 1)])
def test_resident_fuse_equals_the_stepwise_fuse(matcher, w, h, seed):
    """sgmhip_fuse_disparities (projection of every pair + per-pixel fusion in one resident call) against the same steps on the oracle backend;
    pairs of different valid-grid sizes and sub-pixel steps, one pair that produces no depth."""
    from openmvs_amd import sgm_pipeline
    from tests.tsgm_backends import OracleBackend
    H, Q, iH, iQ = pc.rectification(seed)
    r = np.random.RandomState(seed + 3)
    pairs = []
    for k, steps in enumerate((1, 4, 4, 2)):
        depth = (3.0 + 0.4 * np.sin(np.arange(h * w).reshape(h, w) / (40.0 + 7 * k)) + 0.02 * k).astype(np.float32)
        depth[r.rand(h, w) < 0.15] = 0
        vw, vh = (w - 6, h - 6) if k != 2 else (w - 10, h - 9)                       # a pair rectified to a smaller grid
        disp = po.sgm_depth2disparity_map(depth, np.eye(3), iQ, steps, (vw, vh))
        if k == 3:
            disp[:] = tsgm_no_disp()                                                  # nothing to project: the pair must be dropped
        pairs.append(dict(disparity=disp, cost=pc.cost_map(vw, vh, seed + k), Q=Q, subpixel_steps=steps, image_size=(w, h)))
    for mv in (1, 2, 3):
        want_d, want_c = sgm_pipeline.fuse_pairs(OracleBackend(), pairs, mv)
        got_d, got_c, used = matcher.fuse_disparities(pairs, (w, h), mv)
        assert used == 3
        assert np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32)) and np.array_equal(got_c.view(np.uint32), want_c.view(np.uint32))
        assert mv > 1 or (got_d > 0).mean() > 0.3
    d0, c0, used = matcher.fuse_disparities(pairs[3:], (w, h), 1)
    assert used == 0 and not d0.any() and not c0.any()


def tsgm_no_disp():
    from openmvs_amd import tsgm
    return tsgm.NO_DISP

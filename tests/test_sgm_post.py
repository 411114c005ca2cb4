"""CPU tests of the tSGM steps around Match: known answers of the sequential oracle, and the parallel forms of the device kernels
(openmvs_amd/csrc/sgm_post.h, run through a host emulation with scrambled thread orders) against it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import sgm_cases as sc
from tests import sgm_post_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))
NO = pc.NO_DISP


@pytest.fixture(scope="module")
def emul():
    src = os.path.join(HERE, "cpp", "sgm_post_emul.cpp")
    out = os.path.join(HERE, "cpp", "build", "libsgm_post_emul.so")
    hdr = os.path.join(HERE, "..", "openmvs_amd", "csrc", "sgm_post.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(out):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", out, src])
    return C.CDLL(out)


def test_cross_check_known_answers():
    l2r = np.full((1, 8), NO, np.int16); r2l = np.full((1, 8), NO, np.int16)
    l2r[0, 2] = 3; r2l[0, 5] = -3          # consistent
    l2r[0, 3] = 3; r2l[0, 6] = -5          # |3-5| = 2 > 1
    l2r[0, 4] = 9                          # lands outside
    l2r[0, 0] = 1                          # partner invalid
    l2r[0, 1] = 3; r2l[0, 4] = -2          # |3-2| = 1 <= 1
    out = po.sgm_cross_check(l2r, r2l)
    assert list(out[0]) == [NO, 3, 3, NO, NO, NO, NO, NO]
    assert list(po.sgm_cross_check(l2r, r2l, thCross=2)[0][:4]) == [NO, 3, 3, 3]


def test_flip_direction_known_answers():
    l2r = np.full((1, 10), NO, np.int16)
    l2r[0, 1] = 2; l2r[0, 2] = 2; l2r[0, 8] = 3; l2r[0, 0] = -4
    out = po.sgm_flip_direction(l2r)[0]
    # pixel 1 (d=2) writes -2 to columns 2..4, pixel 2 (d=2) then overwrites 3..5; pixel 8 (d=3) clips to column 9 (10 is outside)
    exp = np.full(10, NO, np.int16); exp[2:5] = -2; exp[3:6] = -2
    assert np.array_equal(out, exp)
    l2r = np.full((1, 6), NO, np.int16); l2r[0, 4] = -4; l2r[0, 2] = -1
    out = po.sgm_flip_direction(l2r)[0]     # pixel 2 -> columns 0..2 = 1 ; pixel 4 -> columns max(-1,0)..1 = 4 (later column wins on 0,1)
    assert list(out) == [4, 4, 1, NO, NO, NO]


def test_extract_and_upscale_mask_known_answers():
    d = np.array([[NO, 5, NO, 6, 7, 8, NO, 9, 1, NO]], np.int16)
    m = po.sgm_extract_mask(d, thValid=2)[0]
    # from the left: columns 0..3 go INVALID (2nd valid at column 3); from the right: 9, 8, 7 (2nd valid at column 7)
    assert list(m) == [0, 0, 0, 0, 255, 255, 255, 0, 0, 0]
    m0 = np.full((1, 10), 255, np.uint8); m0[0, 1] = 0      # already-invalid pixels are skipped and do not count
    m = po.sgm_extract_mask(d, mask=m0, thValid=2)[0]
    assert list(m) == [0, 0, 0, 0, 0, 255, 255, 0, 0, 0]
    up = po.sgm_upscale_mask(np.array([[255, 0], [0, 255]], np.uint8), (8, 9))
    exp = np.zeros((9, 8), np.uint8); exp[3:5, 3:5] = 255; exp[5:7, 5:7] = 255
    assert np.array_equal(up, exp)
    assert np.array_equal(po.sgm_upscale_mask(np.full((3, 3), 255, np.uint8), (8, 8))[3:, 3:], np.full((5, 5), 255, np.uint8))   # clipped at the border


def test_refine_known_answers():
    px = np.zeros(4, [("idx", np.uint64), ("minDisp", np.int16), ("maxDisp", np.int16), ("pad", np.int32)])
    px["idx"] = [0, 4, 8, 12]; px["minDisp"] = [0, 0, 0, 5]; px["maxDisp"] = [4, 4, 4, 6]
    acc = np.array([10, 20, 30, 40,   40, 20, 40, 50,   40, 30, 20, 10,   7, 0, 0, 0], np.uint16)
    d = np.array([0, 1, 3, 5], np.int16)
    out = po.sgm_refine(d, px, acc, mode=1, steps=4)
    # pixel 0 at its range minimum: +0.5*10/20 = 0.25 -> 1; pixel 1 symmetric parabola -> ld == rd -> x = 1 -> linear 0.5 -> offset 0 -> 4;
    # pixel 2 at the last disparity: 3 - 0.5*10/20 = 2.75 -> 11; pixel 3 has a single disparity: returned unscaled (reference quirk)
    assert list(out) == [1, 4, 11, 5]
    assert list(po.sgm_refine(d, px, acc, mode=0, steps=4)) == [0, 4, 12, 20]
    assert list(po.sgm_refine(d, px, acc, mode=6, steps=1)) == [0, 1, 3, 5]
    none = po.sgm_refine(np.array([NO, NO, NO, NO], np.int16), px, acc)
    assert np.all(none == NO)


@pytest.mark.parametrize("w,h,seed", [(64, 40, 0), (131, 77, 1), (9, 5, 2)])
def test_parallel_forms_match_the_sequential_oracle(emul, w, h, seed):
    l2r, r2l = pc.disparity_pair(w, h, seed)
    E = dict(impl=emul, prefix="emu_sgm_")
    for th in (0, 1, 3):
        assert np.array_equal(po.sgm_cross_check(l2r, r2l, th, **E), po.sgm_cross_check(l2r, r2l, th))
    assert np.array_equal(po.sgm_cross_check(l2r, r2l[:, :w - 5].copy(), 1, **E), po.sgm_cross_check(l2r, r2l[:, :w - 5].copy(), 1))
    cost = pc.cost_map(w, h, seed)
    assert np.array_equal(po.sgm_filter_by_cost(l2r, cost, 1300, **E), po.sgm_filter_by_cost(l2r, cost, 1300))
    for tv in (1, 3, 50):
        assert np.array_equal(po.sgm_extract_mask(l2r, thValid=tv, **E), po.sgm_extract_mask(l2r, thValid=tv))
        m0 = pc.mask_map(w, h, seed)
        assert np.array_equal(po.sgm_extract_mask(l2r, m0, tv, **E), po.sgm_extract_mask(l2r, m0, tv))
    m = pc.mask_map(w, h, seed)
    for size in ((2 * w + 6, 2 * h + 6), (2 * w + 5, 2 * h + 7), (2 * w + 1, 2 * h)):
        assert np.array_equal(po.sgm_upscale_mask(m, size, **E), po.sgm_upscale_mask(m, size))
    assert np.array_equal(po.sgm_flip_direction(l2r, **E), po.sgm_flip_direction(l2r))
    flipped = po.sgm_flip_direction(po.sgm_cross_check(l2r, r2l))
    if w >= 64:
        assert (flipped != NO).sum() > 0 and np.all(flipped[flipped != NO] < 0)


def test_refine_parallel_form_on_a_real_volume(emul):
    """Sums and disparities of an actual Match (the oracle's) at a small size; all seven fits."""
    w, h = 80, 60
    lb, lg, rg = sc.stereo_pair(w, h, 5, seed=3)
    px, n, mx = sc.ranges(w, h, "ragged", -3, 20, seed=4)
    d, c, costs, acc = po.sgm_match(lb, lg, rg, px, n, mx, 3, po.sgm_generate_p2s())
    for mode in range(7):
        for steps in (1, 4, 16):
            a = po.sgm_refine(d, px, acc, mode, steps)
            b = po.sgm_refine(d, px, acc, mode, steps, impl=emul, prefix="emu_sgm_")
            assert np.array_equal(a, b), (mode, steps)
    ref4 = po.sgm_refine(d, px, acc, 6, 4)
    ok = (d != NO) & ((px["maxDisp"] - px["minDisp"]).reshape(d.shape) >= 2)
    assert np.all(np.abs(ref4[ok].astype(np.int32) - 4 * d[ok].astype(np.int32)) <= 2)      # a sub-pixel offset is at most half a disparity


def test_range_map_known_answers():
    """A constant map: median = min = max = d -> numDisp 0 < minNumDisp -> [2d - n/2, 2d + (n+1)/2); masked pixels get an empty range."""
    d = np.full((6, 8), 7, np.int16)
    mask = np.full((2 * 6 + 6, 2 * 8 + 6), 255, np.uint8)
    mask[3 + 2 * 2, 3 + 2 * 5] = 0                       # low-resolution pixel (2,5) is masked out
    px, n, mx = po.sgm_disparity2range_map(d, mask, 5, 7)
    t = px.reshape(mask.shape)
    assert mx == 5 and t["minDisp"][0, 0] == 12 and t["maxDisp"][0, 0] == 17
    assert t["minDisp"][3 + 4, 3 + 10] == NO and t["maxDisp"][3 + 4 + 1, 3 + 10 + 1] == NO      # its 2x2 block
    nd = (t["maxDisp"].astype(np.int64) - t["minDisp"]).ravel()
    assert n == nd.sum() and np.array_equal(px["idx"], np.concatenate([[0], np.cumsum(nd)[:-1]]).astype(np.uint64))
    # fewer than 3 valid disparities around a pixel: symmetric default range min(2w/3, minNumDispInvalid)
    d2 = np.full((6, 8), NO, np.int16); d2[0, 0] = 1
    t2 = po.sgm_disparity2range_map(d2, mask, 5, 7)[0].reshape(mask.shape)
    assert t2["minDisp"][10, 10] == -5 and t2["maxDisp"][10, 10] == 5          # 8*2/3 = 5 < 7


@pytest.mark.parametrize("w,h,seed", [(64, 40, 0), (97, 53, 1)])
def test_range_map_and_conversions_parallel_forms(emul, w, h, seed):
    E = dict(impl=emul, prefix="emu_sgm_")
    d = pc.smooth_disparity(w, h, seed)
    mask = pc.mask_map(2 * w + 7, 2 * h + 6, seed)
    for a, b in ((11, 33), (5, 7)):
        px0, n0, m0 = po.sgm_disparity2range_map(d, mask, a, b)
        px1, n1, m1 = po.sgm_disparity2range_map(d, mask, a, b, **E)
        assert n0 == n1 and m0 == m1 and np.array_equal(px0["idx"], px1["idx"]) and np.array_equal(px0["minDisp"], px1["minDisp"]) and np.array_equal(px0["maxDisp"], px1["maxDisp"])
        assert 0 < m0 <= 66 and n0 > 0
    H, Q, iH, iQ = pc.rectification(seed)
    depth = (3.0 + 0.5 * np.sin(np.arange(h * w).reshape(h, w) / 50.0)).astype(np.float32)
    depth[np.random.RandomState(seed).rand(h, w) < 0.2] = 0
    for steps in (1, 4):
        a = po.sgm_depth2disparity_map(depth, iH, iQ, steps, (w - 6, h - 6))
        b = po.sgm_depth2disparity_map(depth, iH, iQ, steps, (w - 6, h - 6), **E)
        assert np.array_equal(a, b) and (a != NO).mean() > 0.5
        cost = pc.cost_map(w - 6, h - 6, seed)
        for cst in (None, cost):
            da, ca = po.sgm_disparity2depth_map(a, cst, H, Q, steps, (w, h))
            db, cb = po.sgm_disparity2depth_map(a, cst, H, Q, steps, (w, h), **E)
            assert np.array_equal(da.view(np.uint32), db.view(np.uint32)) and (cst is None or np.array_equal(ca.view(np.uint32), cb.view(np.uint32)))
        # depth -> disparity -> depth comes back (up to the disparity quantisation) where both are defined
        back = po.sgm_disparity2depth_map(a, None, H, Q, steps, (w, h))[0]
        ok = (back > 0) & (depth > 0)
        assert ok.mean() > 0.3 and np.median(np.abs(back[ok] - depth[ok]) / depth[ok]) < 0.06


def _pair_maps(w, h, seed, n_pairs=4):
    """What SemiGlobalMatcher::Fuse sees: the same surface projected through n pairs with different noise, holes and trust ranges."""
    r = np.random.RandomState(seed + 500)
    base = (3.0 + 0.5 * np.sin(np.arange(h * w).reshape(h, w) / 40.0)).astype(np.float32)
    deps, rgs, cfs = [], [], []
    for p in range(n_pairs):
        d = (base * (1 + 0.01 * r.randn(h, w))).astype(np.float32)
        o = r.rand(h, w) < 0.1; d[o] *= r.choice([0.6, 1.5], int(o.sum())).astype(np.float32)
        d[r.rand(h, w) < 0.25] = 0
        half = (0.02 + 0.03 * r.rand(h, w)).astype(np.float32) * d
        deps.append(d); rgs.append(np.stack([d - half, d + half], -1).astype(np.float32)); cfs.append(r.rand(h, w).astype(np.float32))
    return base, deps, rgs, cfs


@pytest.mark.parametrize("w,h,seed", [(64, 40, 0), (97, 53, 1)])
def test_projection_and_pair_fusion_parallel_forms(emul, w, h, seed):
    E = dict(impl=emul, prefix="emu_sgm_")
    H, Q, iH, iQ = pc.rectification(seed)
    depth = (3.0 + 0.5 * np.sin(np.arange(h * w).reshape(h, w) / 50.0)).astype(np.float32)
    depth[np.random.RandomState(seed).rand(h, w) < 0.15] = 0
    cost = pc.cost_map(w - 6, h - 6, seed)
    for steps in (1, 4):
        disp = po.sgm_depth2disparity_map(depth, np.linalg.inv(np.eye(3)), iQ, steps, (w - 6, h - 6))     # identity homography: the rectified frame is the image
        for cst in (None, cost):
            ok0, d0, r0, c0 = po.sgm_project_disparity2depth_map(disp, cst, Q, steps, (w, h))
            ok1, d1, r1, c1 = po.sgm_project_disparity2depth_map(disp, cst, Q, steps, (w, h), **E)
            assert ok0 and ok1 and np.array_equal(d0.view(np.uint32), d1.view(np.uint32)) and np.array_equal(r0.view(np.uint32), r1.view(np.uint32))
            assert cst is None or np.array_equal(c0.view(np.uint32), c1.view(np.uint32))
        m = (d0 > 0) & (depth > 0)
        assert m.mean() > 0.3 and np.median(np.abs(d0[m] - depth[m]) / depth[m]) < 0.06
        assert np.all(r0[d0 > 0].min(1) > 0) and np.all(r0[d0 > 0][:, 0] > r0[d0 > 0][:, 1])      # depth at disparity-1 is farther than at disparity+1
    assert not po.sgm_project_disparity2depth_map(np.full((h - 6, w - 6), NO, np.int16), None, Q, 4, (w, h))[0]
    base, deps, rgs, cfs = _pair_maps(w, h, seed)
    for mv in (1, 2, 3):
        a = po.sgm_fuse_pairs(deps, rgs, cfs, mv)
        b = po.sgm_fuse_pairs(deps, rgs, cfs, mv, **E)
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    f2 = po.sgm_fuse_pairs(deps, rgs, cfs, 2)[0]
    ok = f2 > 0
    assert 0.3 < ok.mean() < 1 and np.median(np.abs(f2[ok] - base[ok]) / base[ok]) < 0.01


def test_pair_fusion_known_answer():
    one = lambda v: np.full((1, 1), v, np.float32)
    rng = lambda a, b: np.array([[[a, b]]], np.float32)
    # pair 0: 2.0 in [1.9,2.1); pair 1: 2.05 in [2.0,2.2) joins pair 0's cluster (range shrinks to [2.0,2.1)); pair 2: 5.0 alone
    d, c = po.sgm_fuse_pairs([one(2.0), one(2.05), one(5.0)], [rng(1.9, 2.1), rng(2.0, 2.2), rng(4.9, 5.1)], [one(0.2), one(0.4), one(0.9)], 2)
    assert np.isclose(d[0, 0], (np.float32(2.0) + np.float32(2.05)) / 2) and np.isclose(c[0, 0], 0.3)
    assert po.sgm_fuse_pairs([one(2.0), one(2.05), one(5.0)], [rng(1.9, 2.1), rng(2.0, 2.2), rng(4.9, 5.1)], [one(0.2), one(0.4), one(0.9)], 3)[0][0, 0] == 0
    # the shrunk range [2.0, 2.1) no longer admits 1.95: it founds its own cluster; both clusters have... the first of the largest wins
    d, _ = po.sgm_fuse_pairs([one(2.0), one(2.05), one(1.95)], [rng(1.9, 2.1), rng(2.0, 2.2), rng(1.9, 2.0)], [one(0), one(0), one(0)], 1)
    assert np.isclose(d[0, 0], 2.025)


def test_filter_speckles(emul):
    E = dict(impl=emul, prefix="emu_sgm_")
    d = np.full((6, 10), NO, np.int16)
    d[0, 0:3] = [10, 12, 14]            # a chain: consecutive differences 2 <= maxDiff although the ends differ by 4
    d[2, 0:2] = [10, 20]                # two singletons (difference 10)
    d[4:6, 4:9] = 7                     # a 10-pixel block
    out = po.sgm_filter_speckles(d, maxSpeckleSize=3, maxDiff=2)
    assert np.all(out[0, 0:3] == NO) and np.all(out[2, 0:2] == NO) and np.all(out[4:6, 4:9] == 7)      # size 3 <= 3 erased, 10 kept
    assert np.all(po.sgm_filter_speckles(d, maxSpeckleSize=2, maxDiff=2)[0, 0:3] == [10, 12, 14])       # size 3 > 2 kept
    for w, h, seed in ((64, 40, 0), (131, 77, 1), (200, 150, 2)):
        base = pc.smooth_disparity(w, h, seed)
        r = np.random.RandomState(seed + 9)
        noisy = base.copy(); o = r.rand(h, w) < 0.08; noisy[o] = r.randint(-60, 60, int(o.sum())).astype(np.int16)
        for mx, df in ((100, 5), (10, 1), (0, 0), (5000, 50)):
            a = po.sgm_filter_speckles(noisy, mx, df)
            b = po.sgm_filter_speckles(noisy, mx, df, **E)
            assert np.array_equal(a, b), (w, h, mx, df)
        a = po.sgm_filter_speckles(noisy, 100, 5)
        assert ((noisy != NO) & (a == NO)).sum() > 0 and np.all(a[a != NO] == noisy[a != NO])

"""CPU known-answer tests of the FilterDepthMap oracle (SceneDensify.cpp:1050-1299)."""
import numpy as np
import pytest

from oracle import pyoracle as po


@pytest.fixture(scope="module")
def gt_scene():
    from openmvs_amd import synth
    sc = synth.make_scene(5, 96, 64, n_src=4)
    depth = sc.gt_depth.copy()
    depth[:, :4] = 0; depth[:, -4:] = 0; depth[:, :, :4] = 0; depth[:, :, -4:] = 0     # estimated maps have an empty 4-px border
    conf = np.where(depth > 0, np.float32(0.8), np.float32(0)).astype(np.float32)
    return sc, depth, conf


@pytest.mark.parametrize("adjust", [True, False])
def test_consistent_depth_maps_survive(gt_scene, adjust):
    sc, depth, conf = gt_scene
    rc, nd, nc = po.filter_depth_map(depth, conf, sc.K, sc.R, sc.C, 0, list(sc.neighbors[0]), sc.dmin[0], sc.dmax[0], bAdjust=adjust)
    assert rc == 0
    m = depth[0] > 0
    kept = (nd > 0) & m
    assert kept.sum() > 0.6 * m.sum() and (nd[~m] == 0).all()
    assert np.abs(nd[kept] - depth[0][kept]).max() / depth[0][kept].mean() < 6e-3   # averaged with splatted neighbour depths (thDepthDiff 1.2 %)
    if adjust:
        assert (nc[kept] >= np.float32(0.8)).all()          # own confidence + agreeing neighbours - violations > own
    else:
        assert (nc[kept] == np.float32(0.8)).all() and np.array_equal(nd[kept], depth[0][kept])


def test_outliers_are_removed_and_too_few_neighbours_is_refused(gt_scene):
    sc, depth, conf = gt_scene
    d = depth.copy()
    d[0, 20:30, 30:50] *= 1.5                                # a block of wrong depths in the reference view
    rc, nd, nc = po.filter_depth_map(d, conf, sc.K, sc.R, sc.C, 0, list(sc.neighbors[0]), sc.dmin[0], sc.dmax[0] * 2)
    assert rc == 0 and (nd[20:30, 30:50] == 0).mean() > 0.9
    rc, _, _ = po.filter_depth_map(d, conf, sc.K, sc.R, sc.C, 0, [1], sc.dmin[0], sc.dmax[0])   # N = 1 < nMinViewsFilter = 2 (:1060)
    assert rc == 1


def test_gap_interpolation_fills_small_similar_gaps_only():
    h, w = 12, 30
    depth = np.full((h, w), 2.0, np.float32); conf = np.full((h, w), 0.5, np.float32)
    normal = np.zeros((h, w, 3), np.float32); normal[..., 2] = -1
    depth[3, 5:9] = 0            # 4-px row gap between equal depths -> filled
    depth[5, 10:18] = 0          # 8-px gap > nIpolGapSize -> untouched
    depth[7, 0:3] = 0            # touches the row start -> row pass leaves it; the column pass fills it (1-px column gaps)
    depth[9, 20:23] = 0; depth[9, 23:] = 2.2   # ends differ by 10 % > 2.5 % -> row pass leaves it; column pass fills it
    conf[3, 4] = 0.3
    d, n, c = po.gap_interpolation(depth, normal, conf)
    assert np.allclose(d[3, 5:9], 2.0) and np.allclose(c[3, 5:9], 0.3) and np.allclose(n[3, 5:9], [0, 0, -1], atol=1e-6)
    assert (d[5, 10:18] == 2.0).all()      # the column pass closes 1-row gaps whatever their width
    assert (d[7, 0:3] == 2.0).all() and (d[9, 20:23] == 2.0).all()
    depth2 = depth.copy(); depth2[4:7, 10:18] = 0   # now 3 rows missing: columns fill it (3 <= 7)
    d2, _, _ = po.gap_interpolation(depth2, normal, conf)
    assert (d2[4:7, 10:18] == 2.0).all()
    depth3 = np.full((20, 30), 2.0, np.float32); depth3[4:14, 10:20] = 0   # 10x10 hole: too large both ways
    d3, _, _ = po.gap_interpolation(depth3, np.broadcast_to(normal[0, 0], (20, 30, 3)).copy(), np.full((20, 30), 0.5, np.float32))
    assert (d3[4:14, 10:20] == 0).all()
    ramp = np.full((4, 12), 2.0, np.float32); ramp[1, 3:6] = 0; ramp[1, 6:] = 2.04     # 2 % step: interpolated linearly
    dr, _, _ = po.gap_interpolation(ramp, np.broadcast_to(normal[0, 0], (4, 12, 3)).copy(), np.full((4, 12), 0.5, np.float32))
    assert np.allclose(dr[1, 3:6], [2.01, 2.02, 2.03], atol=1e-6)

"""Pins the oracle's multi-level driver (oracle/pm_oracle.cpp: EstimateDepthMap, ScaleDepthData) to the REFERENCE'S OWN TEXT: DepthMapsData::ScaleDepthData and
DepthMapsData::EstimateDepthMap (libs/MVS/SceneDensify.cpp:578-601, :616-805), cut verbatim at build time and compiled with the verbatim estimator and pass bodies
(oracle/ref/ref_driver_harness.cpp -> oracle/_ref/libref_driver.so).  What this checks bit for bit: the level loop, the hand-off of the low-resolution estimate
(INTER_LINEAR depth / INTER_NEAREST normal; INTER_NEAREST for both with ignore masks, :661), the prior, release of the initial estimate (:665-669), the per-level
pixel list, iteration numbering of the geometric rounds (:626-627), the x 1.333 threshold of EndDepthMapTmp when geometric rounds follow (:774-776), ScaleK.
cv::resize itself is OpenCV (un-vendored): the reference's calls go to the oracle's restated resamplers on both sides, so the resamplers are NOT what
is pinned here -- which one is called, with which size and on which map, is."""
import numpy as np
import pytest

from openmvs_amd import synth
from oracle import pyoracle as po
from oracle import pyref as pr

pytestmark = pytest.mark.skipif(not pr.driver_available(), reason="oracle/_ref/libref_driver.so not built (needs /root/reference)")


def _eq(a, b, what):
    for x, y, nm in zip(a, b, ("depth", "normal", "conf")):
        same = (x == y) | (np.isnan(x) & np.isnan(y))
        assert same.all(), "%s: %s differs in %d of %d values" % (what, nm, int((~same).sum()), x.size)


@pytest.fixture(scope="module")
def scene():
    return synth.make_scene(5, 160, 120, n_src=4)


def _both(sc, v, opt, n_src=None, **kw):
    ids = [v] + list(sc.neighbors[v])[:n_src]
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps=kw.pop("src", None))
    mask = kw.pop("mask", None); mask_mode = kw.pop("mask_mode", False)
    a = pr.ref_estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), opt, mask=mask, mask_mode=mask_mode, **kw)
    if mask is not None or mask_mode:
        b = po.estimate_depth_map_masked(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), opt, mask, mask_mode=True, **kw)
    else:
        b = po.estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), opt, **kw)
    return a, b


@pytest.mark.parametrize("levels", [0, 1, 2])
@pytest.mark.parametrize("geo_follow", [0, 2])
def test_photometric_pass_is_the_reference_driver(scene, levels, geo_follow):
    # geo_follow: OPTDENSE::nEstimationGeometricIters -- non-zero raises the end-of-pass threshold by 1.333 (SceneDensify.cpp:774-776)
    opt = po.default_opt(seed=3 + levels, viewID=1, rngMode=2, nThreads=1, nSubResolutionLevels=levels, nEstimationGeometricIters=geo_follow)
    a, b = _both(scene, 1, opt)
    _eq(a, b, "levels %d, geometric rounds to follow %d" % (levels, geo_follow))
    assert (a[0] > 0).mean() > 0.6


def test_geometric_round_is_the_reference_driver(scene):
    sc = scene
    opt = po.default_opt(seed=5, viewID=2, rngMode=2, nThreads=1, nEstimationGeometricIters=2)
    photo = {}
    for v in range(sc.n_views):
        ids = [v] + list(sc.neighbors[v])
        views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
        photo[v] = po.estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), po.default_opt(seed=5, viewID=v, rngMode=2, nThreads=1, nEstimationGeometricIters=2))
    for g in (0, 1):
        a, b = _both(sc, 2, opt, geo_iter=g, depth=photo[2][0], normal=photo[2][1], src={u: photo[u][0] for u in range(sc.n_views)})
        _eq(a, b, "geometric round %d" % g)
    assert (a[0] > 0).mean() > 0.5


def test_initial_estimate_goes_through_the_nearest_downscale(scene):
    sc = scene; v = 0
    r = np.random.RandomState(4)
    d0 = (sc.gt_depth[v] * (1 + 0.01 * r.randn(*sc.gt_depth[v].shape))).astype(np.float32)
    n0 = np.zeros(d0.shape + (3,), np.float32); n0[..., 2] = -1
    opt = po.default_opt(seed=8, viewID=v, rngMode=2, nThreads=1)
    a, b = _both(sc, v, opt, depth=d0, normal=n0)
    _eq(a, b, "initial estimate")


def test_ignore_mask_and_nearest_hand_off(scene):
    sc = scene; v = 3
    h, w = sc.gray[0].shape
    mask = np.ones((h, w), np.uint8); mask[30:70, 40:110] = 0; mask[:, :7] = 0
    opt = po.default_opt(seed=9, viewID=v, rngMode=2, nThreads=1)
    a, b = _both(sc, v, opt, mask=mask)
    _eq(a, b, "ignore mask")
    assert (a[0][30:70, 40:110] == 0).all()
    a, b = _both(sc, v, opt, mask=None, mask_mode=True)      # label set but no mask file for this image: only the INTER_NEAREST hand-off (:661)
    _eq(a, b, "mask mode without a mask")


def test_sizes_that_do_not_divide(scene):
    sc = synth.make_scene(4, 163, 121, n_src=3)          # level sizes cvRound -> 82x60, 41x30
    opt = po.default_opt(seed=4, viewID=2, rngMode=2, nThreads=1)
    a, b = _both(sc, 2, opt)
    _eq(a, b, "163x121")


@pytest.mark.parametrize("n_src", [1, 2])
def test_few_sources(scene, n_src):
    opt = po.default_opt(seed=6, viewID=4, rngMode=2, nThreads=1)
    a, b = _both(scene, 4, opt, n_src=n_src)
    _eq(a, b, "%d sources" % n_src)


def test_scale_depth_data_of_a_view(scene):
    """ScaleDepthData (SceneDensify.cpp:578-601): image INTER_AREA, K through Camera::GetScaledK(image size, new size), the source's depth map INTER_AREA to the image's
    new size with cameraDepthMap's K scaled the same way -- against the oracle's pieces."""
    import ctypes as C
    sc = scene
    ids = [0, 1]
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps={1: sc.gt_depth[1].astype(np.float32)})
    for f in (2, 4):
        img, K, dep, Kd = pr.ref_scale_view(views[1], f)
        assert np.array_equal(img, po.resize_area(sc.gray[1], f)) and np.array_equal(dep, po.resize_area(sc.gt_depth[1].astype(np.float32), f))
        h, w = sc.gray[1].shape; nh, nw = img.shape
        K0 = np.asarray(sc.K[1], np.float64)
        sx, sy = np.float64(nw) / np.float64(w), np.float64(nh) / np.float64(h)
        want = np.array([[K0[0, 0] * sx, K0[0, 1] * sx, (K0[0, 2] + 0.5) * sx - 0.5], [0, K0[1, 1] * sy, (K0[1, 2] + 0.5) * sy - 0.5], [0, 0, 1]])
        assert np.array_equal(K, want) and np.array_equal(Kd, want)


def test_threaded_reference_run_is_statistically_the_sequential_one(scene):
    """scene.nMaxThreads = 4: one estimator per thread on the shared pixel counter (SceneDensify.cpp:631-750) -- the reference's own, racy, schedule.  Not reproducible
    bit for bit (the reference is not either); the maps must agree with the sequential ones on most pixels.  This is the mode bench.py times as cpu_baseline."""
    sc = scene; v = 1
    ids = [v] + list(sc.neighbors[v])
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
    seq = pr.ref_estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), po.default_opt(seed=3, viewID=v, rngMode=2, nThreads=1))
    par = pr.ref_estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), po.default_opt(seed=3, viewID=v, rngMode=2, nThreads=4))
    both = (seq[0] > 0) & (par[0] > 0)
    assert both.mean() > 0.6 and abs(float((seq[0] > 0).mean()) - float((par[0] > 0).mean())) < 0.05
    rel = np.abs(seq[0][both] - par[0][both]) / seq[0][both]
    assert np.median(rel) < 2e-3

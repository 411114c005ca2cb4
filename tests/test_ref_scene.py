"""Pins the view selection (openmvs_amd/csrc/mvs_front.cpp: mvsf_select_neighbor_views; openmvs_amd/views.py) to the REFERENCE'S OWN TEXT: Scene::SelectNeighborViews and
Scene::FilterNeighborViews (libs/MVS/Scene.cpp:801-934, :953-968) cut verbatim and compiled with the camera members they call (oracle/ref/ref_scene_harness.cpp ->
oracle/_ref/libref_scene.so), on the reference's own pipeline-test scene (apps/Tests/data/scene.mvs, copied as data to tests/data/scene).  Integers (neighbour order,
shared point counts, kept points) must be identical; the float fields are compared exactly against the C++ front end (same libm) and to a few ulps against numpy."""
import os

import numpy as np
import pytest

from openmvs_amd import mvsfront, mvsi, views
from oracle import pyref as pr

pytestmark = pytest.mark.skipif(not pr.scene_available(), reason="oracle/_ref/libref_scene.so not built (needs /root/reference)")
HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "data", "scene", "scene.mvs")


@pytest.fixture(scope="module")
def scene():
    cf = mvsfront.SceneFront(SCENE); py = mvsi.load(SCENE)
    return cf, py


def _inputs(cf, sizes=None):
    n = cf.n_images
    szs = [(cf.image_info(i)[1], cf.image_info(i)[2]) for i in range(n)] if sizes is None else list(sizes)
    cams = [cf.camera(i, szs[i] if sizes is not None else (0, 0)) for i in range(n)]
    pts = np.zeros((cf.n_points, 3), np.float32); pv = []
    for i in range(cf.n_points):
        X, v = cf.point(i); pts[i] = X; pv.append(np.asarray(v, np.uint32))
    return cams, szs, pts, pv


@pytest.mark.parametrize("sizes", [None, [(320, 240)] * 4])
def test_select_neighbor_views_is_the_reference_function(scene, sizes):
    cf, py = scene
    cams, szs, pts, pv = _inputs(cf, sizes)
    pcams = views.Cameras(py, sizes)
    for ID in range(cf.n_images):
        for nMinPointViews, angle in ((2, 12.0), (3, 12.0), (2, 7.5)):
            okr, nbr, ptr, avgr = pr.ref_select_neighbor_views(cams, szs, pts, pv, ID, 2, nMinPointViews, angle, 1)
            okc, nbc, ptc, avgc = cf.select_neighbor_views(ID, nMinViews=2, nMinPointViews=nMinPointViews, fOptimAngle=angle, nInsideROI=1, sizes=sizes)
            assert okr == okc and np.array_equal(ptr, ptc), "C++ front end: kept points of image %d" % ID
            assert len(nbr) == len(nbc) and len(nbr) >= 2
            for k in ("ID", "points"):
                assert np.array_equal(nbr[k], nbc[k]), ("C++ front end", ID, k)
            for k in ("scale", "angle", "area", "score"):
                assert np.array_equal(nbr[k], nbc[k]), ("C++ front end", ID, k, nbr[k], nbc[k])
            assert np.float32(avgr) == np.float32(avgc)
            okp, nbp, ptp, avgp = views.select_neighbor_views(py, pcams, ID, 2, nMinPointViews, np.deg2rad(np.float32(angle)), 1)
            assert okr == okp and np.array_equal(ptr, ptp)
            for k in ("ID", "points"):
                assert np.array_equal(nbr[k], nbp[k]), ("numpy", ID, k)
            for k in ("scale", "angle", "area", "score"):
                assert np.allclose(nbr[k], nbp[k], rtol=2e-5, atol=0), ("numpy", ID, k)          # libm's acosf / expf vs numpy's


def test_filter_neighbor_views_is_the_reference_function(scene):
    cf, py = scene
    cams, szs, pts, pv = _inputs(cf)
    r = np.random.RandomState(3)
    for ID in range(cf.n_images):
        ok, nb, _, _ = pr.ref_select_neighbor_views(cams, szs, pts, pv, ID)
        # the real list, and synthetic longer ones that exercise every clause (area / scale / angle bounds, the "more than max(4, 3/4 nMaxViews) remain" rule, the cut)
        lists = [nb]
        for n in (6, 13, 20):
            a = np.zeros(n, pr.VIEW_SCORE)
            a["ID"] = np.arange(n); a["points"] = r.randint(3, 500, n)
            a["scale"] = r.uniform(0.1, 3.6, n).astype(np.float32); a["angle"] = r.uniform(0.0, 1.3, n).astype(np.float32)
            a["area"] = r.uniform(0.0, 0.3, n).astype(np.float32); a["score"] = np.sort(r.uniform(0, 50, n).astype(np.float32))[::-1]
            lists.append(a)
        for a in lists:
            for nMax in (12, 4, 8):
                args = (0.05, 0.2, 3.2, float(np.float32(np.deg2rad(np.float32(3.0)))), float(np.float32(np.deg2rad(np.float32(65.0)))), nMax)
                want = pr.ref_filter_neighbor_views(a, *args)
                got = views.filter_neighbor_views(a.astype(views.VIEW_SCORE_DTYPE), *args)
                assert np.array_equal(want["ID"], got["ID"]), (ID, nMax, want["ID"], got["ID"])


def test_select_views_cut_matches_the_reference_pieces(scene):
    """DepthMapsData::SelectViews + the score cut of InitViews (SceneDensify.cpp:273-293, :333-340) = SelectNeighborViews, FilterNeighborViews and a short loop: the C++ front
    end's mvsf_select_views against the same composition made of the reference's two functions."""
    cf, py = scene
    cams, szs, pts, pv = _inputs(cf)
    opt = mvsfront.default_options()
    for ID in range(cf.n_images):
        ok, nb, keep, avg = pr.ref_select_neighbor_views(cams, szs, pts, pv, ID, opt.nMinViews, max(2, opt.nMinViewsTrustPoint), opt.fOptimAngle, opt.nPointInsideROI)
        nb = pr.ref_filter_neighbor_views(nb, opt.fMinArea, 0.2, 3.2, float(np.float32(np.deg2rad(np.float32(opt.fMinAngle)))), float(np.float32(np.deg2rad(np.float32(opt.fMaxAngle)))), opt.nMaxViews)
        fMin = max(np.float32(nb[0]["score"]) * np.float32(opt.fViewMinScoreRatio), np.float32(opt.fViewMinScore))
        cut = len(nb)
        for i in range(len(nb)):
            if (opt.nNumViews and i + 1 > opt.nNumViews) or nb[i]["score"] < fMin:
                cut = i; break
        got = cf.select_views(ID)
        assert got is not None and np.array_equal(got[0]["ID"], nb[:cut]["ID"]) and np.array_equal(got[0]["score"], nb[:cut]["score"]) and np.array_equal(got[1], keep)

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


def test_pixel_cameras_are_the_reference_chain(scene, tmp_path):
    """Scene::LoadInterface's normalisation of K, Platform::GetCamera (R = Rc Rp, C = Rp^T Cc + Cp) and Camera::GetK at a working resolution, in the reference's own
    text, against mvsf_camera and mvsi.Scene.camera: the archive's scene, a rig whose platform camera is rotated and offset, a camera stored normalised, and one
    without a principal point (ComposeK puts it at the image centre)."""
    cf, py = scene
    a, b, c = 0.31, -0.17, 0.09
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    import copy
    variants = {}
    variants["archive"] = py
    rig = copy.deepcopy(py); rig.platforms[0].cameras[0].R = Rz @ Ry @ Rx; rig.platforms[0].cameras[0].C = np.array([0.013, -0.027, 0.041]); variants["rig"] = rig
    nrm = copy.deepcopy(py); cam = nrm.platforms[0].cameras[0]
    s = 1.0 / float(np.float32(max(cam.width, cam.height)))
    cam.K = np.array([[cam.K[0][0] * s, 0, (cam.K[0][2] + 0.5) * s - 0.5], [0, cam.K[1][1] * s, (cam.K[1][2] + 0.5) * s - 0.5], [0, 0, 1]]); cam.width = cam.height = 0
    variants["normalised"] = nrm
    ctr = copy.deepcopy(nrm); k = ctr.platforms[0].cameras[0].K; k[0][2] = 0.0; k[1][2] = 0.0; variants["no principal point"] = ctr
    for name, sc in variants.items():
        p = str(tmp_path / (name.replace(" ", "_") + ".mvs"))
        mvsi.save(p, sc)
        c2, p2 = mvsfront.SceneFront(p), mvsi.load(p)
        pcam = p2.platforms[0].cameras[0]
        for i in range(c2.n_images):
            im = p2.images[i]
            Rp, Cp = p2.platforms[0].poses_R[im.pose_id], p2.platforms[0].poses_C[im.pose_id]
            for size in ((640, 479), (321, 240), (1280, 958), (479, 640)):
                want = pr.ref_pixel_camera(pcam.K, pcam.R, pcam.C, (pcam.width, pcam.height), Rp, Cp, size)
                got_c = c2.camera(i, size)
                got_p = p2.camera(i, size)[:3]
                for w_, gc, gp, what in zip(want, got_c, got_p, "KRC"):
                    assert np.array_equal(w_, gc), (name, i, size, what, "C++ front end")
                    assert np.array_equal(w_, gp), (name, i, size, what, "numpy front end")

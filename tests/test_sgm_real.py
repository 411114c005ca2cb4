"""The SGM path end to end on the reference's own pipeline-test scene (tests/data/scene), CPU only: stereo rectification of two image pairs
(openmvs_amd/rectify.py), the tSGM coarse-to-fine loop (openmvs_amd/tsgm.py, driven on the oracle backend), ProjectDisparity2DepthMap and the
per-pixel pair fusion.  The resulting depth maps are compared with the SfM points of the scene, which never entered the computation: this is
the real-data check of the restated chain (the device runs the same loop through tests/test_gpu_sgm_post.py)."""
import os

import numpy as np
import pytest

from openmvs_amd import mvsi, rectify, tsgm, views
from oracle import pyoracle as po
from tests.tsgm_backends import OracleBackend

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "data", "scene")


def _gray(b):
    f = np.float32
    return (f(0.114) * (b[..., 0].astype(f) / f(255)) + f(0.587) * (b[..., 1].astype(f) / f(255))) + f(0.299) * (b[..., 2].astype(f) / f(255))


@pytest.fixture(scope="module")
def scene():
    from PIL import Image
    sc = mvsi.load(os.path.join(SCENE, "scene.mvs"))
    cams = views.Cameras(sc)
    bgr = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(SCENE, im.name)).convert("RGB"))[..., ::-1]) for im in sc.images]
    own = np.repeat(np.arange(len(sc.vertices)), np.diff(sc.vertex_view_start)); ids = sc.vertex_views["image_id"]
    seen = []
    for i in range(len(sc.images)):
        s = np.zeros(len(sc.vertices), bool); s[own[ids == i]] = True; seen.append(s)
    return sc, cams, bgr, seen


def _w2i3(cams, i, X):
    """Camera::TransformPointW2I3 (libs/MVS/Camera.h:396-399)."""
    cx = (X - cams.C[i]) @ cams.R[i].T; K = cams.K[i]
    return np.stack([K[0, 2] + K[0, 0] * cx[:, 0] / cx[:, 2], K[1, 2] + K[1, 1] * cx[:, 1] / cx[:, 2], cx[:, 2]], 1).astype(np.float32)


def _pair_depth(scene, A, B):
    sc, cams, bgr, seen = scene
    X = sc.vertices[seen[A] & seen[B]].astype(np.float64)
    p1, p2 = _w2i3(cams, A, X), _w2i3(cams, B, X)
    r = rectify.stereo_rectify_images(bgr[A], cams.K[A], cams.R[A], cams.C[A], bgr[B], cams.K[B], cams.R[B], cams.C[B], p1, p2)
    assert r is not None and abs(r["t"]) > 0.5
    # rectified: corresponding points share their row, and the rotations are rotations
    a = rectify._project_h(r["H"], p1[:, :2])
    b = rectify._project_h(r["K2"] @ r["R2"] @ rectify._inv_k(cams.K[B]), p2[:, :2])
    assert np.abs(a[:, 1] - b[:, 1]).max() < 1e-3
    assert np.allclose(r["R1"] @ r["R1"].T, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(r["R1"]), 1)
    # Q maps (x', y', d) of the rectified left image to the depth in the original image: exact on the sparse points themselves
    d = b[:, 0] - a[:, 0]
    xs = np.rint(a[:, 0]).astype(int); ys = np.rint(a[:, 1]).astype(int)
    Q = r["Q"]
    wq = Q[3, 0] * xs + Q[3, 1] * ys - Q[3, 2] * d + Q[3, 3]; zq = (Q[2, 0] * xs + Q[2, 1] * ys - Q[2, 2] * d + Q[2, 3]) / wq
    assert np.median(np.abs(zq - p1[:, 2]) / p1[:, 2]) < 2e-3
    w, h = r["size"]; w4, h4 = w // 4 * 4, h // 4 * 4
    lb, rb = r["rect1"][:h4, :w4].copy(), r["rect2"][:h4, :w4].copy()
    disp, cost, levels = tsgm.tsgm_match(OracleBackend(), lb, _gray(lb), rb, _gray(rb), r["mask1"][:h4, :w4].copy(), r["mask2"][:h4, :w4].copy(), min_resolution=160)
    assert levels == 3 and (disp != tsgm.NO_DISP).mean() > 0.4
    H0, W0 = bgr[A].shape[:2]
    ok, dep, rg, cf = po.sgm_project_disparity2depth_map(disp, cost, r["Q"], 4, (W0, H0))
    assert ok and (dep > 0).mean() > 0.4
    return dep, rg, cf


def _against_sfm(scene, A, dep):
    sc, cams, bgr, seen = scene
    p = _w2i3(cams, A, sc.vertices[seen[A]].astype(np.float64))
    xi = np.rint(p[:, 0]).astype(int); yi = np.rint(p[:, 1]).astype(int)
    m = (xi >= 0) & (yi >= 0) & (xi < dep.shape[1]) & (yi < dep.shape[0])
    dm = dep[yi[m], xi[m]]; v = dm > 0
    rel = np.abs(dm[v] - p[m][v, 2]) / p[m][v, 2]
    return v.mean(), np.median(rel), np.percentile(rel, 90)


def test_sgm_chain_matches_the_sfm_points(scene):
    pairs = [_pair_depth(scene, 0, 2), _pair_depth(scene, 0, 3)]
    for dep, rg, cf in pairs:
        cover, med, p90 = _against_sfm(scene, 0, dep)
        assert cover > 0.6 and med < 6e-3 and p90 < 3e-2, (cover, med, p90)
        good = dep > 0
        assert np.all(rg[good].min(1) <= dep[good] * 1.001) and np.all(rg[good].max(1) >= dep[good] * 0.999)   # the depths at disparity -1 / +1 bracket it
    fused, conf = po.sgm_fuse_pairs([p[0] for p in pairs], [p[1] for p in pairs], [p[2] for p in pairs], minViews=2)
    cover, med, p90 = _against_sfm(scene, 0, fused)
    assert (fused > 0).mean() > 0.25 and cover > 0.4 and med < 5e-3 and p90 < 2e-2, (cover, med, p90)


def test_sgm_pipeline_module(scene, tmp_path):
    """openmvs_amd/sgm_pipeline.py (the orchestration a host would call) gives the same maps, and its pair data survives a .dimap round trip."""
    from openmvs_amd import dmap, sgm_pipeline
    sc, cams, bgr, seen = scene
    be = OracleBackend()
    cam = lambda i: (cams.K[i], cams.R[i], cams.C[i])
    pairs = []
    for B in (2, 3):
        p = sgm_pipeline.match_pair(be, bgr[0], cam(0), bgr[B], cam(B), sc.vertices[seen[0] & seen[B]], min_resolution=160)
        assert p is not None
        f = str(tmp_path / ("0000_%04d.dimap" % B))
        dmap.save_dimap(f, p["image_size"], p["H"], p["Q"], p["subpixel_steps"], p["disparity"], p["cost"])
        g = dmap.load_dimap(f)
        assert np.array_equal(g["disparity"], p["disparity"]) and np.array_equal(g["cost"], p["cost"]) and np.array_equal(g["Q"], p["Q"])
        pairs.append(dict(p, disparity=g["disparity"], cost=g["cost"]))
    depth, conf = sgm_pipeline.fuse_pairs(be, pairs, 2)
    cover, med, p90 = _against_sfm(scene, 0, depth)
    assert cover > 0.4 and med < 5e-3 and p90 < 2e-2
    assert conf[depth > 0].min() > 0 and not conf[depth == 0].any()


def test_seeded_first_level(scene):
    """The rough depth map from the sparse points (corners + dense, SemiGlobalMatcher.cpp:608-625) seeds the first level: the initial disparities
    already agree with the matched ones, both triangulation implementations give the same seed, and the seeded result is at least as good."""
    from openmvs_amd import mvsfront, sgm_pipeline
    sc, cams, bgr, seen = scene
    be = OracleBackend()
    cam = lambda i: (cams.K[i], cams.R[i], cams.C[i])
    ok, nb, pts, avg = views.select_neighbor_views(sc, cams, 0)
    pts = np.nonzero(seen[0])[0].astype(np.uint32)                   # Match() collects every point the image sees (:553-556)
    cf = mvsfront.SceneFront(os.path.join(SCENE, "scene.mvs"))
    seeds = {}

    def seed(w, h):
        K, R, C, _, _ = sc.camera(0, (w, h))
        P = K @ np.hstack([R, -(R @ C)[:, None]])
        d, mn, mx = views.triangulate_points_depth_map(K, P, sc.vertices[pts], w, h, avg_depth=avg)
        dc, mnc, mxc = cf.triangulate_depth_map(0, pts, (w, h), avg_depth=avg)
        assert np.array_equal(d, dc) and (mn, mx) == (mnc, mxc) and (d > 0).all()
        seeds[(w, h)] = d
        return d
    X = sc.vertices[seen[0] & seen[2]]
    p = sgm_pipeline.match_pair(be, bgr[0], cam(0), bgr[2], cam(2), X, min_resolution=160, seed_depth=seed)
    q = sgm_pipeline.match_pair(be, bgr[0], cam(0), bgr[2], cam(2), X, min_resolution=160)
    assert p["seeded"] and not q["seeded"] and list(seeds) == [(80, 60)]
    # the seed itself, pushed through Depth2DisparityMap, predicts the final disparities (x 2^3 levels, x 4 sub-pixel steps)
    r = rectify.stereo_rectify_images(bgr[0], *cam(0), bgr[2], *cam(2), sgm_pipeline.world_to_image3(*cam(0), X), sgm_pipeline.world_to_image3(*cam(2), X))
    H2, Q2 = rectify.scale_stereo_rectification(r["H"], r["Q"], 0.125)
    w, h = r["size"][0] // 4 * 4, r["size"][1] // 4 * 4
    hw, hh = sgm_pipeline.compute_resize((w // 4, h // 4), 0.5)
    init = be.Depth2DisparityMap(seeds[(80, 60)], np.linalg.inv(H2), np.linalg.inv(Q2), 1, (hw - 6, hh - 6))
    assert (init != tsgm.NO_DISP).mean() > 0.7
    fin = p["disparity"]
    ys, xs = np.nonzero(init != tsgm.NO_DISP)
    fy, fx = np.minimum(ys * 8 + 24, fin.shape[0] - 1), np.minimum(xs * 8 + 24, fin.shape[1] - 1)       # (r + 3) * 8 - 3 + ... roughly the same place
    v = fin[fy, fx] != tsgm.NO_DISP
    err = np.abs(init[ys, xs][v] * 8.0 - fin[fy, fx][v] / 4.0)
    assert v.mean() > 0.5 and np.median(err) < 8                     # within one coarse-level disparity step
    for m in (p, q):
        ok, dep, rg, cf_ = po.sgm_project_disparity2depth_map(m["disparity"], m["cost"], m["Q"], 4, m["image_size"])
        m["stats"] = _against_sfm(scene, 0, dep)
    assert p["stats"][0] >= q["stats"][0] - 0.02 and p["stats"][1] < 6e-3, (p["stats"], q["stats"])


def test_scene_level_match_and_fuse(scene, tmp_path):
    """`--fusion-mode -1` then `-2` over the scene (sgm_pipeline.match_scene / fuse_scene): each image matched against its best neighbour, a pair stored once whichever
    image asked first, the other image fusing it through the swapped-pair Q (SemiGlobalMatcher.cpp:767-777); both land on the SfM points."""
    from openmvs_amd import sgm_pipeline
    sc, cams, bgr, seen = scene
    be = OracleBackend()
    nbs, avg = {}, {}
    for i in (0, 2):
        ok, nb, pts, a = views.select_neighbor_views(sc, cams, i)
        assert ok
        nbs[i] = nb; avg[i] = a
    partner = int(nbs[0]["ID"][0])
    nbs = {0: nbs[0], partner: views.select_neighbor_views(sc, cams, partner)[1]}
    avg[partner] = views.select_neighbor_views(sc, cams, partner)[3]
    # make the partner's best neighbour image 0, so that its only pair is the one image 0 writes
    order = np.argsort([0 if int(v["ID"]) == 0 else 1 for v in nbs[partner]], kind="stable")
    nbs[partner] = nbs[partner][order]
    d = str(tmp_path / "pairs")
    done = sgm_pipeline.match_scene(be, sc, cams, bgr, nbs, d, n_views=1, f_min_score=0.0, min_resolution=160, avg_depth=avg)
    assert done == [(0, partner)] and sorted(os.listdir(d)) == [sgm_pipeline.pair_file_name(0, partner)]
    assert sgm_pipeline.match_scene(be, sc, cams, bgr, nbs, d, n_views=1, f_min_score=0.0, min_resolution=160, avg_depth=avg) == []     # nothing left to do
    sizes = {i: (bgr[i].shape[1], bgr[i].shape[0]) for i in nbs}
    fused = sgm_pipeline.fuse_scene(be, cams, sizes, nbs, d, n_views=1, f_min_score=0.0, min_views=1)
    for i in (0, partner):
        depth, normal, conf = fused[i]
        cover, med, p90 = _against_sfm(scene, i, depth)
        assert (depth > 0).mean() > 0.3 and cover > 0.5 and med < 6e-3 and p90 < 3e-2, (i, cover, med, p90)
        assert normal.shape == depth.shape + (3,) and (np.linalg.norm(normal[depth > 0], axis=-1) > 0.99).mean() > 0.8 and not normal[depth == 0].any()
        assert conf[depth > 0].min() > 0
    # the swapped Q really is another matrix, and without it the partner's map would be wrong
    from openmvs_amd import dmap
    g = dmap.load_dimap(os.path.join(d, sgm_pipeline.pair_file_name(0, partner)))
    cam = lambda i: (cams.K[i], cams.R[i], cams.C[i])
    Qs = sgm_pipeline.swapped_pair_q(g["Q"], cam(partner), cam(0))
    assert not np.allclose(Qs, g["Q"]) and np.allclose(sgm_pipeline.swapped_pair_q(g["Q"], cam(0), cam(0)), g["Q"], rtol=1e-9, atol=1e-9)


def test_sgm_modes_of_dense_reconstruction(tmp_path):
    """sgm_pipeline.dense_reconstruction: `--fusion-mode -1` then `-2` from the archive, at a quarter of the resolution: pair files once per pair, then depthNNNN.dmap with
    depth, estimated normals and confidence that the .dmap reader takes back, on the SfM points."""
    from openmvs_amd import dmap, optdense, sgm_pipeline
    opt = optdense.defaults()
    opt.nResolutionLevel = 2; opt.nMinResolution = 80; opt.nNumViews = 2; opt.nEstimateNormals = 2; opt.fViewMinScore = 0.0     # (Fuse wants two agreeing pairs per pixel, :2053)
    mvs = os.path.join(SCENE, "scene.mvs")
    d = str(tmp_path / "sgm")
    be = OracleBackend()
    done = sgm_pipeline.dense_reconstruction(be, mvs, d, -1, opt, min_resolution=40)
    assert 4 <= len(done) <= 6 and sorted(os.listdir(d)) == sorted(sgm_pipeline.pair_file_name(a, b) for a, b in done)
    assert not any((b, a) in done for a, b in done)                        # a pair is matched once
    assert sgm_pipeline.dense_reconstruction(be, mvs, d, -1, opt, min_resolution=40) == []
    fused = sgm_pipeline.dense_reconstruction(be, mvs, d, -2, opt)
    assert sorted(fused) == [0, 1, 2, 3]
    sc = mvsi.load(mvs)
    for i in fused:
        f = dmap.load(os.path.join(d, "depth%04d.dmap" % i))
        depth, normal, conf = fused[i]
        assert f["depth_map"].shape == (120, 160) and np.array_equal(f["depth_map"], depth) and np.array_equal(f["normal_map"], normal) and np.array_equal(f["confidence_map"], conf)
        assert f["depth_min"] == np.float32(1e-4) and f["reference_view_id"] == i
        K, R, C, _, _ = sc.camera(i, (160, 120))
        X = sc.vertices.astype(np.float64)
        cx = (X - C) @ R.T
        u = np.rint(K[0, 2] + K[0, 0] * cx[:, 0] / cx[:, 2]).astype(int); v = np.rint(K[1, 2] + K[1, 1] * cx[:, 1] / cx[:, 2]).astype(int)
        m = (cx[:, 2] > 0) & (u >= 0) & (v >= 0) & (u < 160) & (v < 120)
        dm = depth[v[m], u[m]]; ok = dm > 0
        rel = np.abs(dm[ok] - cx[m][ok, 2]) / cx[m][ok, 2]
        assert (depth > 0).mean() > 0.2 and ok.mean() > 0.3 and np.median(rel) < 2e-2, (i, (depth > 0).mean(), ok.mean(), np.median(rel))

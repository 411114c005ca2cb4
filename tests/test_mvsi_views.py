"""CPU tests of the scene front end: the MVSI archive reader/writer against a golden produced by the reference's own reader
(tests/golden/make_mvsi_golden.py), and the view-selection / depth-initialisation host logic on the reference's pipeline-test scene."""
import hashlib
import json
import os

import numpy as np
import pytest

from openmvs_amd import densify, mvsi, views

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "data", "scene", "scene.mvs")


def _sha(a, dt):
    return hashlib.sha256(np.ascontiguousarray(a, dt).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def scene():
    return mvsi.load(SCENE)


def test_reader_matches_the_reference_reader_golden(scene):
    with open(os.path.join(HERE, "golden", "mvsi_golden.json")) as f:
        g = json.load(f)
    assert scene.version == g["version"]
    assert len(scene.platforms) == len(g["platforms"])
    for pl, gp in zip(scene.platforms, g["platforms"]):
        assert pl.name == gp["name"] and len(pl.cameras) == len(gp["cameras"])
        for cam, gc in zip(pl.cameras, gp["cameras"]):
            assert (cam.name, cam.width, cam.height) == (gc["name"], gc["width"], gc["height"])
            assert np.array_equal(cam.K, np.array(gc["K"]))
            assert np.array_equal(pl.poses_R, np.array([p["R"] for p in gc["poses"]]))
            assert np.array_equal(pl.poses_C, np.array([p["C"] for p in gc["poses"]]))
    assert len(scene.images) == len(g["images"])
    for im, gi in zip(scene.images, g["images"]):
        assert (im.name, im.mask_name, im.platform_id, im.camera_id, im.pose_id, im.id) == \
               (gi["name"], gi["mask_name"], gi["platform_id"], gi["camera_id"], gi["pose_id"], gi["id"])
    assert len(scene.vertices) == g["n_vertices"]
    assert _sha(scene.vertices, "<f4") == g["vertices_sha256"]
    assert _sha(np.diff(scene.vertex_view_start), "<i8") == g["views_per_vertex_sha256"]
    assert _sha(scene.vertex_views["image_id"], "<u4") == g["view_image_ids_sha256"]
    assert _sha(scene.vertex_views["confidence"], "<f4") == g["view_confidences_sha256"]
    assert _sha(scene.vertices_color, "u1") == g["vertices_color_sha256"]
    assert len(scene.vertices_normal) == g["n_normals"] and len(scene.lines) == g["n_lines"]
    assert np.array_equal(scene.transform, np.array(g["transform"]))
    assert np.array_equal(scene.obb_rot, np.array(g["obb"]["rot"])) and np.array_equal(scene.obb_max, np.array(g["obb"]["pt_max"]))
    assert [float(x) for x in scene.vertices[0]] == g["first_vertex"]["X"]
    assert [int(v) for v in scene.views_of(len(scene.vertices) - 1)["image_id"]] == [w["image_id"] for w in g["last_vertex"]["views"]]


def test_writer_round_trip_is_byte_identical_and_all_versions_reload(scene, tmp_path):
    p = str(tmp_path / "copy.mvs")
    mvsi.save(p, scene)
    with open(p, "rb") as a, open(SCENE, "rb") as b:
        assert a.read() == b.read()
    assert not os.path.exists(p + ".tmp")
    for v in range(0, mvsi.MVSI_PROJECT_VER + 1):
        sc = scene
        if v > 6:       # exercise the stored view scores of version 7
            sc = mvsi.load(SCENE)
            sc.images[0].view_scores = np.array([(2, 999, 1.04, 0.15, 0.62, 487.9), (3, 762, 1.03, 0.24, 0.59, 422.2)], mvsi.VIEW_SCORE_DTYPE)
            sc.images[0].min_depth, sc.images[0].avg_depth, sc.images[0].max_depth = 5.0, 8.5, 27.0
        mvsi.save(p, sc, version=v)
        s2 = mvsi.load(p)
        assert s2.version == v and np.array_equal(s2.vertices, scene.vertices) and np.array_equal(s2.vertex_views, scene.vertex_views)
        assert np.array_equal(s2.platforms[0].poses_C, scene.platforms[0].poses_C)
        if v > 6:
            assert np.array_equal(s2.images[0].view_scores, sc.images[0].view_scores) and s2.images[0].avg_depth == 8.5
        if v > 5:
            assert np.array_equal(s2.obb_rot, scene.obb_rot)


def test_reader_rejects_bad_archives(tmp_path):
    raw = open(SCENE, "rb").read()
    p = str(tmp_path / "bad.mvs")
    open(p, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(ValueError):
        mvsi.load(p)
    open(p, "wb").write(raw[:4] + (99).to_bytes(4, "little") + raw[8:])
    with pytest.raises(ValueError):
        mvsi.load(p)
    q = str(tmp_path / "bad.bin")
    open(q, "wb").write(b"XXXX" + raw[4:])
    with pytest.raises(ValueError):
        mvsi.load(q)


def test_camera_composition(scene):
    K, R, C, w, h = scene.camera(2)
    assert (w, h) == (640, 479)
    assert np.allclose(K, scene.platforms[0].cameras[0].K, rtol=0, atol=1e-9)      # normalise/de-normalise round trip
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and np.isclose(np.linalg.det(R), 1, atol=1e-6)
    K2, _, _, w2, h2 = scene.camera(2, (320, 240))                                      # ScaleK half-pixel convention
    assert np.isclose(K2[0, 0], K[0, 0] / 2) and np.isclose(K2[0, 2], (K[0, 2] + 0.5) / 2 - 0.5)
    cams = views.Cameras(scene)
    X = scene.vertices[:50]
    assert np.all(cams.point_depth(0, X[scene.vertex_views["image_id"][scene.vertex_view_start[:50]] == 0]) > 0)
    assert not scene.is_bounded()


def test_select_neighbor_views_on_the_reference_scene(scene):
    cams = views.Cameras(scene)
    expect_first = {0: 2, 1: 3, 2: 0, 3: 1}          # the neighbour closest to the 12 degree optimum wins in this orbit
    for ID in range(4):
        ok, nb, points, avg = views.select_neighbor_views(scene, cams, ID)
        assert ok and len(nb) == 3 and set(nb["ID"]) == set(range(4)) - {ID}
        assert np.all(np.diff(nb["score"]) <= 0) and nb["ID"][0] == expect_first[ID]
        assert np.all((nb["area"] > 0.3) & (nb["area"] <= 1)) and np.all((nb["scale"] > 0.9) & (nb["scale"] < 1.1))
        assert np.all((nb["angle"] > np.deg2rad(3)) & (nb["angle"] < np.deg2rad(20)))
        # every returned point is seen by ID and by >= 2 views; the counts are the shared-point counts
        seen_by = [set(scene.views_of(int(p))["image_id"]) for p in points]
        assert all(ID in s and len(s) >= 2 for s in seen_by)
        for n in nb:
            shared = sum(1 for v in range(len(scene.vertices)) if {ID, int(n["ID"])} <= set(scene.views_of(v)["image_id"]))
            dup = sum(list(scene.views_of(v)["image_id"]).count(int(n["ID"])) - 1 for v in range(len(scene.vertices))
                      if {ID, int(n["ID"])} <= set(scene.views_of(v)["image_id"]))
            assert n["points"] == shared + dup
        assert 5 < avg < 15
    # symmetry of the pair statistics: angle(A,B) == angle(B,A) up to float summation order
    _, nb0, _, _ = views.select_neighbor_views(scene, cams, 0)
    _, nb2, _, _ = views.select_neighbor_views(scene, cams, 2)
    assert np.isclose(nb0[nb0["ID"] == 2]["angle"][0], nb2[nb2["ID"] == 0]["angle"][0], rtol=1e-3)


def test_filter_neighbor_views_rules():
    nb = np.zeros(14, mvsi.VIEW_SCORE_DTYPE)
    nb["ID"] = np.arange(14); nb["score"] = np.arange(14, 0, -1); nb["area"] = 0.5; nb["scale"] = 1; nb["angle"] = 0.2
    nb["area"][3] = 0.01; nb["scale"][5] = 4.0; nb["angle"][13] = 0.01
    out = views.filter_neighbor_views(nb)
    assert list(out["ID"]) == [0, 1, 2, 4, 6, 7, 8, 9, 10, 11, 12]          # 3, 5, 13 dropped; still above the floor of 9
    few = nb[:5].copy()                                                        # at most max(4, 9) = 9 -> nothing is dropped
    assert list(views.filter_neighbor_views(few)["ID"]) == [0, 1, 2, 3, 4]
    assert len(views.filter_neighbor_views(np.concatenate([nb, nb]), nMaxViews=12)) == 12


def test_resolution_rules():
    assert views.compute_max_resolution(640, 479, 1, 640, 3200) == (640, 0)         # too small to halve
    assert views.compute_max_resolution(4000, 3000, 1, 640, 3200) == (2000, 1)
    assert views.compute_max_resolution(4000, 3000, 0, 640, 3200) == (3200, 0)
    assert views.compute_max_resolution(2000, 1000, 3, 640, 3200) == (1000, 1)
    assert views.resized_size(4000, 3000, 2000) == (2000, 1500)
    assert views.resized_size(640, 479, 640) == (640, 479)
    img = (np.arange(8 * 12 * 3) % 251).astype(np.uint8).reshape(8, 12, 3)
    half = densify._resize_area_u8(img, 6, 4)
    blocks = img.reshape(4, 2, 6, 2, 3).astype(int).sum(axis=(1, 3))
    assert half.shape == (4, 6, 3) and np.array_equal(half, (blocks + 2) >> 2)     # OpenCV's integer rule for a factor of 2: (a + b + c + d + 2) >> 2, halves round UP
    ties = np.array([[1, 0], [1, 0]], np.uint8)                                     # mean 0.5: 1 under that rule (0 if it were rounded to even)
    assert densify._resize_area_u8(np.tile(ties, (2, 2)), 2, 2).tolist() == [[1, 1], [1, 1]]
    third = densify._resize_area_u8(np.tile(np.array([[1, 0, 0], [0, 0, 0], [0, 0, 0]], np.uint8) * 4 + np.uint8(0), (1, 1)), 1, 1)   # factor 3: sum * (1.f / 9) rounded to even: 4/9 -> 0
    assert third.tolist() == [[0]]
    with pytest.raises(NotImplementedError):
        densify._resize_area_u8(img, 13, 4)                                          # enlarging is OpenCV's bilinear path: not restated


def test_area_resize_by_a_non_integer_factor():
    """cv::resize(INTER_AREA) for a factor that is not an integer (Image::ResizeImage under --max-resolution, libs/MVS/Image.cpp:139-155): densify's vectorised restatement of
    OpenCV's general area path against a literal per-pixel walk of the same published algorithm (computeResizeAreaTab + ResizeArea_Invoker: x cells per source row in table
    order, then the rows per destination row, everything in float), plus what any area resize must do.  OpenCV itself is not vendored with the reference: unpinned."""
    r = np.random.RandomState(3)
    img = r.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    w, h = 20, 14
    out = densify._resize_area_u8(img, w, h)
    xd, xs, xa = densify._area_tab(53, w); yd, ys, ya = densify._area_tab(37, h)
    assert abs(float(xa[xd == 0].sum()) - 1) < 1e-6 and abs(float(ya[yd == h - 1].sum()) - 1) < 1e-6      # a cell's weights sum to one
    ref = np.zeros((h, w, 3), np.uint8)
    acc = np.zeros((w, 3), np.float32); prev = yd[0]
    for j in range(len(yd)):
        buf = np.zeros((w, 3), np.float32)
        S = img[ys[j]].astype(np.float32)
        for k in range(len(xd)):
            buf[xd[k]] = buf[xd[k]] + S[xs[k]] * xa[k]
        if yd[j] != prev:
            ref[prev] = np.clip(np.rint(acc), 0, 255).astype(np.uint8); acc = ya[j] * buf; prev = yd[j]
        else:
            acc = acc + ya[j] * buf
    ref[prev] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    assert np.array_equal(out, ref)
    assert np.unique(densify._resize_area_u8(np.full((40, 64, 3), 77, np.uint8), 25, 17)).tolist() == [77]   # a constant image stays constant
    assert abs(float(out.mean()) - float(img.mean())) < 0.5                                                  # the mean survives
    g = densify._resize_area_u8(img[..., 0].copy(), w, h)
    assert g.shape == (h, w) and np.array_equal(g, out[..., 0])                                              # single-channel images too
    # the size rule that leads here: a 4000x3000 image under --max-resolution 3200 becomes 3200x2400 (factor 1.25)
    assert views.resized_size(4000, 3000, 3200) == (3200, 2400)


def test_to_gray_is_the_bgr_weighted_sum():
    rgb = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [12, 200, 77]]], np.uint8)
    g = views.to_gray(rgb)[0]
    assert g.dtype == np.float32
    assert np.allclose(g[:4], [0.299, 0.587, 0.114, 1.0], atol=1e-6)
    c = np.float32([77, 200, 12]) * (np.float32(1) / np.float32(255))          # NormRGB_t: times the reciprocal, not a division
    assert g[4] == (np.float32(0.114) * c[0] + np.float32(0.587) * c[1]) + np.float32(0.299) * c[2]


@pytest.mark.parametrize("trust", [2, 1])
def test_init_depth_map_modes(scene, trust):
    cams = views.Cameras(scene)
    opt = views.DenseOptions(nMinViewsTrustPoint=trust)
    nb, points, _ = views.select_views(scene, cams, 1, opt)
    d, n, dmin, dmax = views.init_depth_map(scene, cams, 1, points, opt)
    z = cams.point_depth(1, scene.vertices[points]).astype(np.float32)
    assert np.isclose(dmin, z.min() * 0.9, rtol=1e-6) and np.isclose(dmax, z.max() * 1.1, rtol=1e-6)
    m = d > 0
    assert d.shape == (479, 640) and 0.01 < m.mean() < 0.2 and np.all((d[m] >= z.min()) & (d[m] <= z.max()))
    if trust >= 2:                       # 2x2 splat with unit normals facing the camera
        assert m.sum() <= 4 * len(points)
        nn = np.linalg.norm(n[m], axis=1)
        assert np.all(np.abs(nn - 1) < 1e-5)
        K = cams.K[1]; ys, xs = np.nonzero(m)
        rays = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones(len(xs))], 1)
        assert np.all((rays * n[m]).sum(1) < 0)
    else:                                # 5x5 splat with zero normals (the estimator randomises those)
        assert m.sum() <= 25 * len(points) and np.all(n == 0)
    e = views.init_depth_map(scene, cams, 1, np.zeros(0, np.uint32), opt)
    assert e[2:] == (0.1, 100.0) and not e[0].any()


def test_load_scene_prepares_every_view():
    sv = densify.load_scene(SCENE)
    assert (sv.width, sv.height, sv.n_views, sv.ids) == (640, 479, 4, [0, 1, 2, 3])
    assert [list(map(int, n)) for n in sv.neighbors] == [[2, 3, 1], [3, 2, 0], [0, 1, 3], [1, 0, 2]]
    assert all(g.shape == (479, 640) and g.dtype == np.float32 and 0 <= g.min() and g.max() <= 1 for g in sv.gray)
    assert all(0 < sv.dmin[i] < sv.dmax[i] for i in sv.ids) and set(sv.init_depth) == set(sv.ids)


def test_roi_weighting_and_restriction(scene):
    """nPointInsideROI (Scene.cpp:824-838): 1 down-weights points outside the ROI box (x0.7), 2 ignores them."""
    sc = mvsi.load(SCENE)
    cams = views.Cameras(sc)
    X = sc.vertices
    lo, hi = np.percentile(X, 25, axis=0), np.percentile(X, 75, axis=0)
    sc.obb_rot = np.eye(3); sc.obb_min = lo.astype(np.float64); sc.obb_max = hi.astype(np.float64)
    assert sc.is_bounded()
    inside = sc.roi_contains(X)
    assert 0.05 < inside.mean() < 0.6 and np.array_equal(inside, np.all((X >= lo.astype(np.float32)) & (X <= hi.astype(np.float32)), axis=1))
    _, nb0, p0, _ = views.select_neighbor_views(sc, cams, 0, nInsideROI=0)
    _, nb1, p1, _ = views.select_neighbor_views(sc, cams, 0, nInsideROI=1)
    ok2, nb2, p2, _ = views.select_neighbor_views(sc, cams, 0, nInsideROI=2)
    assert np.array_equal(p0, p1) and len(p2) < len(p1) and set(p2) <= set(p1) and np.all(inside[p2])
    for a, b in zip(np.sort(nb0, order="ID"), np.sort(nb1, order="ID")):      # same statistics, smaller score
        assert a["points"] == b["points"] and a["angle"] == b["angle"] and 0.7 * a["score"] - 1e-3 <= b["score"] < a["score"]
    assert all(n["points"] <= m["points"] for n, m in zip(np.sort(nb2, order="ID"), np.sort(nb1, order="ID")))


def test_archive_with_lines_and_normals_round_trips(scene, tmp_path):
    sc = mvsi.load(SCENE)
    r = np.random.RandomState(0)
    sc.vertices_normal = r.randn(len(sc.vertices), 3).astype(np.float32)
    sc.lines = r.randn(5, 2, 3).astype(np.float32)
    sc.line_view_start = np.array([0, 2, 4, 4, 7, 9], np.int64)
    sc.line_views = np.zeros(9, mvsi.VIEW_DTYPE); sc.line_views["image_id"] = r.randint(0, 4, 9); sc.line_views["confidence"] = r.rand(9)
    sc.lines_normal = r.randn(5, 3).astype(np.float32); sc.lines_color = r.randint(0, 255, (5, 3)).astype(np.uint8)
    sc.transform = r.randn(4, 4)
    p = str(tmp_path / "full.mvs")
    mvsi.save(p, sc, version=7)
    s2 = mvsi.load(p)
    for k in ("vertices_normal", "lines", "line_view_start", "line_views", "lines_normal", "lines_color", "transform"):
        assert np.array_equal(getattr(s2, k), getattr(sc, k)), k
    mvsi.save(p, sc, version=0)                                   # the head-less first format: no lines, no transform
    s0 = mvsi.load(p)
    assert s0.version == 0 and len(s0.lines) == 0 and np.array_equal(s0.vertices_normal, sc.vertices_normal)

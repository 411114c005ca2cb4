"""DepthData::ViewData::ScaleImage in the scene front end (openmvs_amd/densify.py): the float resamplers against literal per-pixel walks of OpenCV's published algorithms
(OpenCV itself is not vendored with the reference: unpinned), resampled copies of neighbours as extra source-only slots, images of different sizes, and the
driver's hand-offs at the round boundaries (recorded on a stand-in engine; the engine side of sized source views with their own stored depth maps is
tests/test_gpu_patchmatch.py::test_mixed_resolution_neighbours_parity)."""
import math
import os

import numpy as np
import pytest

from openmvs_amd import densify, mvsi, views

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "data", "scene", "scene.mvs")
f32 = np.float32


def _area_literal(src, w, h, scale):
    """computeResizeAreaTab + ResizeArea_Invoker for one float channel, statement by statement (imgproc/src/resize.cpp)."""
    def tab(ssize, dsize):
        t = []
        for dx in range(dsize):
            fsx1 = dx * scale; fsx2 = fsx1 + scale
            cell = min(scale, ssize - fsx1)
            sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
            sx2 = min(sx2, ssize - 1); sx1 = min(sx1, sx2)
            if sx1 - fsx1 > 1e-3:
                t.append((dx, sx1 - 1, f32((sx1 - fsx1) / cell)))
            for sx in range(sx1, sx2):
                t.append((dx, sx, f32(1.0 / cell)))
            if fsx2 - sx2 > 1e-3:
                t.append((dx, sx2, f32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
        return t
    H, W = src.shape
    xtab, ytab = tab(W, w), tab(H, h)
    dst = np.zeros((h, w), f32)
    total = np.zeros(w, f32); prev = ytab[0][0]
    for dy, sy, beta in ytab:
        buf = np.zeros(w, f32)
        for dx, sx, alpha in xtab:
            buf[dx] = f32(buf[dx] + f32(src[sy, sx] * alpha))
        if dy != prev:
            dst[prev] = total
            total = (beta * buf).astype(f32)
            prev = dy
        else:
            total = (total + (beta * buf).astype(f32)).astype(f32)
    dst[prev] = total
    return dst


def _cubic_literal(src, w, h, scale):
    """resizeGeneric_ with HResizeCubic / VResizeCubic in scalar form."""
    A = f32(-0.75)

    def coeffs(x):
        c0 = f32(f32(f32(f32(f32(A * f32(x + 1)) - f32(5 * A)) * f32(x + 1)) + f32(8 * A)) * f32(x + 1)) - f32(4 * A)
        c1 = f32(f32(f32(f32(f32(A + 2) * x) - f32(A + 3)) * x) * x) + f32(1)
        y = f32(1 - x)
        c2 = f32(f32(f32(f32(f32(A + 2) * y) - f32(A + 3)) * y) * y) + f32(1)
        return f32(c0), f32(c1), f32(c2), f32(f32(f32(f32(1) - f32(c0)) - f32(c1)) - f32(c2))
    H, W = src.shape
    rows = np.zeros((H, w), f32)
    for dx in range(w):
        fx = f32((dx + 0.5) * scale - 0.5); sx = math.floor(fx); fx = f32(fx - f32(sx))
        c = coeffs(fx)
        for y in range(H):
            v = f32(0)
            for j in range(4):
                sxj = min(max(sx + j - 1, 0), W - 1)
                v = f32(v + f32(src[y, sxj] * c[j]))
            rows[y, dx] = v
    dst = np.zeros((h, w), f32)
    for dy in range(h):
        fy = f32((dy + 0.5) * scale - 0.5); sy = math.floor(fy); fy = f32(fy - f32(sy))
        b = coeffs(fy)
        r = [rows[min(max(sy - 1 + k, 0), H - 1)] for k in range(4)]
        dst[dy] = ((r[0] * b[0] + r[1] * b[1]) + r[2] * b[2]) + r[3] * b[3]
    return dst


@pytest.mark.parametrize("shape,scale", [((23, 31), 0.8), ((23, 31), 0.6), ((19, 17), 0.35), ((16, 24), 0.5), ((17, 25), 0.5), ((18, 21), 1 / 3), ((20, 20), 0.25)])
def test_scale_image_shrinks_with_the_area_paths(shape, scale):
    rng = np.random.default_rng(hash((shape, scale)) & 0xFFFF)
    img = rng.random(shape).astype(f32)
    out = densify.scale_image(img, scale)
    s = float(f32(scale)); H, W = shape
    w, h = int(np.rint(W * s)), int(np.rint(H * s))
    assert out.shape == (h, w) and out.dtype == f32
    sc = 1.0 / s
    if abs(sc - round(sc)) < np.finfo(np.float64).eps:          # integer factor: the estimator pyramid's own rule (oracle/pm_oracle.cpp resizeArea), cut blocks averaged
        f = int(round(sc))
        for y in range(h):
            for x in range(w):
                blk = img[y * f:y * f + f, x * f:x * f + f]
                if blk.shape == (f, f) and f == 2:
                    want = ((blk[0, 0] + blk[0, 1]) + (blk[1, 0] + blk[1, 1])) * f32(0.25)
                else:
                    acc = f32(0)
                    for v in blk.reshape(-1):
                        acc = f32(acc + v)
                    want = f32(acc * f32(1.0 / (f * f))) if blk.shape == (f, f) else f32(acc / f32(blk.size))
                assert out[y, x] == want, (y, x)
    else:
        assert np.array_equal(out, _area_literal(img, w, h, sc))
    assert abs(float(out.mean()) - float(img.mean())) < 0.05


@pytest.mark.parametrize("shape,scale", [((13, 17), 1.25), ((11, 9), 1.5), ((8, 12), 2.0), ((10, 10), 1.2)])
def test_scale_image_enlarges_with_the_cubic_path(shape, scale):
    rng = np.random.default_rng(int(scale * 100))
    img = rng.random(shape).astype(f32)
    out = densify.scale_image(img, scale)
    s = float(f32(scale)); H, W = shape
    w, h = int(np.rint(W * s)), int(np.rint(H * s))
    assert out.shape == (h, w) and np.array_equal(out, _cubic_literal(img, w, h, 1.0 / s))
    ramp = np.tile(np.arange(W, dtype=f32), (H, 1))              # sanity: a ramp stays a ramp away from the clamped border (OpenCV's A = -0.75 kernel is not exact on it)
    r = densify.scale_image(ramp, scale)
    xs = (np.arange(w) + 0.5) / s - 0.5
    inner = (xs > 1) & (xs < W - 2)
    assert np.allclose(r[:, inner], np.tile(xs[inner], (h, 1)), atol=0.06)
    assert np.allclose(densify.scale_image(np.full(shape, 0.37, f32), scale), 0.37, atol=1e-6)


def test_need_scale_image_is_the_float_test():
    assert not densify.need_scale_image(1.0) and not densify.need_scale_image(1.1499) and not densify.need_scale_image(0.8501)
    assert densify.need_scale_image(1.15) == bool(abs(f32(1.15) - f32(1)) >= f32(0.15)) and densify.need_scale_image(0.85) == bool(abs(f32(0.85) - f32(1)) >= f32(0.15))
    assert densify.need_scale_image(0.8) and densify.need_scale_image(1.3) and densify.scale_image(np.zeros((4, 4), f32), 1.05) is None


class _Recorder:
    """Stands in for PatchMatchHIP: records the scene calls of densify.compute_depth_maps / scene_load."""

    def __init__(self, n, shapes):
        self.calls, self.shapes = [], shapes
        self.depth = {i: np.full(shapes[i], float(i + 1), f32) for i in range(n)}

    def __getattr__(self, name):
        def rec(*a, **k):
            self.calls.append((name,) + tuple(a))
            if name == "scene_get_maps":
                return self.depth[a[0]], None, None
        return rec


@pytest.fixture(scope="module")
def scaled_scene(tmp_path_factory):
    """The pipeline-test scene with stored view scores (archive version 7) in which image 0 sees image 1 at 0.8x and image 2 at 1.25x, and image 3 sees image 1 at 0.8x too."""
    py = mvsi.load(SCENE)
    cams = views.Cameras(py)
    for i, im in enumerate(py.images):
        ok, nb, pts, avg = views.select_neighbor_views(py, cams, i)
        assert ok
        nb = nb.copy()
        if i == 0:
            nb["scale"][nb["ID"] == 1] = 0.8; nb["scale"][nb["ID"] == 2] = 1.25
        if i == 3:
            nb["scale"][nb["ID"] == 1] = 0.8
        im.view_scores, im.avg_depth = nb, avg
    d = tmp_path_factory.mktemp("scaled")
    p = str(d / "scene7.mvs")
    mvsi.save(p, py, version=7)
    rng = np.random.default_rng(5)
    imgs = {im.name: rng.integers(0, 255, (120, 160, 3)).astype(np.uint8) for im in py.images}
    return p, (lambda path: imgs[os.path.relpath(path, str(d)).replace(os.sep, "/")] if os.path.relpath(path, str(d)).replace(os.sep, "/") in imgs else imgs[[k for k in imgs if path.endswith(k)][0]])


def test_load_scene_makes_resampled_copies_of_neighbours(scaled_scene):
    path, loader = scaled_scene
    sv = densify.load_scene(path, opt=views.DenseOptions(nResolutionLevel=0, nMinResolution=64), image_loader=loader)
    assert (sv.width, sv.height) == (160, 120) and sv.n_views == 4 + 2                      # image 1 at 0.8x is needed twice and made once
    a08 = [a for a, j in sv.alias_of.items() if j == 1]; a125 = [a for a, j in sv.alias_of.items() if j == 2]
    assert len(a08) == 1 and len(a125) == 1 and sorted(sv.alias_of) == [4, 5]
    assert sv.sizes[a08[0]] == (128, 96) and sv.gray[a08[0]].shape == (96, 128) and sv.sizes[a125[0]] == (200, 150) and sv.gray[a125[0]].shape == (150, 200)
    assert np.array_equal(sv.gray[a08[0]], densify.scale_image(sv.gray[1], 0.8)) and np.array_equal(sv.gray[a125[0]], densify.scale_image(sv.gray[2], 1.25))
    sc = mvsi.load(path)
    for a, (w, h) in ((a08[0], (128, 96)), (a125[0], (200, 150))):
        K, R, C, _, _ = sc.camera(sv.alias_of[a], (w, h))                                  # Image::GetCamera(platforms, image.size())
        assert np.array_equal(sv.K[a], K) and np.array_equal(sv.R[a], R) and np.array_equal(sv.C[a], C)
        assert len(sv.neighbors[a]) == 0 and len(sv.estimate_neighbors[a]) == 0 and a not in sv.ids
        assert abs(sv.K[a][0, 0] / sv.K[sv.alias_of[a]][0, 0] - w / 160) < 1e-12
    e0 = list(sv.estimate_neighbors[0]); n0 = list(sv.neighbors[0])
    assert [sv.alias_of.get(s, s) for s in e0] == n0 and a08[0] in e0 and a125[0] in e0 and 1 not in e0 and 2 not in e0
    assert a08[0] in list(sv.estimate_neighbors[3]) and list(sv.estimate_neighbors[1]) == list(sv.neighbors[1]) and list(sv.estimate_neighbors[2]) == list(sv.neighbors[2])
    # the engine is loaded with the estimation lists; the copies as sized source-only views
    from openmvs_amd.patchmatch import PatchMatchHIP
    rec = _Recorder(6, {i: (sv.sizes[i][1], sv.sizes[i][0]) for i in range(6)})
    PatchMatchHIP.scene_load(rec, sv, n_levels=2)
    kinds = [c[0] for c in rec.calls]
    assert kinds == ["scene_create"] + ["scene_set_view"] * 4 + ["scene_set_view_sized"] * 2 and rec.calls[0][1:] == (6, 160, 120, 2)
    assert list(rec.calls[1][-1]) == e0 and rec.calls[5][1] == 4 and rec.calls[5][2].shape == sv.gray[4].shape
    # the driver: after every commit the copies receive the depth map of their image with its own camera; the filter sees the images themselves
    rec.calls.clear()
    from openmvs_amd.patchmatch import PMHipParams
    p = PMHipParams(); p.nEstimationGeometricIters = 2
    densify.compute_depth_maps(rec, sv.ids, p, scene=sv)
    seq = [(c[0], c[1] if len(c) > 1 and not isinstance(c[1], (list, np.ndarray)) else None) for c in rec.calls]
    names = [s[0] for s in seq]
    i_commit = [k for k, n in enumerate(names) if n == "scene_commit_round"]
    assert len(i_commit) == 2
    for k in i_commit:
        after = rec.calls[k + 1:k + 1 + 2 * len(sv.alias_of)]
        assert [c[0] for c in after] == ["scene_get_maps", "scene_set_source_depth"] * len(sv.alias_of)
        for g, s_ in zip(after[0::2], after[1::2]):
            a, j = s_[1], g[1]
            assert sv.alias_of[a] == j and np.array_equal(s_[2], rec.depth[j]) and np.array_equal(s_[3], sv.K[j]) and np.array_equal(s_[4], sv.R[j]) and np.array_equal(s_[5], sv.C[j])
        assert names[k + 1 + 2 * len(sv.alias_of)] == "Init"
    i_filter = names.index("scene_filter")
    back = [c for c in rec.calls[:i_filter] if c[0] == "scene_set_view"]
    assert sorted(c[1] for c in back) == [0, 3] and all(c[2] is None for c in back)
    assert list(back[0][-1]) == list(sv.neighbors[back[0][1]]) and names.index("scene_set_view") > i_commit[-1]
    # without copies nothing changes for the driver
    rec.calls.clear()
    densify.compute_depth_maps(rec, sv.ids, p)
    assert "scene_set_source_depth" not in [c[0] for c in rec.calls] and "scene_set_view" not in [c[0] for c in rec.calls]


def test_load_scene_takes_images_of_different_sizes(scaled_scene):
    path, loader = scaled_scene
    py = mvsi.load(SCENE)
    small = py.images[2].name

    def mixed(p):
        img = loader(p)
        return img[:90, :120] if p.endswith(small) else img          # image 2 is 120x90, the others 160x120

    sv = densify.load_scene(SCENE, opt=views.DenseOptions(nResolutionLevel=0, nMinResolution=64), image_loader=mixed)
    assert (sv.width, sv.height) == (160, 120) and sv.sizes[2] == (120, 90) and sv.sizes[0] == (160, 120)
    assert sv.gray[2].shape == (90, 120) and sv.init_depth[2].shape == (90, 120) and sv.init_depth[0].shape == (120, 160)
    # the footprints now differ by the resolution ratio, so the reference's rule resamples across it: image 2 is seen enlarged by ~4/3 by the others, and sees them at ~3/4
    assert sv.alias_of and set(sv.alias_of.values()) >= {2}
    for a, j in sv.alias_of.items():
        scales = [float(v["scale"]) for i in sv.ids for v in sv.view_scores[i] if int(v["ID"]) == j and densify.need_scale_image(v["scale"])]
        W, H = sv.sizes[j]
        assert sv.sizes[a] in [(int(np.rint(W * float(f32(s)))), int(np.rint(H * float(f32(s))))) for s in scales]
        assert (j == 2) == (sv.sizes[a][0] > sv.sizes[j][0])                                  # image 2 only ever grows, the others only shrink
    from openmvs_amd.patchmatch import PatchMatchHIP
    n = sv.n_views
    rec = _Recorder(n, {i: (sv.sizes[i][1], sv.sizes[i][0]) for i in range(n)})
    PatchMatchHIP.scene_load(rec, sv)
    assert [c[0] for c in rec.calls][:5] == ["scene_create", "scene_set_view", "scene_set_view", "scene_set_view_sized", "scene_set_view"]
    assert all((c[0] == "scene_set_view_sized") == (tuple(sv.sizes[c[1]]) != (160, 120)) for c in rec.calls[1:])


def test_checkpoint_files_and_resume(tmp_path):
    """The reference's file contract through densify.compute_depth_maps(dmap_dir=): a view whose depthNNNN.dmap exists is read back instead of estimated in the photometric
    pass (SceneDensify.cpp:2010-2029), every estimated map is written (atomically) after the pass / each geometric round / the filters."""
    from openmvs_amd import dmap, synth
    from openmvs_amd.patchmatch import PMHipParams
    sc = synth.make_scene(4, 32, 24, n_src=2)
    sc.sizes = [(32, 24)] * 4
    rec = _Recorder(4, {i: (24, 32) for i in range(4)})
    full = rec.scene_get_maps                                         # depth from the recorder; give it normals and confidences too
    rec.__dict__["scene_get_maps"] = lambda v: (rec.depth[v], np.zeros((24, 32, 3), f32), np.full((24, 32), 0.5, f32))
    d = str(tmp_path / "dmaps")
    p = PMHipParams(); p.nEstimationGeometricIters = 1
    assert densify.compute_depth_maps(rec, [0, 1, 2, 3], p, scene=sc, dmap_dir=d) == []
    assert sorted(os.listdir(d)) == ["depth%04d.dmap" % i for i in range(4)]              # no .tmp left behind
    got = dmap.load(os.path.join(d, "depth0002.dmap"))
    assert np.array_equal(got["depth_map"], rec.depth[2]) and got["reference_view_id"] == 2 and got["neighbor_view_ids"] == [int(i) for i in sc.neighbors[2]]
    kinds = [c[0] for c in rec.calls]
    assert kinds.count("scene_estimate") == 2 and list(rec.calls[kinds.index("scene_estimate")][1]) == [0, 1, 2, 3]
    # second run: views 1 and 3 still have their files -> read back, only 0 and 2 go through the photometric pass; the geometric round takes all four
    os.remove(os.path.join(d, "depth0000.dmap")); os.remove(os.path.join(d, "depth0002.dmap"))
    rec.calls.clear()
    assert densify.compute_depth_maps(rec, [0, 1, 2, 3], p, scene=sc, dmap_dir=d) == [1, 3]
    est = [c for c in rec.calls if c[0] == "scene_estimate"]
    assert list(est[0][1]) == [0, 2] and est[0][2] == -1 and list(est[1][1]) == [0, 1, 2, 3] and est[1][2] == 0
    loaded = [c for c in rec.calls if c[0] in ("scene_set_maps", "scene_set_conf")]
    assert [(c[0], c[1]) for c in loaded] == [("scene_set_maps", 1), ("scene_set_conf", 1), ("scene_set_maps", 3), ("scene_set_conf", 3)]
    assert np.array_equal(loaded[0][2], rec.depth[1]) and np.array_equal(loaded[1][2], np.full((24, 32), 0.5, f32))
    assert sorted(os.listdir(d)) == ["depth%04d.dmap" % i for i in range(4)]
    # a checkpoint written without normals: they are estimated from its depths, as InitViews does (EstimateNormalMap)
    ys, xs = np.mgrid[0:24, 0:32].astype(f32)
    plane = (3 + 0.01 * xs + 0.02 * ys).astype(f32)
    dmap.save(os.path.join(d, "depth0003.dmap"), "x.jpg", [3, 0], (32, 24), sc.K[3], sc.R[3], sc.C[3], 1.0, 9.0, plane, None, np.ones((24, 32), f32))
    rec.calls.clear()
    assert 3 in densify.compute_depth_maps(rec, [0, 1, 2, 3], p, scene=sc, dmap_dir=d)
    got3 = [c for c in rec.calls if c[0] == "scene_set_maps" and c[1] == 3][0]
    assert np.array_equal(got3[2], plane) and np.array_equal(got3[3], views.estimate_normal_map(sc.K[3], plane)) and (np.linalg.norm(got3[3][2:-2, 2:-2], axis=-1) > 0.999).all()
    # a file of another size is not this view's checkpoint
    dmap.save(os.path.join(d, "depth0001.dmap"), "x.jpg", [1, 0], (16, 12), sc.K[1], sc.R[1], sc.C[1], 1.0, 2.0, np.ones((12, 16), f32), None, np.ones((12, 16), f32))
    with pytest.raises(ValueError):
        densify.compute_depth_maps(rec, [0, 1, 2, 3], p, scene=sc, dmap_dir=d)
    with pytest.raises(ValueError):
        densify.compute_depth_maps(rec, [0], p, dmap_dir=d)


def test_fusion_order_and_the_dense_scene_archive(tmp_path):
    """The output side: FuseDepthMaps' processing order, the option table's fusion values, and `<scene>_dense.mvs` (Scene::SaveInterface) read back by both archive readers."""
    from openmvs_amd import mvsfront, optdense
    sv = densify.load_scene(SCENE, opt=views.DenseOptions(nResolutionLevel=1, nMinResolution=64), image_loader=lambda p: np.zeros((479, 640, 3), np.uint8))
    assert sorted(sv.all_view_scores) == [0, 1, 2, 3] and all(len(sv.all_view_scores[i]) >= len(sv.neighbors[i]) for i in range(4)) and all(sv.avg_depth[i] > 0 for i in range(4))
    assert densify.fuse_order(sv) == sorted(range(4), key=lambda i: (-len(sv.all_view_scores[i]), i))
    sv.all_view_scores[2] = sv.all_view_scores[2][:1]; sv.all_view_scores[1] = sv.all_view_scores[1][:0]
    assert densify.fuse_order(sv) == [0, 3, 2]                                            # by the size of the whole list; an image without neighbours is dropped
    rec = _Recorder(4, {i: (240, 320) for i in range(4)})
    opt = optdense.defaults(); opt.nMinViewsFuse = 3; opt.nEstimateNormals = 2; opt.fNormalDiffThreshold = 20.0
    densify.fuse_depth_maps(rec, sv, opt, bgr={i: np.zeros((240, 320, 3), np.uint8) for i in range(4)})
    assert [c[0] for c in rec.calls] == ["scene_set_color"] * 4 + ["scene_fuse"] and rec.calls[-1][1:] == ([0, 3, 2], 3, float(f32(0.01)), 20.0, True, True)
    rec.calls.clear(); opt.nEstimateColors = 1                                            # "final": colours are not estimated during the fusion (SceneDensify.cpp:1700,1733)
    densify.fuse_depth_maps(rec, sv, opt, bgr={i: np.zeros((240, 320, 3), np.uint8) for i in range(4)})
    assert [c[0] for c in rec.calls] == ["scene_fuse"] and rec.calls[-1][5:] == (False, True)
    # a small cloud into the archive
    cloud = dict(nPoints=3, points=np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], f32), viewStart=np.array([0, 2, 5, 7], np.uint32), views=np.array([0, 1, 0, 2, 3, 1, 3], np.uint32),
                 weights=np.array([.5, .25, 1, 2, 3, .125, 4], f32), normals=np.array([[0, 0, -1], [0, -1, 0], [-1, 0, 0]], f32), colors=np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], np.uint8))
    out = str(tmp_path / "scene_dense.mvs")
    densify.save_dense_scene(SCENE, out, cloud, sv, version=7)
    back = mvsi.load(out)
    assert back.version == 7 and np.array_equal(back.vertices, cloud["points"]) and np.array_equal(back.vertices_normal, cloud["normals"]) and np.array_equal(back.vertices_color, cloud["colors"])
    assert list(back.views_of(1)["image_id"]) == [0, 2, 3] and list(back.views_of(1)["confidence"]) == [1.0, 2.0, 3.0] and list(back.views_of(2)["image_id"]) == [1, 3]
    src = mvsi.load(SCENE)
    assert len(back.images) == 4 and [im.name for im in back.images] == [im.name for im in src.images] and np.array_equal(back.platforms[0].cameras[0].K, src.platforms[0].cameras[0].K)
    assert back.images[0].view_scores.tobytes() == sv.all_view_scores[0].tobytes() and len(back.images[1].view_scores) == 0 and back.images[3].avg_depth == f32(sv.avg_depth[3])
    cf = mvsfront.SceneFront(out)                                                         # the C++ reader: same points, and the stored lists are the images' neighbour lists now
    assert cf.n_points == 3 and np.array_equal(cf.point(1)[0], cloud["points"][1]) and list(cf.point(1)[1]) == [0, 2, 3]
    assert cf.neighbors(0).tobytes() == sv.all_view_scores[0].tobytes() and cf.image_depths(3)[1] == f32(sv.avg_depth[3])
    densify.save_dense_scene(SCENE, out, dict(cloud, normals=None, colors=None))        # the archive's own version (6): no view scores, no normals / colours
    b6 = mvsi.load(out)
    assert b6.version == src.version and len(b6.vertices) == 3 and len(b6.vertices_normal) == 0 and len(b6.vertices_color) == 0

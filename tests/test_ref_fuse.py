"""Pins oracle/fuse_oracle.cpp to the REFERENCE'S OWN TEXT: DepthMapsData::FuseDepthMaps and MergeDepthMaps (libs/MVS/SceneDensify.cpp:1303-1646) with Conf2Weight,
cut verbatim and compiled with the reference's TPixel, camera members and container semantics (oracle/ref/ref_fuse_harness.cpp -> oracle/_ref/libref_fuse.so).
Point order, view lists, weights, positions, colours and normals must be identical bit for bit; the oracle is given the processing order the reference's own std::sort
produced (ties among equally connected images are the standard library's choice).  The device fusion (csrc/pm_fuse.hip) is compared with this oracle in tests/test_gpu_fuse.py."""
import numpy as np
import pytest

from openmvs_amd import synth
from oracle import pyoracle as po
from oracle import pyref as pr
from tests import fuse_cases as fc

pytestmark = pytest.mark.skipif(not pr.fuse_available(), reason="oracle/_ref/libref_fuse.so not built (needs /root/reference)")


def _same(a, b, what):
    assert a["nPoints"] == b["nPoints"], "%s: %d vs %d points" % (what, a["nPoints"], b["nPoints"])
    for k in ("viewStart", "views"):
        assert np.array_equal(a[k], b[k]), "%s: %s differs" % (what, k)
    for k in ("points", "weights", "normals"):
        if a[k] is None or b[k] is None:
            assert a[k] is None and b[k] is None, "%s: %s presence" % (what, k)
            continue
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), "%s: %s differs in %d values" % (what, k, int((a[k].view(np.uint32) != b[k].view(np.uint32)).sum()))
    if a["colors"] is None or b["colors"] is None:
        assert a["colors"] is None and b["colors"] is None, what + ": colour presence"
    else:
        assert np.array_equal(a["colors"], b["colors"]), what + ": colours differ"


@pytest.fixture(scope="module")
def scenes():
    return synth.make_scene(5, 160, 120, n_src=4), synth.make_scene(9, 128, 96, n_src=8)


@pytest.mark.parametrize("which,seed", [(0, 0), (0, 3), (1, 1)])
@pytest.mark.parametrize("opts", [dict(), dict(nMinViewsFuse=3, fDepthDiffThreshold=0.02, fNormalDiffThreshold=15.0), dict(bEstimateColor=False, bEstimateNormal=False),
                                  dict(nMinViewsFuse=100)])
def test_fuse_is_the_reference_function(scenes, which, seed, opts):
    sc = scenes[which]
    deps, nrms, cnfs = fc.make_maps(sc, seed)
    nbs = [list(sc.neighbors[v]) for v in range(sc.n_views)]
    ref, order = pr.ref_fuse_depth_maps(deps, nrms, cnfs, sc.bgr, sc.K, sc.R, sc.C, nbs, **opts)
    orc = po.fuse_depth_maps(deps, nrms, cnfs, sc.bgr, sc.K, sc.R, sc.C, nbs, order=order, **opts)
    _same(ref, orc, "fuse %s" % opts)
    assert ref["nPoints"] > 1000 or opts.get("nMinViewsFuse", 2) > 5


def test_fuse_with_missing_maps_confidences_and_uneven_neighbour_lists(scenes):
    sc = scenes[1]
    deps, nrms, cnfs = fc.make_maps(sc, 5)
    deps = list(deps); deps[3] = None                                       # an image without a depth map (DepthData::IsValid false)
    r = np.random.RandomState(2)
    nbs = [list(sc.neighbors[v])[:r.randint(2, 9)] for v in range(sc.n_views)]   # different connection scores: the reference's sort decides the order
    ref, order = pr.ref_fuse_depth_maps(deps, nrms, None, sc.bgr, sc.K, sc.R, sc.C, nbs)
    assert 3 not in order and len(order) == sc.n_views - 1
    orc = po.fuse_depth_maps(deps, nrms, None, sc.bgr, sc.K, sc.R, sc.C, nbs, order=order)
    _same(ref, orc, "missing map, no confidences")
    # the order itself: decreasing neighbour count
    counts = [len(nbs[i]) for i in order]
    assert counts == sorted(counts, reverse=True)


def test_merge_is_the_reference_function(scenes):
    sc = scenes[0]
    deps, nrms, cnfs = fc.make_maps(sc, 7)
    nbs = [list(sc.neighbors[v]) for v in range(sc.n_views)]
    for kw in (dict(), dict(bEstimateColor=False), dict(bEstimateNormal=False)):
        ref, _ = pr.ref_fuse_depth_maps(deps, nrms, cnfs, sc.bgr, sc.K, sc.R, sc.C, nbs, nMinViewsFuse=1, **kw)
        orc = po.fuse_depth_maps(deps, nrms, cnfs, sc.bgr, sc.K, sc.R, sc.C, nbs, nMinViewsFuse=1, **kw)
        ref = dict(ref); orc = dict(orc); ref["weights"] = None; orc["weights"] = None      # MergeDepthMaps stores no weights
        _same(ref, orc, "merge %s" % kw)


@pytest.mark.parametrize("seed,opts", [(0, dict()), (1, dict(nMinViewsFuse=3, fDepthDiffThreshold=0.02)), (0, dict(nMinViewsFuse=1))])
def test_fuse_of_depth_maps_of_different_sizes_is_the_reference_function(seed, opts):
    """Every depth map of its own size (DepthMapsData::InitViews sizes each DepthData on its image): projections are tested against the neighbour's own map (:1548) and
    pixel indices run over the view's own width."""
    deps, nrms, cnfs, bgrs, K, R, Cc, nbs = fc.make_mixed(seed)
    ref, order = pr.ref_fuse_depth_maps(deps, nrms, cnfs, bgrs, K, R, Cc, nbs, **opts)
    orc = po.fuse_depth_maps(deps, nrms, cnfs, bgrs, K, R, Cc, nbs, order=order, **opts)
    if opts.get("nMinViewsFuse", 2) < 2:
        ref = dict(ref); orc = dict(orc); ref["weights"] = None; orc["weights"] = None
    _same(ref, orc, "mixed sizes %s" % opts)
    assert ref["nPoints"] > 1000


# ---- MVS::EstimateNormalMap (DepthMap.cpp:1522-1613, verbatim) against openmvs_amd.views.estimate_normal_map ---------------------------------------------------
@pytest.mark.parametrize("case", ["plane", "surface", "holes", "steps", "tiny", "garbage"])
def test_estimate_normal_map_is_the_reference_function(case):
    """The normals the reference gives a depth map that has none (FuseDepthMaps :1427, InitViews :413, the SGM fuse mode :2055): bit for bit."""
    from openmvs_amd import views
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    w, h = (67, 45) if case != "tiny" else (3, 2)
    K = np.array([[80.5, 0, (w - 1) / 2 + 0.3], [0, 79.25, (h - 1) / 2 - 0.2], [0, 0, 1]])
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    if case == "plane":
        d = (5 + 0.01 * xs + 0.02 * ys).astype(np.float32)
    elif case == "steps":
        d = (4 + (xs > w / 2) * 0.5 + 0.003 * ys).astype(np.float32)                        # a depth step of more than 3 %: neighbours across it do not count
    elif case == "garbage":
        d = rng.uniform(-1, 9, (h, w)).astype(np.float32)                                    # negative depths, nothing similar to anything
    else:
        d = (6 + 0.4 * np.sin(xs / 7) * np.cos(ys / 5) + rng.normal(0, 0.002, (h, w))).astype(np.float32)
    if case == "holes":
        d[rng.random((h, w)) < 0.35] = 0
        d[10:14] = 0
    from openmvs_amd import mvsfront
    got = views.estimate_normal_map(K, d)
    want = pr.ref_estimate_normal_map(K, d)
    assert np.array_equal(mvsfront.estimate_normal_map(K, d).view(np.uint32), want.view(np.uint32)), "C++ front end"
    assert got.dtype == np.float32 and got.shape == (h, w, 3)
    bad = np.flatnonzero(got.view(np.uint32).ravel() != want.view(np.uint32).ravel())
    assert bad.size == 0, (case, bad[:5], got.ravel()[bad[:5]], want.ravel()[bad[:5]])
    if case in ("plane", "surface"):
        nz = np.linalg.norm(got, axis=-1)
        assert (nz[2:-2, 2:-2] > 0.999).all() and (got[..., 2][nz > 0] < 0).all()              # unit normals facing the camera
    if case == "garbage":
        assert (np.linalg.norm(got, axis=-1) == 0).mean() > 0.9


@pytest.mark.parametrize("seed", [2, 5, 11])
def test_dense_initialisation_rasteriser_is_the_reference_code(seed):
    """TImage::RasterizeTriangleBary + EdgeFunction + the perspective-correct barycentric coordinates + TRasterMeshBase + the RasterDepth functor of
    TriangulatePoints2DepthMap (all verbatim in libref_fuse.so) against views._raster_face over a random Delaunay mesh that overhangs the image: depth and normal maps bit
    for bit.  (mvs_front.cpp's rasterFace is compared with the numpy one on whole dense maps in tests/test_mvsfront.py.)"""
    from scipy.spatial import Delaunay
    from openmvs_amd import views
    rng = np.random.default_rng(seed)
    w, h, nv = 64, 48, 30
    proj = (rng.random((nv, 2)) * [w + 10, h + 10] - 5).astype(np.float32)
    z = (rng.random(nv) * 3 + 2).astype(np.float32)
    nrm = rng.normal(size=(nv, 3)); nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    faces = []
    for f in Delaunay(proj.astype(np.float64)).simplices.astype(np.uint32):
        a, b, c = proj[f].astype(np.float64)
        faces.append(f if (c[0] - a[0]) * (b[1] - a[1]) - (c[1] - a[1]) * (b[0] - a[0]) > 0 else f[[0, 2, 1]])      # front-facing for EdgeFunction (back faces are culled)
    faces = np.array(faces, np.uint32)
    rd, rn = pr.ref_raster_faces(w, h, proj, z, nrm, faces)
    d = np.zeros((h, w), np.float32); n = np.zeros((h, w, 3), np.float32)
    for fa in faces:
        views._raster_face(proj[fa], z[fa], nrm[fa], d, n)
    assert (rd > 0).mean() > 0.5
    assert np.array_equal(rd.view(np.uint32), d.view(np.uint32)) and np.array_equal(rn.view(np.uint32), n.view(np.uint32))
    d2 = np.zeros((h, w), np.float32)                                            # the depth-only variant (DepthMap.cpp:1229-1247): the same depths
    for fa in faces:
        views._raster_face(proj[fa], z[fa], None, d2, None)
    assert np.array_equal(d2, pr.ref_raster_faces(w, h, proj, z, None, faces)[0]) and np.array_equal(d2, d)
    rd_back = pr.ref_raster_faces(w, h, proj, z, nrm, faces[:, [0, 2, 1]])[0]    # back-facing triangles are culled
    assert not rd_back.any()


def test_untrusted_point_initialisation_is_the_reference_code():
    """`Min Views Trust Point = 1`: InitViews splats the sparse points' depths on 5x5 blocks with zero normals and takes the depth range from them (SceneDensify.cpp:418-451,
    verbatim) -- against mvsf_init_depth_map and views.init_depth_map on the pipeline-test scene at two working resolutions, and with no points at all."""
    import os
    from openmvs_amd import mvsfront, mvsi, views
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "scene", "scene.mvs")
    cf, py = mvsfront.SceneFront(path), mvsi.load(path)
    opt = views.DenseOptions(nMinViewsTrustPoint=1)
    for size in ((640, 479), (320, 240)):
        cams = views.Cameras(py, [size] * 4)
        for i in (0, 3):
            ok, nb, pts, avg = views.select_neighbor_views(py, cams, i)
            want = pr.ref_init_views_splat(cams.K[i], cams.R[i], cams.C[i], size, py.vertices, pts)
            got_c = cf.init_depth_map(i, pts, size, nMinViewsTrustPoint=1)
            got_p = views.init_depth_map(py, cams, i, pts, opt)
            for name, got in (("C++", got_c), ("numpy", got_p)):
                assert np.array_equal(got[0], want[0]) and not got[1].any() and not want[1].any(), (name, size, i)
                assert np.float32(got[2]) == np.float32(want[2]) and np.float32(got[3]) == np.float32(want[3]), (name, size, i, got[2:], want[2:])
            assert (want[0] > 0).mean() > 0.02
    none = pr.ref_init_views_splat(cams.K[0], cams.R[0], cams.C[0], (320, 240), py.vertices, np.zeros(0, np.uint32))
    assert not none[0].any() and np.float32(none[2]) == np.float32(0.1) and none[3] == 100.0

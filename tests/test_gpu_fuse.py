"""-m gpu: FuseDepthMaps on the device (deterministic reservations, csrc/pm_fuse.hip) against the sequential oracle
(oracle/fuse_oracle.cpp).  Everything is compared exactly: point order, view lists, projections, weights, positions, colours, normals."""
import time

import numpy as np
import pytest

from openmvs_amd import synth
from openmvs_amd.patchmatch import PatchMatchHIP
from oracle import pyoracle as po
from tests import fuse_cases as fc

pytestmark = pytest.mark.gpu


def _load(e, sc, maps):
    e.scene_load(sc, n_levels=0)
    d, n, c = maps
    for v in range(sc.n_views):
        e.scene_set_maps(v, d[v], n[v]); e.scene_set_conf(v, c[v]); e.scene_set_color(v, sc.bgr[v])


def _order(sc):
    return po.fuse_order([len(x) for x in sc.neighbors])


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_device_fuse_is_the_sequential_fuse(small_scene, nine_scene, case):
    sc, seed, kw = [(small_scene, 1, {}), (small_scene, 2, dict(nMinViewsFuse=3)), (nine_scene, 3, {}),
                    (nine_scene, 4, dict(fDepthDiffThreshold=0.03, fNormalDiffThreshold=60.0, bEstimateColor=False, bEstimateNormal=False))][case]
    maps = fc.make_maps(sc, seed=seed)
    e = PatchMatchHIP(0)
    _load(e, sc, maps)
    got = e.scene_fuse(_order(sc), **kw)
    ref = po.fuse_depth_maps(*maps, list(sc.bgr), sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], **kw)
    fc.same_cloud(got, ref, f"case {case}")
    assert got["nPoints"] > 1000 and got["rounds"] < 40 * sc.n_views
    # the scene's own maps are untouched (fusion works on copies), and a second call gives the same cloud
    d0, _, _ = e.scene_get_maps(0)
    assert np.array_equal(d0, maps[0][0])
    fc.same_cloud(e.scene_fuse(_order(sc), **kw), ref, f"case {case} again")
    e.close()


def test_fuse_option_sweep(small_scene):
    """FuseDepthMaps with thresholds and view counts away from the defaults: more views than any point has, very tight and very loose depth / normal
    thresholds, colours without normals and the reverse."""
    sc = small_scene
    maps = fc.make_maps(sc, seed=5)
    e = PatchMatchHIP(0)
    _load(e, sc, maps)
    sets = [dict(nMinViewsFuse=4), dict(nMinViewsFuse=5), dict(fDepthDiffThreshold=0.001, fNormalDiffThreshold=5.0), dict(fDepthDiffThreshold=0.2, fNormalDiffThreshold=179.0),
            dict(bEstimateColor=True, bEstimateNormal=False), dict(bEstimateColor=False, bEstimateNormal=True), dict(nMinViewsFuse=2, fDepthDiffThreshold=0.05)]
    some = 0
    for kw in sets:
        got = e.scene_fuse(_order(sc), **kw)
        ref = po.fuse_depth_maps(*maps, list(sc.bgr), sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], **kw)
        fc.same_cloud(got, ref, str(kw))
        some += got["nPoints"]
    assert some > 1000
    e.close()


def test_device_fuse_reproduces_the_golden_cloud():
    """The committed fixture (tests/golden/fuse_golden_96x64.npz, written by the oracle): same inputs through the C ABI."""
    from tests.test_fuse import _golden_inputs
    g, z, d, n, c, bgr, nbrs = _golden_inputs()
    nv = int(g["n_views"])

    class S:      # the attributes scene_load reads
        n_views = nv; width = 96; height = 64; gray = g["gray"]; K = g["K"]; R = g["R"]; C = g["C"]; dmin = g["dmin"]; dmax = g["dmax"]; neighbors = g["neighbors"]
    e = PatchMatchHIP(0)
    e.scene_load(S, n_levels=0)
    for v in range(nv):
        e.scene_set_maps(v, d[v], n[v]); e.scene_set_conf(v, c[v]); e.scene_set_color(v, bgr[v])
    order = po.fuse_order([len(x) for x in nbrs])
    for tag, kw in (("fuse2", dict(nMinViewsFuse=2)), ("fuse3", dict(nMinViewsFuse=3, fNormalDiffThreshold=40.0))):
        r = e.scene_fuse(order, **kw)
        assert r["nDepths"] == int(z[tag + "_nDepths"])
        for k in ("points", "viewStart", "views", "weights", "projs", "colors", "normals"):
            assert np.array_equal(r[k], z[tag + "_" + k]), (tag, k)
    e.close()


def test_device_merge_mode(small_scene):
    """nMinViewsFuse < 2 routes to MergeDepthMaps (SceneDensify.cpp:1305-1368)."""
    sc = small_scene
    maps = fc.make_maps(sc, seed=6)
    e = PatchMatchHIP(0)
    _load(e, sc, maps)
    order = list(range(sc.n_views))
    got = e.scene_fuse(order, nMinViewsFuse=1)
    ref = po.fuse_depth_maps(*maps, list(sc.bgr), sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], order=order, nMinViewsFuse=1)
    fc.same_cloud(got, ref, "merge")
    assert got["nPoints"] == sum(int((d > 0).sum()) for d in maps[0])
    e.close()


def test_device_fuse_custom_order_and_errors(small_scene):
    sc = small_scene
    maps = fc.make_maps(sc, seed=5)
    e = PatchMatchHIP(0)
    e.scene_load(sc, n_levels=0)
    for v in range(sc.n_views):
        e.scene_set_maps(v, maps[0][v], maps[1][v]); e.scene_set_conf(v, maps[2][v])
    with pytest.raises(Exception):
        e.scene_fuse(_order(sc))                       # colours requested but never uploaded
    order = [4, 0, 2, 1]                               # view 3 is only ever a neighbour
    got = e.scene_fuse(order, bEstimateColor=False)
    ref = po.fuse_depth_maps(*maps, None, sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], order=order, bEstimateColor=False)
    fc.same_cloud(got, ref, "custom order")
    e.close()


def test_device_fuse_full_size_is_exact_and_timed():
    """9 views of 1920x1080 (BASELINE config 2's resolution): 15 M depths.  The sequential oracle needs ~10 s for this, so the
    comparison stays exact at full size; the guarantees of fusion are asserted as well and both times are printed."""
    sc = synth.make_scene(9, 1920, 1080, n_src=8, device="cuda")
    maps = fc.make_maps(sc, seed=7)
    e = PatchMatchHIP(0)
    _load(e, sc, maps)
    e.scene_fuse(_order(sc))                                  # warm-up: allocations
    t = time.time(); r = e.scene_fuse(_order(sc)); dt = time.time() - t
    t = time.time()
    ref = po.fuse_depth_maps(*maps, list(sc.bgr), sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors])
    dt_ref = time.time() - t
    fc.same_cloud(r, ref, "full size")
    nv = np.diff(r["viewStart"].astype(np.int64))
    assert r["nPoints"] > 1_000_000 and nv.min() >= 2 and nv.max() <= 9 and r["viewStart"][-1] == len(r["views"])
    key = r["views"].astype(np.int64) * (1 << 32) + r["projs"][:, 1].astype(np.int64) * 65536 + r["projs"][:, 0]
    assert len(np.unique(key)) == len(key)
    first = r["viewStart"][:-1]
    v0 = r["views"][first]; xy = r["projs"][first]
    z = np.einsum("ij,ij->i", r["points"].astype(np.float64) - sc.C[v0], sc.R[v0][:, 2])
    gt = sc.gt_depth[v0, xy[:, 1], xy[:, 0]]
    assert np.median(np.abs(z - gt) / gt) < 2e-3
    print(f"\nfuse 9x1080p: {r['nPoints']} points from {r['nDepths']} depths; device {dt*1e3:.0f} ms incl. download ({r['rounds']} rounds), "
          f"sequential oracle {dt_ref*1e3:.0f} ms")
    e.close()


def test_device_fuse_of_depth_maps_of_different_sizes(W=160, H=120):
    """A scene whose views -- and therefore their depth, normal, confidence maps and colour images -- have different sizes (DepthMapsData::InitViews sizes every DepthData
    on its own image, SceneDensify.cpp:306-459): the device fusion, the merge (nMinViewsFuse < 2) and a second call equal the sequential oracle, which equals the
    reference's own FuseDepthMaps on such scenes (tests/test_ref_fuse.py)."""
    from openmvs_amd import synth
    deps, nrms, cnfs, bgrs, K, R, Cc, nbs = fc.make_mixed(0, W, H)
    n = len(deps)
    base = synth.make_scene(n, W, H, n_src=4)
    e = PatchMatchHIP(0)
    e.scene_load(base, n_levels=0)
    for v in range(n):
        if deps[v].shape != (H, W):
            gray = np.zeros(deps[v].shape, np.float32)        # (fusion does not read the gray image)
            e.scene_set_view_sized(v, gray, K[v], R[v], Cc[v], float(base.dmin[v]), float(base.dmax[v]), nbs[v])
        e.scene_set_maps(v, deps[v], nrms[v]); e.scene_set_conf(v, cnfs[v]); e.scene_set_color(v, bgrs[v])
    order = po.fuse_order([len(x) for x in nbs])
    for kw in (dict(), dict(nMinViewsFuse=3, fDepthDiffThreshold=0.02), dict(bEstimateColor=False, bEstimateNormal=False), dict(nMinViewsFuse=1)):
        got = e.scene_fuse(order if kw.get("nMinViewsFuse", 2) >= 2 else list(range(n)), **kw)
        ref = po.fuse_depth_maps(deps, nrms, cnfs, bgrs, K, R, Cc, nbs, **kw)
        if kw.get("nMinViewsFuse", 2) < 2:
            got = dict(got); ref = dict(ref); got["weights"] = None; ref["weights"] = None
        fc.same_cloud(got, ref, "mixed sizes %s" % kw)
        assert got["nPoints"] > 1000
    d2, _, _ = e.scene_get_maps(2)
    assert d2.shape == deps[2].shape and np.array_equal(d2, deps[2])
    e.close()


def test_fuse_with_a_source_only_slot_that_has_no_colour(small_scene):
    """A scene that also holds a source-only slot -- a resampled copy of a neighbour the estimation read (ViewData::ScaleImage; densify.load_scene puts them behind the images) --
    fuses with colours on: the copy has its own size, no depth map and no colour image, is nobody's neighbour at fusion time and must neither be asked for a colour nor be read.
    The cloud is the one the images alone give."""
    sc = small_scene
    maps = fc.make_maps(sc, seed=7)
    ref = po.fuse_depth_maps(*maps, list(sc.bgr), sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors])
    e = PatchMatchHIP(0)
    n = sc.n_views
    e.scene_create(n + 1, sc.width, sc.height, 0)
    for v in range(n):
        e.scene_set_view(v, sc.gray[v], sc.K[v], sc.R[v], sc.C[v], float(sc.dmin[v]), float(sc.dmax[v]), sc.neighbors[v])
    w2, h2 = sc.width * 3 // 4, sc.height * 3 // 4
    K2 = np.array(sc.K[1], np.float64); K2[0] *= w2 / sc.width; K2[1] *= h2 / sc.height
    e.scene_set_view_sized(n, np.ascontiguousarray(sc.gray[1][:h2, :w2]), K2, sc.R[1], sc.C[1], float(sc.dmin[1]), float(sc.dmax[1]), np.zeros(0, np.int32))
    d, nm, c = maps
    for v in range(n):
        e.scene_set_maps(v, d[v], nm[v]); e.scene_set_conf(v, c[v]); e.scene_set_color(v, sc.bgr[v])
    got = e.scene_fuse(_order(sc))
    fc.same_cloud(got, ref, "source-only slot without colour")
    assert got["nPoints"] > 1000 and got["colors"] is not None
    e.close()

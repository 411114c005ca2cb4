"""-m gpu: parity of the HIP engine (through the C ABI) with the CPU oracle.  The bar is
bit-exact: both sides run the same IEEE operation sequences (csrc/pm_math.h, -ffp-contract=off),
so depth RMSE vs the oracle is 0, well inside north_star's 1e-4 * scene-diameter tolerance; the
tolerance assert is kept next to the exact one so a future relaxation is explicit."""
import os

import numpy as np
import pytest

from openmvs_amd import synth
from openmvs_amd.patchmatch import default_params
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4  # x scene diameter (BASELINE.json north_star)


def _same(a, b, what):
    bad = np.flatnonzero(a.ravel().view(np.uint32) != b.ravel().view(np.uint32))
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} values differ, first at {bad[:5]}: {a.ravel()[bad[:5]]} vs {b.ravel()[bad[:5]]}"


def test_device_math_is_bitwise_identical_to_host():
    from openmvs_amd.patchmatch import PatchMatchHIP
    e = PatchMatchHIP(0)
    r = np.random.RandomState(0)
    n = 1 << 20
    cases = [(0, (-90 * r.rand(n)).astype(np.float32), None), (1, (2 * r.rand(n) - 1).astype(np.float32), None),
             (2, r.randn(n).astype(np.float32), r.randn(n).astype(np.float32)),
             (3, (8 * (2 * r.rand(n) - 1)).astype(np.float32), None), (4, (8 * (2 * r.rand(n) - 1)).astype(np.float32), None),
             (5, (100 * r.rand(n)).astype(np.float32), None), (6, r.randn(n).astype(np.float32), (r.randn(n) + 3).astype(np.float32)),
             (7, (3 * r.randn(n)).astype(np.float32), (3 * r.randn(n)).astype(np.float32)),
             # pm_div2 (shared-reciprocal correctly rounded division) vs '/': pixel-like magnitudes, tiny and huge operands
             (8, (4096 * r.rand(n)).astype(np.float32), (0.5 + r.rand(n)).astype(np.float32)),
             (8, (r.randn(n) * 10.0 ** r.randint(-30, 30, n)).astype(np.float32), (r.randn(n) * 10.0 ** r.randint(-14, 14, n)).astype(np.float32)),
             (9, (2000 * r.rand(n)).astype(np.float32), (0.01 + r.rand(n)).astype(np.float32))]
    for kind, a, b in cases:
        _same(e.math_eval(kind, a, b), po.math_eval(kind, a, b), f"pm_math kind {kind}")
    e.close()


def test_device_resampling_matches_oracle(engine):
    r = np.random.RandomState(1)
    img = r.rand(48, 64).astype(np.float32)
    _same(engine.resize(0, img, 2), po.resize_area(img, 2), "area 2")
    _same(engine.resize(0, img, 4), po.resize_area(img, 4), "area 4")
    _same(engine.resize(1, img), po.resize_linear(img, 128, 96), "linear x2")
    _same(engine.resize(2, img), po.resize_nearest(img, 128, 96), "nearest x2")


def _oracle(sc, v, seed, geo_iter=-1, depth=None, normal=None, src=None, **kw):
    ids = [v] + list(sc.neighbors[v])
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps=src)
    opt = po.default_opt(seed=seed, viewID=v, **kw)
    return po.estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), opt, geo_iter=geo_iter, depth=depth, normal=normal)


@pytest.mark.parametrize("levels", [0, 2])
def test_single_view_photometric_parity_N4(engine, small_scene, levels):
    sc = small_scene
    engine.Init(False)
    p = default_params(seed=11, nSubResolutionLevels=levels)
    ids = [1] + list(sc.neighbors[1])
    d, n, c = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[1], sc.dmax[1], params=p)
    od, on, oc = _oracle(sc, 1, 11, nSubResolutionLevels=levels)
    _same(d, od, "depth"); _same(n, on, "normal"); _same(c, oc, "conf")
    m = (d > 0) & (od > 0)
    assert np.sqrt(np.mean((d[m] - od[m]) ** 2)) <= TOL * sc.diameter
    assert m.mean() > 0.7


def test_single_view_parity_N8_and_N1(engine, nine_scene):
    sc = nine_scene
    engine.Init(False)
    for nsrc in (8, 1, 2, 3):
        p = default_params(seed=5)
        ids = [4] + list(sc.neighbors[4][:nsrc])
        d, n, c = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[4], sc.dmax[4], params=p)
        views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
        od, on, oc = po.estimate_depth_map(views, len(ids), float(sc.dmin[4]), float(sc.dmax[4]), po.default_opt(seed=5, viewID=4))
        _same(d, od, f"depth N={nsrc}"); _same(n, on, f"normal N={nsrc}"); _same(c, oc, f"conf N={nsrc}")


@pytest.mark.parametrize("lanes", [4, 8])
def test_views_per_lane_mappings_parity(nine_scene, small_scene, lanes):
    """The sweep kernel's work decompositions (PMHipTuning::sweepLanes): 4 lanes per pixel with 2 or 4 source views per lane (16 pixels per wavefront), or 8 lanes with 1 or 2,
    instead of one view per lane, and the mappings 8, 3-4 and 9-16 sources fall to; MINMEAN is order-free, so the maps are the same bits.  Photometric pass over the
    pyramid and a geometric round."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    e = PatchMatchHIP(0)
    e.tuning(wideMaxViews=-1, sweepLanes=lanes)
    test_single_view_parity_N8_and_N1(e, nine_scene)                 # 8 sources: (4,2) / (8,1); 1-3 sources: a quad of lanes
    test_single_view_photometric_parity_N4(e, small_scene, 2)        # 4 sources: (4,1)
    sc = nine_scene
    p = default_params(seed=5, nEstimationGeometricIters=1)
    e.Init(False); e.scene_load(sc, n_levels=2)
    allv = list(range(sc.n_views))
    e.scene_estimate(allv, -1, p)
    photo = [e.scene_get_maps(v) for v in allv]
    e.scene_commit_round(); e.Init(True)
    e.scene_estimate([4, 0], 0, p)
    for v in (4, 0):
        od, on, oc = _oracle(sc, v, 5, nEstimationGeometricIters=1)
        _same(photo[v][0], od, f"lanes {lanes}: photometric depth v{v}")
        gd, gn, gc = _oracle(sc, v, 5, geo_iter=0, depth=od, normal=on, src={u: photo[u][0] for u in allv}, nEstimationGeometricIters=1)
        d, n, c = e.scene_get_maps(v)
        _same(d, gd, f"lanes {lanes}: geometric depth v{v}"); _same(n, gn, "normal"); _same(c, gc, "conf")
    test_many_source_views_parity(e)                                 # 9 .. 16 sources: (4,4) / (8,2)
    e.close()


SWEEP_VARIANTS = {
    # how the sweep kernels' tap rows address the quad images (PMHipTuning::quadBuffer); both must give the oracle's bits
    "quad_buffer": 1,       # the default, named: 32-bit entry index into the level's buffer (buffer_load ... idxen, hardware range check)
    "quad_pointer": 2,      # through each view's own pointer with clamped coordinates (what a batch with source views of their own image size falls to)
}


@pytest.mark.parametrize("variant", sorted(SWEEP_VARIANTS))
def test_sweep_kernel_variants_parity(nine_scene, small_scene, variant, quick=False):
    """Both addressing modes of the sweep kernel against the oracle: 8 / 1 / 2 / 3 sources, the pyramid with 4 sources, and a scene batch with a geometric round."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    e = PatchMatchHIP(0)
    e.tuning(wideMaxViews=-1, quadBuffer=SWEEP_VARIANTS[variant])
    test_single_view_parity_N8_and_N1(e, nine_scene)
    if not quick:
        test_single_view_photometric_parity_N4(e, small_scene, 2)
    sc = nine_scene
    p = default_params(seed=5, nEstimationGeometricIters=1)
    e.Init(False); e.scene_load(sc, n_levels=2)
    allv = list(range(sc.n_views))
    e.scene_estimate(allv, -1, p)
    photo = [e.scene_get_maps(v) for v in allv]
    e.scene_commit_round(); e.Init(True)
    e.scene_estimate([4, 0], 0, p)
    for v in (4, 0):
        od, on, oc = _oracle(sc, v, 5, nEstimationGeometricIters=1)
        _same(photo[v][0], od, f"{variant}: photometric depth v{v}")
        gd, gn, gc = _oracle(sc, v, 5, geo_iter=0, depth=od, normal=on, src={u: photo[u][0] for u in allv}, nEstimationGeometricIters=1)
        d, n, c = e.scene_get_maps(v)
        _same(d, gd, f"{variant}: geometric depth v{v}"); _same(n, gn, "normal"); _same(c, gc, "conf")
    e.close()


def test_tuning_through_the_abi(nine_scene, views=None):
    """pmhip_set_tuning (include/pmhip.h): the mapping of a batch onto the GPU is chosen through the C ABI, not through the environment; every setting gives the
    oracle's bits."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    sc = nine_scene
    od, on, oc = _oracle(sc, 4, 5)
    e = PatchMatchHIP(0)
    t0 = e.tuning()
    assert t0["viewGroups"] >= 1 and t0["quadBuffer"] in (1, 2)
    per = (sc.n_views if views is None else len(views)) / 2.0        # reference views per group in the two-group settings: a launch has (diagonal length x per) pixels
    for kw in (dict(wideMaxViews=-1, sweepLanes=4), dict(wideMaxViews=-1, sweepLanes=8, viewGroups=3), dict(wideMaxViews=64, wideHyps=2), dict(wideMaxViews=64, wideHyps=8, quadBuffer=2),
               dict(wideMaxViews=-1, sweepLanes=-1, quadBuffer=1, viewGroups=1),
               # per-launch choice: the short diagonals of a batch with the eight-wide speculative kernel, the middle ones with the two-wide one, the long ones with pm_sweep2
               dict(wideMaxViews=-1, sweepLanes=4, viewGroups=2, widePixels=max(8, int(sc.height * per * 0.45)), wide8Pixels=max(4, int(sc.height * per * 0.12))),
               # three view groups, each running its whole pass on its own stream, with the two-wide speculative kernel
               dict(wideMaxViews=64, wideHyps=-1, widePixels=-1, wide8Pixels=-1, viewGroups=3)):
        got = e.tuning(**kw)
        for k, v in kw.items():
            assert got[k] == v, (k, got)
        e.Init(False); e.scene_load(sc, n_levels=2)
        e.scene_estimate(list(range(sc.n_views)) if views is None else list(views), -1, default_params(seed=5))   # (the CPU emulator's run estimates a batch of two: one view per group)
        d, n, c = e.scene_get_maps(4)
        _same(d, od, "tuning %s: depth" % kw); _same(n, on, "normal"); _same(c, oc, "conf")
    with pytest.raises(Exception):
        e.tuning(sweepLanes=5)
    e.close()


def test_wide_latency_mode_parity(nine_scene, small_scene, quick=False, hyps="8"):
    """The one-wave-per-pixel sweep kernel (eight hypotheses of a pixel scored side by side, sequential accept rule replayed over them) gives the bits of the sequential walk:
    8 / 4 / 1-3 sources, pyramid, geometric round, ignore masks, option sets that change the iteration budget (nRandomIters 8 and 2: more and fewer than one round holds),
    low-confidence pixels that take the random-restart stage.  Chosen through pmhip_set_tuning: wideMaxViews = 16 makes every batch here speculative.
    hyps: "8" = this test's subject, the eight-wide kernel, for every batch size; "4" / "2" = pm_sweep_widen_kernel; None = the engine's choice by batch size."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    e = PatchMatchHIP(0)
    e.tuning(wideMaxViews=16, wideHyps=-1 if hyps is None else int(hyps))
    for k in ((0, 5) if quick else (0, 3, 5)):
        test_non_default_options_parity(e, small_scene, k)
    if quick:                                                          # the CPU emulator run: 8 sources with the pyramid, and the option sets above
        sc = nine_scene
        ids = [4] + list(sc.neighbors[4])
        e.Init(False)
        d, n, c = e.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[4], sc.dmax[4], params=default_params(seed=5))
        od, on, oc = _oracle(sc, 4, 5)
        _same(d, od, "wide: depth N=8"); _same(n, on, "wide: normal"); _same(c, oc, "wide: conf")
        e.close(); return
    test_single_view_parity_N8_and_N1(e, nine_scene)
    test_single_call_with_ignore_mask(e, small_scene)
    test_single_view_photometric_parity_N4(e, small_scene, 2)
    test_initial_estimate_is_honoured(e, small_scene)
    sc = nine_scene
    p = default_params(seed=5, nEstimationGeometricIters=1)
    e.Init(False); e.scene_load(sc, n_levels=2)
    allv = list(range(sc.n_views))
    e.scene_estimate(allv, -1, p)                                     # nine views in one batch, all in latency mode
    photo = [e.scene_get_maps(v) for v in allv]
    e.scene_commit_round(); e.Init(True)
    e.scene_estimate([4, 0], 0, p)
    for v in (4, 0):
        od, on, oc = _oracle(sc, v, 5, nEstimationGeometricIters=1)
        _same(photo[v][0], od, f"wide: photometric depth v{v}"); _same(photo[v][2], oc, f"wide: photometric conf v{v}")
        gd, gn, gc = _oracle(sc, v, 5, geo_iter=0, depth=od, normal=on, src={u: photo[u][0] for u in allv}, nEstimationGeometricIters=1)
        d, n, c = e.scene_get_maps(v)
        _same(d, gd, f"wide: geometric depth v{v}"); _same(n, gn, "normal"); _same(c, gc, "conf")
    e.close()


def test_non_divisible_image_size_parity(engine):
    # 163x121 with 2 sub-levels: level sizes cvRound -> 82x60, 41x30; INTER_AREA border rule on both axes
    sc = synth.make_scene(4, 163, 121, n_src=3)
    engine.Init(False)
    r = np.random.RandomState(5)
    img = r.rand(121, 163).astype(np.float32)
    _same(engine.resize(0, img, 2), po.resize_area(img, 2), "area 2 odd"); _same(engine.resize(0, img, 4), po.resize_area(img, 4), "area 4 odd")
    d0 = (sc.gt_depth[2] * (1 + 0.01 * r.randn(121, 163))).astype(np.float32)     # exercises the nearest down-sampling of the initial estimate
    n0 = np.zeros((121, 163, 3), np.float32); n0[..., 2] = -1
    p = default_params(seed=4)
    ids = [2] + list(sc.neighbors[2])
    d, n, c = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[2], sc.dmax[2], depth=d0, normal=n0, params=p)
    od, on, oc = _oracle(sc, 2, 4, depth=d0, normal=n0)
    _same(d, od, "depth"); _same(n, on, "normal"); _same(c, oc, "conf")


def test_initial_estimate_is_honoured(engine, small_scene):
    sc = small_scene
    engine.Init(False)
    r = np.random.RandomState(2)
    d0 = (sc.gt_depth[0] * (1 + 0.02 * r.randn(*sc.gt_depth[0].shape))).astype(np.float32)
    d0[::3] = 0                                        # unset rows -> random init there
    n0 = np.zeros(d0.shape + (3,), np.float32); n0[..., 2] = -1; n0[:, ::4] = 0   # some unset normals
    p = default_params(seed=3)
    ids = [0] + list(sc.neighbors[0])
    d, n, c = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[0], sc.dmax[0], depth=d0, normal=n0, params=p)
    od, on, oc = _oracle(sc, 0, 3, depth=d0, normal=n0)
    _same(d, od, "depth"); _same(n, on, "normal"); _same(c, oc, "conf")


def test_geometric_round_parity_and_golden(engine):
    g = np.load(os.path.join(GOLD, "pm_golden_96x64.npz"))
    gray, K, R, Cc, nbr, dmin, dmax = (g[k] for k in ("gray", "K", "R", "C", "neighbors", "dmin", "dmax"))
    nv = int(g["n_views"])
    p = default_params(seed=int(g["seed"]))
    engine.Init(False)
    photo = {}
    for v in range(nv):
        ids = [v] + list(nbr[v])
        photo[v] = engine.EstimateDepthMap(gray, K, R, Cc, ids, dmin[v], dmax[v], params=p)
        _same(photo[v][0], g["depth_photo_all"][v], f"photometric depth view {v} vs golden")
    _same(photo[0][1], g["normal_photo"], "normal vs golden"); _same(photo[0][2], g["conf_photo"], "conf vs golden")
    engine.Release(); engine.Init(True)                 # SceneDensify.cpp:1910-1916
    ids = [0] + list(nbr[0])
    d, n, c = engine.EstimateDepthMap(gray, K, R, Cc, ids, dmin[0], dmax[0], depth=photo[0][0], normal=photo[0][1],
                                      src_depths={v: photo[v][0] for v in range(nv)}, nGeometricIter=0, params=p)
    _same(d, g["depth_geo0"], "geo depth vs golden"); _same(n, g["normal_geo0"], "geo normal"); _same(c, g["conf_geo0"], "geo conf")
    engine.Release(); engine.Init(False)


def test_scene_batch_full_schedule_matches_oracle(small_scene):
    """HBM-resident scene path: all views concurrently, photometric + 2 geometric rounds."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    sc = small_scene
    seed = 21
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(sc, n_levels=2)
    allv = list(range(sc.n_views))
    p = default_params(seed=seed)
    e.scene_estimate(allv, -1, p)
    e.scene_commit_round()
    e.Init(True)
    ref = {v: _oracle(sc, v, seed) for v in allv}
    for v in allv:
        d, n, c = e.scene_get_maps(v)
        _same(d, ref[v][0], f"photo depth v{v}"); _same(n, ref[v][1], f"photo normal v{v}"); _same(c, ref[v][2], f"photo conf v{v}")
    for geo in range(2):
        e.scene_estimate(allv, geo, p)
        e.scene_commit_round()
        prev = {v: ref[v][0] for v in allv}
        ref = {v: _oracle(sc, v, seed, geo_iter=geo, depth=ref[v][0], normal=ref[v][1], src=prev) for v in allv}
        for v in allv:
            d, n, c = e.scene_get_maps(v)
            _same(d, ref[v][0], f"geo{geo} depth v{v}"); _same(n, ref[v][1], f"geo{geo} normal v{v}"); _same(c, ref[v][2], f"geo{geo} conf v{v}")
    e.close()


@pytest.mark.parametrize("adjust", [True, False])
def test_filter_depth_map_parity(small_scene, adjust):
    """Scene::DenseReconstructionFilter on the HBM-resident scene vs the FilterDepthMap oracle, bit for bit."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    sc = small_scene
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(sc, n_levels=2)
    allv = list(range(sc.n_views))
    e.scene_estimate(allv, -1, default_params(seed=2, nEstimationGeometricIters=0))
    pre = [e.scene_get_maps(v) for v in allv]
    e.scene_filter(allv, bAdjust=adjust)
    dep = np.stack([p[0] for p in pre]); cnf = np.stack([p[2] for p in pre])
    changed = 0
    for v in allv:
        d, n, c = e.scene_get_maps(v)
        rc, od, oc = po.filter_depth_map(dep, cnf, sc.K, sc.R, sc.C, v, list(sc.neighbors[v]), sc.dmin[v], sc.dmax[v], bAdjust=adjust)
        assert rc == 0
        _same(d, od, f"filtered depth v{v}"); _same(c, oc, f"filtered conf v{v}"); _same(n, pre[v][1], "normals untouched")
        changed += int((d != pre[v][0]).sum())
        assert ((od > 0).sum() > 0.5 * (pre[v][0] > 0).sum())
    assert changed > 0
    e.close()


def test_remove_small_segments_parity(small_scene):
    from openmvs_amd.patchmatch import PatchMatchHIP
    sc = small_scene
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(sc, n_levels=2)
    allv = list(range(sc.n_views))
    e.scene_estimate(allv, -1, default_params(seed=6, nEstimationGeometricIters=0))
    r = np.random.RandomState(12)
    th = np.float32(0.007)
    pre = []
    for v in allv:
        d, n, c = e.scene_get_maps(v)
        if v >= 2:   # synthetic terraces right at the similarity threshold: plenty of one-directional edges and small islands
            lv = (2.0 * (1 + float(th)) ** (r.randint(0, 6, d.shape) * r.choice([0.97, 1.0, 1.03]))).astype(np.float32)
            blk = np.kron(r.rand(d.shape[0] // 4, d.shape[1] // 4) < 0.5, np.ones((4, 4), bool))
            d = np.where(blk, lv, d).astype(np.float32); d[r.rand(*d.shape) < 0.1] = 0
            n = n.copy(); n[d > 0] = [0, 0, -1]; n[d == 0] = 0
        e.scene_set_maps(v, d, n)
        pre.append((d, n, e.scene_get_maps(v)[2]))
    e.scene_remove_small_segments(allv, nSpeckleSize=40)
    removed = 0
    for v in allv:
        d, n, c = e.scene_get_maps(v)
        od, on, oc = po.remove_small_segments(pre[v][0], pre[v][1], pre[v][2], nSpeckleSize=40)
        _same(d, od, f"speckle depth v{v}"); _same(n, on, f"speckle normal v{v}"); _same(c, oc, f"speckle conf v{v}")
        removed += int(((pre[v][0] > 0) & (d == 0)).sum())
    assert removed > 50
    e.close()


def test_gap_interpolation_parity(small_scene):
    from openmvs_amd.patchmatch import PatchMatchHIP
    sc = small_scene
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(sc, n_levels=2)
    allv = list(range(sc.n_views))
    e.scene_estimate(allv, -1, default_params(seed=6, nEstimationGeometricIters=0))
    r = np.random.RandomState(8)
    pre = []
    for v in allv:                                     # punch extra holes of assorted sizes into the estimated maps
        d, n, c = e.scene_get_maps(v)
        for _ in range(60):
            y, x = r.randint(6, sc.height - 12), r.randint(6, sc.width - 14)
            hh, ww = r.randint(1, 11), r.randint(1, 11)
            d[y:y + hh, x:x + ww] = 0; n[y:y + hh, x:x + ww] = 0; c[y:y + hh, x:x + ww] = 0
        e.scene_set_maps(v, d, n); pre.append((d, n, c))
    # conf has no setter: compare on maps whose conf comes from the device (download again after set)
    pre = [(p[0], p[1], e.scene_get_maps(v)[2]) for v, p in enumerate(pre)]
    e.scene_gap_interpolation(allv)
    filled = 0
    for v in allv:
        d, n, c = e.scene_get_maps(v)
        od, on, oc = po.gap_interpolation(*pre[v])
        _same(d, od, f"gap depth v{v}"); _same(n, on, f"gap normal v{v}"); _same(c, oc, f"gap conf v{v}")
        filled += int(((d > 0) & (pre[v][0] == 0)).sum())
    assert filled > 100
    e.close()


def test_post_filter_option_sweep(small_scene):
    """The thresholds and sizes of the three post-filters away from their defaults, all on one estimate: RemoveSmallSegments (speckle size 0, 1, 15,
    huge; loose / tight depth threshold), GapInterpolation (gap 0, 1, 3, 20), FilterDepthMap (view counts 1..3, tight / loose threshold, both
    bFilterAdjust branches)."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    sc = small_scene
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(sc, n_levels=1)
    allv = list(range(sc.n_views))
    e.scene_estimate(allv, -1, default_params(seed=4, nSubResolutionLevels=1, nEstimationGeometricIters=0))
    r = np.random.RandomState(3)
    base = []
    for v in allv:                                     # holes of assorted sizes so that every filter has something to do
        d, n, c = e.scene_get_maps(v)
        for _ in range(40):
            y, x = r.randint(6, sc.height - 12), r.randint(6, sc.width - 14)
            hh, ww = r.randint(1, 9), r.randint(1, 9)
            d[y:y + hh, x:x + ww] = 0; n[y:y + hh, x:x + ww] = 0; c[y:y + hh, x:x + ww] = 0
        base.append((d, n, c))

    def restore():
        for v in allv:
            e.scene_set_maps(v, base[v][0], base[v][1]); e.scene_set_conf(v, base[v][2])
    dep = np.stack([b[0] for b in base]); cnf = np.stack([b[2] for b in base])
    for size, th in ((0, 0.01), (1, 0.01), (15, 0.01), (100000, 0.01), (40, 0.002), (40, 0.08)):
        restore(); e.scene_remove_small_segments(allv, nSpeckleSize=size, fDepthDiffThreshold=th)
        for v in allv:
            got = e.scene_get_maps(v); want = po.remove_small_segments(*base[v], nSpeckleSize=size, fDepthDiffThreshold=th)
            for a, b, what in zip(got, want, ("depth", "normal", "conf")):
                _same(a, b, f"segments size {size} th {th} v{v} {what}")
    for gap, th in ((0, 0.01), (1, 0.01), (3, 0.01), (20, 0.01), (7, 0.001), (7, 0.1)):
        restore(); e.scene_gap_interpolation(allv, nIpolGapSize=gap, fDepthDiffThreshold=th)
        for v in allv:
            got = e.scene_get_maps(v); want = po.gap_interpolation(*base[v], nIpolGapSize=gap, fDepthDiffThreshold=th)
            for a, b, what in zip(got, want, ("depth", "normal", "conf")):
                _same(a, b, f"gap {gap} th {th} v{v} {what}")
    for adjust, mv, mva, th in ((True, 1, 1, 0.01), (True, 3, 2, 0.01), (False, 3, 1, 0.01), (True, 2, 1, 0.001), (False, 2, 1, 0.1)):
        restore(); e.scene_filter(allv, bAdjust=adjust, nMinViewsFilter=mv, nMinViewsFilterAdjust=mva, fDepthDiffThreshold=th)
        for v in allv:
            d, n, c = e.scene_get_maps(v)
            rc, od, oc = po.filter_depth_map(dep, cnf, sc.K, sc.R, sc.C, v, list(sc.neighbors[v]), sc.dmin[v], sc.dmax[v], bAdjust=adjust,
                                             nMinViewsFilter=mv, nMinViewsFilterAdjust=mva, fDepthDiffThreshold=th)
            assert rc == 0
            _same(d, od, f"filter {adjust} {mv} {mva} {th} depth v{v}"); _same(c, oc, f"filter {adjust} {mv} {mva} {th} conf v{v}")
    e.close()


def test_whole_dense_schedule_matches_the_oracle_pipeline(small_scene, tmp_path):
    """Scene::ComputeDepthMaps order of operations (SceneDensify.cpp:1884-1980): photometric, 2 geometric rounds, speckle + gap
    filters after the last round, cross-view filter, .dmap files -- device pipeline vs the same chain of oracle stages."""
    from openmvs_amd import densify, dmap
    from openmvs_amd.patchmatch import PatchMatchHIP
    sc = small_scene
    seed = 31
    e = PatchMatchHIP(0)
    e.scene_load(sc, n_levels=2)
    allv = list(range(sc.n_views))
    p = default_params(seed=seed)
    densify.compute_depth_maps(e, allv, p, n_speckle_size=30)
    paths = densify.save_depth_maps(e, sc, allv, str(tmp_path))
    # oracle chain
    cur = {v: _oracle(sc, v, seed) for v in allv}
    for geo in range(2):
        prev = {v: cur[v][0] for v in allv}
        cur = {v: _oracle(sc, v, seed, geo_iter=geo, depth=cur[v][0], normal=cur[v][1], src=prev) for v in allv}
    cur = {v: po.gap_interpolation(*po.remove_small_segments(*cur[v], nSpeckleSize=30)) for v in allv}
    dep = np.stack([cur[v][0] for v in allv]); cnf = np.stack([cur[v][2] for v in allv])
    for v in allv:
        rc, fd, fc = po.filter_depth_map(dep, cnf, sc.K, sc.R, sc.C, v, list(sc.neighbors[v]), sc.dmin[v], sc.dmax[v])
        assert rc == 0
        f = dmap.load(paths[v])
        _same(f["depth_map"], fd, f"final depth v{v}"); _same(f["confidence_map"], fc, f"final conf v{v}"); _same(f["normal_map"], cur[v][1], f"final normal v{v}")
        assert f["reference_view_id"] == v and f["neighbor_view_ids"] == [int(i) for i in sc.neighbors[v]]
        m = fd > 0
        assert m.mean() > 0.5 and np.median(np.abs(fd[m] - sc.gt_depth[v][m]) / sc.gt_depth[v][m]) < 2e-3
    e.close()


def test_ignore_mask_parity(small_scene):
    """Ignore masks (pmhip_scene_set_mask): photometric pass over 3 levels + one geometric round; the masked view and an unmasked
    view of the same scene (which then also uses the NEAREST depth hand-off) both equal the oracle bit for bit."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    sc = small_scene
    seed = 11
    mask = np.ones((sc.height, sc.width), np.uint8); mask[20:70, 30:90] = 0; mask[::7, ::5] = 0; mask[100:, :12] = 0
    e = PatchMatchHIP(0)
    e.scene_load(sc, n_levels=2)
    e.scene_set_mask(0, mask)
    p = default_params(seed=seed, nEstimationGeometricIters=1)
    allv = list(range(sc.n_views))
    e.Init(False)
    for v in allv:
        e.scene_reset_view(v)
    e.scene_estimate(allv, -1, p)
    photo = {v: e.scene_get_maps(v) for v in allv}
    e.scene_commit_round(); e.Init(True)
    e.scene_estimate(allv, 0, p)
    geo = {v: e.scene_get_maps(v) for v in allv}

    def orc(v, geo_iter=-1, depth=None, normal=None, src=None):
        ids = [v] + list(sc.neighbors[v])
        vw, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps=src)
        return po.estimate_depth_map_masked(vw, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), po.default_opt(seed=seed, viewID=v, nEstimationGeometricIters=1),
                                            mask if v == 0 else None, geo_iter=geo_iter, depth=depth, normal=normal, mask_mode=True)
    ophoto = {v: orc(v) for v in allv}
    prev = {v: ophoto[v][0] for v in allv}
    for v in (0, 2):
        for k, what in enumerate(("depth", "normal", "conf")):
            _same(photo[v][k], ophoto[v][k], f"masked photometric view {v} {what}")
        og = orc(v, geo_iter=0, depth=ophoto[v][0], normal=ophoto[v][1], src=prev)
        for k, what in enumerate(("depth", "normal", "conf")):
            _same(geo[v][k], og[k], f"masked geometric view {v} {what}")
    assert not geo[0][0][mask == 0].any() and (geo[0][0][mask != 0] > 0).mean() > 0.5
    e.scene_set_mask(0, None)                    # removing the mask restores the plain estimator (LINEAR hand-off again)
    e.Init(False); e.scene_reset_view(2); e.scene_estimate([2], -1, p)
    plain = _oracle(sc, 2, seed, nEstimationGeometricIters=1)
    _same(e.scene_get_maps(2)[0], plain[0], "mask removed")
    e.close()


def test_single_call_with_ignore_mask(engine, small_scene):
    """pmhip_estimate_depth_map_masked (the PatchMatchCUDA-shaped call with DepthData::mask): with a mask, with the option but no mask, and a
    plain call afterwards (the engine keeps no mask state between calls)."""
    sc = small_scene
    engine.Init(False)
    p = default_params(seed=5, nSubResolutionLevels=2)
    ids = [1] + list(sc.neighbors[1])
    mask = np.full((sc.height, sc.width), 255, np.uint8); mask[10:60, 40:100] = 0; mask[::9, ::4] = 0
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
    opt = po.default_opt(seed=5, viewID=1, nSubResolutionLevels=2)
    for m, option in ((mask, False), (None, True)):
        got = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[1], sc.dmax[1], params=p, mask=m, mask_option=option)
        want = po.estimate_depth_map_masked(views, len(ids), float(sc.dmin[1]), float(sc.dmax[1]), opt, m, mask_mode=True)
        for k, what in enumerate(("depth", "normal", "conf")):
            _same(got[k], want[k], "single call, mask %s: %s" % ("given" if m is not None else "option only", what))
        if m is not None:
            assert not got[0][m == 0].any() and (got[0][m != 0] > 0).mean() > 0.5
    d, n, c = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[1], sc.dmax[1], params=p)
    od, on, oc = _oracle(sc, 1, 5, nSubResolutionLevels=2)
    _same(d, od, "plain call after masked calls")


def test_many_source_views_parity(engine):
    """9 .. 16 source views (G = 16 lanes per pixel, the widest instantiation, 4 pixels per wavefront) and the partial groups 5, 6; one geometric round
    at 16; more than PMHIP_MAX_SOURCES is an argument error."""
    from openmvs_amd.patchmatch import PatchMatchError
    sc = synth.make_scene(18, 96, 72, n_src=17)
    ref = 8
    engine.Init(False)
    p = default_params(seed=9, nSubResolutionLevels=1)
    res = {}
    for nsrc in (16, 12, 9, 6, 5):
        ids = [ref] + list(sc.neighbors[ref][:nsrc])
        d, n, c = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[ref], sc.dmax[ref], params=p)
        views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
        od, on, oc = po.estimate_depth_map(views, len(ids), float(sc.dmin[ref]), float(sc.dmax[ref]), po.default_opt(seed=9, viewID=ref, nSubResolutionLevels=1))
        _same(d, od, f"depth N={nsrc}"); _same(n, on, f"normal N={nsrc}"); _same(c, oc, f"conf N={nsrc}")
        assert (d > 0).mean() > 0.5
        res[nsrc] = (d, n)
    ids = [ref] + list(sc.neighbors[ref][:16])
    src = {i: np.full((sc.height, sc.width), float(sc.dmin[ref] + sc.dmax[ref]) / 2, np.float32) for i in ids[1:]}     # any maps do: both sides read the same
    engine.Init(True)
    g = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[ref], sc.dmax[ref], depth=res[16][0], normal=res[16][1], src_depths=src, nGeometricIter=0,
                                params=default_params(seed=9, nSubResolutionLevels=1, nEstimationGeometricIters=1))
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps=src)
    og = po.estimate_depth_map(views, len(ids), float(sc.dmin[ref]), float(sc.dmax[ref]),
                               po.default_opt(seed=9, viewID=ref, nSubResolutionLevels=1, nEstimationGeometricIters=1), geo_iter=0, depth=res[16][0], normal=res[16][1])
    for a, b, what in zip(g, og, ("depth", "normal", "conf")):
        _same(a, b, "geometric N=16 " + what)
    engine.Init(False)
    with pytest.raises(PatchMatchError):
        engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, [ref] + list(sc.neighbors[ref][:17]), sc.dmin[ref], sc.dmax[ref], params=p)


def test_degenerate_inputs(engine):
    """Textureless images (every pixel fails the descriptor-magnitude test: empty maps), an image smaller than the patch at its coarsest level, and
    the post-filters and the fusion on empty maps."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    sc = synth.make_scene(5, 64, 48, n_src=4)
    engine.Init(False)
    flat = np.full_like(sc.gray, 0.5)
    ids = [0] + list(sc.neighbors[0])
    d, n, c = engine.EstimateDepthMap(flat, sc.K, sc.R, sc.C, ids, sc.dmin[0], sc.dmax[0], params=default_params(seed=3))
    views, keep = po.make_views(flat, sc.K, sc.R, sc.C, ids)
    od, on, oc = po.estimate_depth_map(views, len(ids), float(sc.dmin[0]), float(sc.dmax[0]), po.default_opt(seed=3, viewID=0))
    _same(d, od, "textureless depth"); _same(n, on, "textureless normal"); _same(c, oc, "textureless conf")
    assert not d.any() and not n.any()
    tiny = synth.make_scene(3, 24, 20, n_src=2)
    ids = [0] + list(tiny.neighbors[0])
    for lv in (0, 1):
        d, n, c = engine.EstimateDepthMap(tiny.gray, tiny.K, tiny.R, tiny.C, ids, tiny.dmin[0], tiny.dmax[0], params=default_params(seed=3, nSubResolutionLevels=lv))
        views, keep = po.make_views(tiny.gray, tiny.K, tiny.R, tiny.C, ids)
        od, on, oc = po.estimate_depth_map(views, len(ids), float(tiny.dmin[0]), float(tiny.dmax[0]), po.default_opt(seed=3, viewID=0, nSubResolutionLevels=lv))
        _same(d, od, "tiny depth, %d levels" % lv); _same(n, on, "tiny normal"); _same(c, oc, "tiny conf")
    from openmvs_amd.patchmatch import PatchMatchError
    with pytest.raises(PatchMatchError, match="too small"):                       # 6 x 5 at level 2: smaller than the 9 x 9 patch -- engine and oracle both refuse
        engine.EstimateDepthMap(tiny.gray, tiny.K, tiny.R, tiny.C, ids, tiny.dmin[0], tiny.dmax[0], params=default_params(seed=3, nSubResolutionLevels=2))
    with pytest.raises(RuntimeError):
        po.estimate_depth_map(views, len(ids), float(tiny.dmin[0]), float(tiny.dmax[0]), po.default_opt(seed=3, viewID=0, nSubResolutionLevels=2))
    e = PatchMatchHIP(0)
    e.scene_load(sc, n_levels=0)
    allv = list(range(sc.n_views))
    for v in allv:
        e.scene_reset_view(v)
    e.scene_remove_small_segments(allv, 100, 0.01); e.scene_gap_interpolation(allv, 7, 0.01); e.scene_filter(allv)
    assert not any(e.scene_get_maps(v)[0].any() for v in allv)
    cloud = e.scene_fuse(allv, nMinViewsFuse=2, bEstimateColor=False)
    assert cloud["nPoints"] == 0 and cloud["nDepths"] == 0
    e.close()


OPTION_SETS = [
    dict(nEstimationIters=4, nRandomIters=8),
    dict(fRandomDepthRatio=0.01, fRandomAngle1Range=10.0, fRandomAngle2Range=5.0),
    dict(fRandomSmoothBonus=0.8, fRandomSmoothDepth=0.05, fRandomSmoothNormal=20.0),
    dict(fNCCThresholdKeep=0.6, fDescriptorMinMagnitudeThreshold=0.05),
    dict(nSubResolutionLevels=1, nEstimationIters=2),
    dict(nSubResolutionLevels=3, nRandomIters=2),
]


@pytest.mark.parametrize("k", range(len(OPTION_SETS)))
def test_non_default_options_parity(engine, small_scene, k):
    """Every OPTDENSE value the estimator reads (DepthMap.cpp:69-90 -> PMHipParams), away from its default: photometric pass, and for the first set a
    geometric round with a non-default fEstimationGeometricWeight on top."""
    sc = small_scene
    kw = OPTION_SETS[k]
    engine.Init(False)
    p = default_params(seed=21 + k, **kw)
    ids = [2] + list(sc.neighbors[2])
    d, n, c = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[2], sc.dmax[2], params=p)
    od, on, oc = _oracle(sc, 2, 21 + k, **kw)
    _same(d, od, "depth %s" % kw); _same(n, on, "normal %s" % kw); _same(c, oc, "conf %s" % kw)
    assert (d > 0).mean() > 0.4
    if k == 0:
        kw = dict(kw, nEstimationGeometricIters=1, fEstimationGeometricWeight=0.3)
        src = {i: _oracle(sc, i, 21, **OPTION_SETS[0])[0] for i in ids[1:]}
        engine.Init(True)
        g = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[2], sc.dmax[2], depth=d, normal=n, src_depths=src, nGeometricIter=0,
                                    params=default_params(seed=21, **kw))
        og = _oracle(sc, 2, 21, geo_iter=0, depth=od, normal=on, src=src, **kw)
        for a, b, what in zip(g, og, ("depth", "normal", "conf")):
            _same(a, b, "geometric, weight 0.3: " + what)
        engine.Init(False)


def test_real_scene_from_mvs_archive_matches_oracle_and_sfm_points():
    """The reference's own pipeline fixture (tests/data/scene: 4 JPEGs 640x479 + MVSI archive): scene front end (reader, view selection,
    sparse initialisation) -> photometric pass + 2 geometric rounds seeded by the sparse maps.  Device == oracle bit for bit on real
    images, and both agree with the SfM points they never saw as constraints (only as 2x2 seeds) to well under a percent."""
    from openmvs_amd import densify, mvsi, views
    from openmvs_amd.patchmatch import PatchMatchHIP
    path = os.path.join(os.path.dirname(__file__), "data", "scene", "scene.mvs")
    sv = densify.load_scene(path)
    seed = 5
    e = PatchMatchHIP(0)
    e.scene_load(sv, n_levels=2)
    p = default_params(seed=seed)
    densify.compute_depth_maps(e, sv.ids, p, n_optimize=0, init_depth=sv.init_depth, init_normal=sv.init_normal)
    got = {v: e.scene_get_maps(v) for v in sv.ids}

    def orc(v, geo=-1, depth=None, normal=None, src=None):
        ids = [v] + list(sv.neighbors[v])
        vw, keep = po.make_views(sv.gray, sv.K, sv.R, sv.C, ids, depth_maps=src)
        return po.estimate_depth_map(vw, len(ids), float(sv.dmin[v]), float(sv.dmax[v]), po.default_opt(seed=seed, viewID=v),
                                     geo_iter=geo, depth=depth, normal=normal)
    cur = {v: orc(v, depth=sv.init_depth[v], normal=sv.init_normal[v]) for v in sv.ids}
    for g in range(2):
        prev = {v: cur[v][0] for v in sv.ids}
        cur = {v: orc(v, geo=g, depth=cur[v][0], normal=cur[v][1], src=prev) for v in sv.ids}
    sc = mvsi.load(path); cams = views.Cameras(sc)
    for v in sv.ids:
        for k, what in enumerate(("depth", "normal", "conf")):
            _same(got[v][k], cur[v][k], f"real scene view {v} {what}")
        d = got[v][0]
        assert 0.6 < (d > 0).mean() < 0.95
        _, _, pts, _ = views.select_neighbor_views(sc, cams, v)
        X = sc.vertices[pts]; pr = cams.project_p(v, X); z = cams.point_depth(v, X)
        xi = np.rint(pr[:, 0]).astype(int); yi = np.rint(pr[:, 1]).astype(int)
        m = (xi >= 4) & (yi >= 4) & (xi < sv.width - 4) & (yi < sv.height - 4)
        dm = d[yi[m], xi[m]]; ok = dm > 0
        rel = np.abs(dm[ok] - z[m][ok]) / z[m][ok]
        assert ok.mean() > 0.95 and np.median(rel) < 5e-3 and np.percentile(rel, 90) < 3e-2
    # the complete schedule (speckle + gap filters, cross-view filter) keeps most of it
    densify.compute_depth_maps(e, sv.ids, p, init_depth=sv.init_depth, init_normal=sv.init_normal)
    final = {v: e.scene_get_maps(v) for v in sv.ids}
    for v in sv.ids:
        assert 0.5 < (final[v][0] > 0).mean() < 0.95
    # ... and fusing them passes the reference's own acceptance for this dataset: >= 200000 points (apps/Tests/Tests.cpp:86)
    from tests import fuse_cases as fcs
    from PIL import Image
    bgr = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(os.path.dirname(path), n)).convert("RGB"))[..., ::-1]) for n in sv.names]
    for v in range(sv.n_views):
        e.scene_set_color(v, bgr[v])
    order = po.fuse_order([len(x) for x in sv.neighbors])
    cloud = e.scene_fuse(order)
    ref = po.fuse_depth_maps([final[v][0] for v in sv.ids], [final[v][1] for v in sv.ids], [final[v][2] for v in sv.ids], bgr,
                             sv.K, sv.R, sv.C, [list(x) for x in sv.neighbors], order=order)
    fcs.same_cloud(cloud, ref, "real scene fuse")
    assert cloud["nPoints"] >= 200000
    e.close()


@pytest.mark.parametrize("wide", [0, 16])
def test_mixed_resolution_neighbours_parity_both_kernels(wide, W=160, H=120):
    """The shipped combination -- one-call boundary, one-wave-per-pixel kernel (the engine's choice for a single depth map), mixed-size sources, geometric
    round with resized depth maps -- and the regular kernel, each on a fresh engine (wide: PMHipTuning::wideMaxViews, 0 = no speculative kernels)."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    e = PatchMatchHIP(0)
    e.tuning(wideMaxViews=wide if wide else -1)
    test_mixed_resolution_neighbours_parity(e, W, H)
    e.close()


def test_mixed_resolution_neighbours_parity(engine, W=160, H=120):
    """Source views of another size than the reference (DepthMap.h:194-204: a neighbour whose scale differs by >= 15 % is rescaled; here one at 0.8x
    and one at 1.25x, rendered at those sizes): every view is projected into with its own K and sampled within its own bounds, on all pyramid
    levels; in the geometric round the neighbours' depth maps keep the size and camera they were stored with (cameraDepthMap, SceneDensify.cpp:
    378-393) while their images are the rescaled ones.  Through the one-call boundary and through the scene interface, bit-exact."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    base = synth.make_scene(5, W, H, n_src=4)
    small = synth.make_scene(5, W * 4 // 5, H * 4 // 5, n_src=4)   # same cameras and surface, rendered at 0.8x ...
    big = synth.make_scene(5, W * 5 // 4, H * 5 // 4, n_src=4)     # ... and at 1.25x
    assert np.array_equal(base.R, small.R) and np.array_equal(base.C, big.C) and np.array_equal(base.neighbors, small.neighbors)
    ref = 0
    ids = [ref] + [int(i) for i in base.neighbors[ref]]
    s1, s2 = ids[1], ids[2]
    gray = {i: base.gray[i] for i in range(5)}; K = {i: base.K[i] for i in range(5)}
    gray[s1] = small.gray[s1]; K[s1] = small.K[s1]
    gray[s2] = big.gray[s2]; K[s2] = big.K[s2]
    seed = 23
    p = default_params(seed=seed, nEstimationGeometricIters=1)
    # photometric pass, 3 levels
    views, keep = po.make_views(gray, K, base.R, base.C, ids)
    opt = po.default_opt(seed=seed, viewID=ref, nEstimationGeometricIters=1)
    od, on, oc = po.estimate_depth_map(views, len(ids), float(base.dmin[ref]), float(base.dmax[ref]), opt)
    engine.Init(False)
    d, n, c = engine.EstimateDepthMap(gray, K, base.R, base.C, ids, base.dmin[ref], base.dmax[ref], params=p)
    _same(d, od, "mixed sizes, photometric depth"); _same(n, on, "normal"); _same(c, oc, "conf")
    assert (d > 0).mean() > 0.5
    # geometric round: neighbour depth maps at the BASE size with the base cameras (as stored in their .dmap), images rescaled
    src = {}
    for i in ids[1:]:
        vi = [i] + [int(j) for j in base.neighbors[i]]
        vw, _k = po.make_views(base.gray, base.K, base.R, base.C, vi)
        src[i] = po.estimate_depth_map(vw, len(vi), float(base.dmin[i]), float(base.dmax[i]), po.default_opt(seed=seed, viewID=i, nEstimationGeometricIters=1))[0]
    cams = {i: (base.K[i], base.R[i], base.C[i]) for i in ids[1:]}
    views, keep = po.make_views(gray, K, base.R, base.C, ids, depth_maps=src, depth_cams=cams)
    gd, gn, gc = po.estimate_depth_map(views, len(ids), float(base.dmin[ref]), float(base.dmax[ref]), opt, geo_iter=0, depth=od, normal=on)
    engine.Init(True)
    d2, n2, c2 = engine.EstimateDepthMap(gray, K, base.R, base.C, ids, base.dmin[ref], base.dmax[ref], depth=d, normal=n, src_depths=src, src_depth_cams=cams,
                                         nGeometricIter=0, params=p)
    _same(d2, gd, "mixed sizes, geometric depth"); _same(n2, gn, "normal"); _same(c2, gc, "conf")
    assert (d2 != d).any()
    # the same engine, same-size call afterwards: nothing of the side storage may leak into it
    engine.Init(False)
    d3 = engine.EstimateDepthMap(base.gray, base.K, base.R, base.C, ids, base.dmin[ref], base.dmax[ref], params=p)[0]
    views, keep = po.make_views(base.gray, base.K, base.R, base.C, ids)
    _same(d3, po.estimate_depth_map(views, len(ids), float(base.dmin[ref]), float(base.dmax[ref]), opt)[0], "same-size call after a mixed one")
    # scene interface: the two resized views are installed as sized source views
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(base, n_levels=2)
    e.scene_set_view_sized(s1, gray[s1], K[s1], base.R[s1], base.C[s1], float(base.dmin[s1]), float(base.dmax[s1]), base.neighbors[s1])
    e.scene_set_view_sized(s2, gray[s2], K[s2], base.R[s2], base.C[s2], float(base.dmin[s2]), float(base.dmax[s2]), base.neighbors[s2])
    e.scene_estimate([ref], -1, p)
    sd, sn, scf = e.scene_get_maps(ref)
    _same(sd, od, "scene interface, sized sources, depth"); _same(sn, on, "normal"); _same(scf, oc, "conf")
    e.Init(True)
    for i in ids[1:]:
        e.scene_set_source_depth(i, src[i], *cams[i])
    e.scene_estimate([ref], 0, p)
    _same(e.scene_get_maps(ref)[0], gd, "scene interface, geometric round with installed source depth maps")
    e.close()


def resampled_neighbour_copies_through_the_driver(W=160, H=120):
    """ViewData::ScaleImage through the scene front end's bookkeeping (densify.SceneViews with `alias_of` / `estimate_neighbors`, as densify.load_scene builds them) and
    densify.compute_depth_maps: reference view 0 reads two of its neighbours as RESAMPLED copies (0.8x: INTER_AREA, 1.25x: INTER_CUBIC, densify.scale_image) in extra
    source-only slots; in the geometric round the copies stand for their images' previous-round depth maps at the images' own size and camera (the saved .dmap,
    SceneDensify.cpp:378-393).  Every view's final map equals the oracle given exactly those inputs."""
    from openmvs_amd import densify
    from openmvs_amd.patchmatch import PatchMatchHIP
    base = synth.make_scene(5, W, H, n_src=4)
    ref = 0
    nb0 = [int(i) for i in base.neighbors[ref]]
    s1, s2 = nb0[0], nb0[1]
    sv = densify.SceneViews()
    sv.width, sv.height = W, H
    sv.gray = [g for g in base.gray]; sv.K = [k for k in base.K]; sv.R = [r for r in base.R]; sv.C = [c for c in base.C]
    sv.dmin = [float(x) for x in base.dmin]; sv.dmax = [float(x) for x in base.dmax]
    sv.neighbors = [np.asarray(n, np.int32) for n in base.neighbors]; sv.ids = list(range(5)); sv.sizes = [(W, H)] * 5
    sv.estimate_neighbors = [n.copy() for n in sv.neighbors]
    for j, scale in ((s1, 0.8), (s2, 1.25)):
        img = densify.scale_image(base.gray[j], scale)
        h, w = img.shape
        f = max(w, h) / float(max(W, H))                                            # Image::GetCamera at the copy's size: K through the normalised form (Camera::ScaleK)
        Kc = np.array(base.K[j], np.float64)
        Kc = np.array([[Kc[0, 0] * f, 0, (Kc[0, 2] + 0.5) * f - 0.5], [0, Kc[1, 1] * f, (Kc[1, 2] + 0.5) * f - 0.5], [0, 0, 1]])
        a = len(sv.gray)
        sv.alias_of[a] = j
        sv.gray.append(img); sv.K.append(Kc); sv.R.append(base.R[j]); sv.C.append(base.C[j]); sv.sizes.append((w, h))
        sv.dmin.append(sv.dmin[j]); sv.dmax.append(sv.dmax[j]); sv.neighbors.append(np.zeros(0, np.int32)); sv.estimate_neighbors.append(np.zeros(0, np.int32))
        sv.estimate_neighbors[ref][list(sv.neighbors[ref]).index(j)] = a
    seed = 41
    p = default_params(seed=seed, nEstimationGeometricIters=1)
    e = PatchMatchHIP(0)
    e.scene_load(sv, n_levels=2)
    densify.compute_depth_maps(e, sv.ids, p, n_optimize=0, scene=sv)
    # the oracle, view by view: photometric pass (view 0 with the copies' images and cameras), then the geometric round on the photometric maps of the images themselves
    def views_of(i, depth_maps=None, depth_cams=None):
        slots = [i] + [int(s) for s in sv.estimate_neighbors[i]]
        gray = {s: sv.gray[s] for s in slots}; K = {s: sv.K[s] for s in slots}; R = {s: sv.R[s] for s in slots}; C = {s: sv.C[s] for s in slots}
        dm = None if depth_maps is None else {s: depth_maps[sv.alias_of.get(s, s)] for s in slots[1:]}
        dc = None if depth_cams is None else {s: depth_cams[sv.alias_of.get(s, s)] for s in slots[1:]}
        return po.make_views(gray, K, R, C, slots, depth_maps=dm, depth_cams=dc), slots
    photo = {}
    for i in sv.ids:
        (vw, keep), slots = views_of(i)
        photo[i] = po.estimate_depth_map(vw, len(slots), sv.dmin[i], sv.dmax[i], po.default_opt(seed=seed, viewID=i, nEstimationGeometricIters=1))
    cams = {i: (sv.K[i], sv.R[i], sv.C[i]) for i in sv.ids}
    for i in sv.ids:
        (vw, keep), slots = views_of(i, {j: photo[j][0] for j in sv.ids}, cams)
        gd, gn, gc = po.estimate_depth_map(vw, len(slots), sv.dmin[i], sv.dmax[i], po.default_opt(seed=seed, viewID=i, nEstimationGeometricIters=1), geo_iter=0,
                                           depth=photo[i][0], normal=photo[i][1])
        d, n, c = e.scene_get_maps(i)
        _same(d, gd, "view %d, depth after the geometric round" % i); _same(n, gn, "normal"); _same(c, gc, "conf")
    assert (e.scene_get_maps(ref)[0] > 0).mean() > 0.5
    e.close()


def test_reference_views_of_different_sizes(W=160, H=120, quick=False):
    """Reference views of different sizes in one scene: the reference sizes every DepthData on its own image (DepthMapsData::InitViews, SceneDensify.cpp:306-459).
    Five views, three sizes (1x, 0.8x, 1.25x), estimated by ONE call (one sweep per size class), then the geometric round (every view reads its neighbours'
    previous-round maps at the neighbours' sizes), the two per-map filters and the cross-view filter (neighbour maps of other sizes: :1085, :1181): every map
    equals the oracle run on that view alone."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    base = synth.make_scene(5, W, H, n_src=4)
    small = synth.make_scene(5, W * 4 // 5, H * 4 // 5, n_src=4)
    big = synth.make_scene(5, W * 5 // 4, H * 5 // 4, n_src=4)
    src_of = {0: base, 1: base, 2: small, 3: big, 4: base}
    gray = {i: src_of[i].gray[i] for i in range(5)}; K = {i: src_of[i].K[i] for i in range(5)}
    seed = 31
    p = default_params(seed=seed, nEstimationGeometricIters=1)
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(base, n_levels=2)
    for i in (2, 3):
        e.scene_set_view_sized(i, gray[i], K[i], base.R[i], base.C[i], float(base.dmin[i]), float(base.dmax[i]), base.neighbors[i])
    allv = list(range(5)) if not quick else [0, 2, 3]
    e.scene_estimate(allv, -1, p)
    maps = {}
    for v in allv:
        ids = [v] + [int(i) for i in base.neighbors[v]]
        views, keep = po.make_views(gray, K, base.R, base.C, ids)
        od, on, oc = po.estimate_depth_map(views, len(ids), float(base.dmin[v]), float(base.dmax[v]), po.default_opt(seed=seed, viewID=v, nEstimationGeometricIters=1))
        d, n, c = e.scene_get_maps(v)
        assert d.shape == gray[v].shape
        _same(d, od, "view %d (%dx%d), photometric depth" % (v, d.shape[1], d.shape[0])); _same(n, on, "normal"); _same(c, oc, "conf")
        assert (d > 0).mean() > 0.4
        maps[v] = (od, on, oc)
    if quick:
        e.close(); return
    # geometric round: the neighbours' previous-round maps at their own sizes with their own cameras
    e.scene_commit_round(); e.Init(True)
    e.scene_estimate(allv, 0, p)
    geo = {}
    for v in allv:
        ids = [v] + [int(i) for i in base.neighbors[v]]
        src = {i: maps[i][0] for i in ids[1:]}; cams = {i: (K[i], base.R[i], base.C[i]) for i in ids[1:]}
        views, keep = po.make_views(gray, K, base.R, base.C, ids, depth_maps=src, depth_cams=cams)
        gd, gn, gc = po.estimate_depth_map(views, len(ids), float(base.dmin[v]), float(base.dmax[v]), po.default_opt(seed=seed, viewID=v, nEstimationGeometricIters=1),
                                           geo_iter=0, depth=maps[v][0], normal=maps[v][1])
        d, n, c = e.scene_get_maps(v)
        _same(d, gd, "view %d, geometric depth" % v); _same(n, gn, "normal"); _same(c, gc, "conf")
        geo[v] = (gd, gn, gc)
    # per-map filters at each view's own size
    e.scene_remove_small_segments(allv, 30, 0.01); e.scene_gap_interpolation(allv, 7, 0.01)
    flt = {}
    for v in allv:
        a = po.remove_small_segments(*geo[v], nSpeckleSize=30, fDepthDiffThreshold=0.01)
        a = po.gap_interpolation(*a, nIpolGapSize=7, fDepthDiffThreshold=0.01)
        d, n, c = e.scene_get_maps(v)
        _same(d, a[0], "view %d, speckle + gap filter depth" % v); _same(n, a[1], "normal"); _same(c, a[2], "conf")
        flt[v] = a
    # cross-view filter: every view against neighbours whose maps have other sizes
    D = {v: flt[v][0] for v in allv}; Cf = {v: flt[v][2] for v in allv}
    for bAdjust in (True, False):
        e.scene_filter(allv, bAdjust=bAdjust, commit=True)
        for v in allv:
            rc, nd, nc = po.filter_depth_map(D, Cf, K, base.R, base.C, v, [int(i) for i in base.neighbors[v]], float(base.dmin[v]), float(base.dmax[v]), bAdjust=bAdjust)
            assert rc == 0
            d, n, c = e.scene_get_maps(v)
            _same(d, nd, "view %d, FilterDepthMap(bAdjust=%s) depth" % (v, bAdjust)); _same(c, nc, "conf")
        for v in allv:                                          # back to the unfiltered maps for the second variant
            e.scene_set_maps(v, D[v], None); e.scene_set_conf(v, Cf[v])
    e.close()


def test_ignore_mask_on_a_view_with_its_own_size(W=128, H=96):
    """pmhip_scene_set_mask on a view that carries its own size: its level masks are its own; photometric pass over the pyramid against the oracle."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    base = synth.make_scene(5, W, H, n_src=4)
    big = synth.make_scene(5, W * 5 // 4, H * 5 // 4, n_src=4)
    gray = {i: base.gray[i] for i in range(5)}; K = {i: base.K[i] for i in range(5)}
    gray[1] = big.gray[1]; K[1] = big.K[1]
    h1, w1 = gray[1].shape
    mask = np.ones((h1, w1), np.uint8); mask[10:50, 20:80] = 0; mask[::6, ::5] = 0
    seed = 17
    p = default_params(seed=seed)
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(base, n_levels=2)
    e.scene_set_view_sized(1, gray[1], K[1], base.R[1], base.C[1], float(base.dmin[1]), float(base.dmax[1]), base.neighbors[1])
    e.scene_set_mask(1, mask)
    e.scene_estimate([0, 1], -1, p)
    for v in (1, 0):
        ids = [v] + [int(i) for i in base.neighbors[v]]
        views, keep = po.make_views(gray, K, base.R, base.C, ids)
        od, on, oc = po.estimate_depth_map_masked(views, len(ids), float(base.dmin[v]), float(base.dmax[v]), po.default_opt(seed=seed, viewID=v), mask if v == 1 else None, mask_mode=True)
        d, n, c = e.scene_get_maps(v)
        _same(d, od, "view %d depth" % v); _same(n, on, "normal"); _same(c, oc, "conf")
    assert not e.scene_get_maps(1)[0][mask == 0].any()
    e.close()


def test_sized_view_api_edges(W=96, H=72):
    """Edges of the own-size views: a contiguous scene range that contains one is refused by pmhip_scene_copy (one by one it works); taking the view back to the scene's
    size releases its own storage and the scene estimates as if it had never been sized; the cross-view filter declines a view with too few neighbour maps whatever its size."""
    from openmvs_amd.patchmatch import PatchMatchError, PatchMatchHIP
    base = synth.make_scene(5, W, H, n_src=4)
    big = synth.make_scene(5, W * 5 // 4, H * 5 // 4, n_src=4)
    seed = 13
    p = default_params(seed=seed)
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(base, n_levels=2)
    e.scene_set_view_sized(2, big.gray[2], big.K[2], base.R[2], base.C[2], float(base.dmin[2]), float(base.dmax[2]), base.neighbors[2])
    assert e.view_size(2) == (W * 5 // 4, H * 5 // 4) and e.view_size(1) == (W, H)
    assert e.scene_device_ptr(1, 2) != 0 and e.scene_device_ptr(1, 2) != e.scene_device_ptr(1, 1) + 4 * W * H      # its maps are its own
    with pytest.raises(PatchMatchError):
        e.scene_copy(1, 1, 3, e.scene_device_ptr(4, 0), False)             # views 1..3 contain the sized view 2
    e.scene_estimate([2], -1, default_params(seed=seed, nSubResolutionLevels=1))
    d_big = e.scene_get_maps(2)[0]
    assert d_big.shape == big.gray[2].shape and (d_big > 0).mean() > 0.3
    e.scene_filter([2], nMinViewsFilter=2)                                # no neighbour has a depth map yet: not filterable, nothing staged, the maps stay
    _same(e.scene_get_maps(2)[0], d_big, "filter declined")
    # back to the scene's size
    e.scene_set_view(2, base.gray[2], base.K[2], base.R[2], base.C[2], float(base.dmin[2]), float(base.dmax[2]), base.neighbors[2])
    assert e.view_size(2) == (W, H)
    e.scene_estimate([0, 2], -1, p)
    for v in (0, 2):
        od, on, oc = _oracle(base, v, seed)
        d, n, c = e.scene_get_maps(v)
        _same(d, od, "view %d after un-sizing" % v); _same(n, on, "normal"); _same(c, oc, "conf")
    # a colour image follows its view's size class: after a view changes class, fusion with colours is refused until the colour is set again (it must not read a freed or
    # never-written image)
    e.scene_estimate([1, 3, 4], -1, p)
    for v in range(5):
        e.scene_set_color(v, base.bgr[v])
    order = po.fuse_order([len(base.neighbors[v]) for v in range(5)])
    n0 = e.scene_fuse(order)["nPoints"]
    e.scene_set_view_sized(2, big.gray[2], big.K[2], base.R[2], base.C[2], float(base.dmin[2]), float(base.dmax[2]), base.neighbors[2])
    with pytest.raises(PatchMatchError):
        e.scene_fuse(order)
    e.scene_set_color(2, big.bgr[2])
    e.scene_set_view(2, base.gray[2], base.K[2], base.R[2], base.C[2], float(base.dmin[2]), float(base.dmax[2]), base.neighbors[2])
    with pytest.raises(PatchMatchError):
        e.scene_fuse(order)
    e.scene_set_color(2, base.bgr[2]); e.scene_estimate([2], -1, p)
    assert e.scene_fuse(order)["nPoints"] == n0
    e.close()


def test_config2_full_size_matches_golden():
    """BASELINE config 2 at its own size: 9-view 1920x1080 scene, every view 1 ref x 8 src, photometric pass + 2 geometric rounds, against the digests the
    SEQUENTIAL oracle produced on the CPU (tests/golden/make_fullsize_golden.py, ~10 CPU-minutes; SceneDensify.cpp:616-805).  Bit-exact, all 27 maps;
    then the same reference view through the one-call boundary (pmhip_estimate_depth_map, SceneDensify.cpp:618-623) for all three rounds."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    from tests import golden_check as gc
    g = gc.load("pm_config2_1920x1080.json")
    c = g["case"]
    sc = synth.make_scene(c["n_views"], c["width"], c["height"], n_src=c["n_src"], device="cuda", gray_only=True, exact=True)
    assert gc.sha(sc.gray) == g["inputs"]["gray"], "the exact scene generator did not reproduce the golden inputs on this machine (images differ)"
    for k in ("K", "R", "C"):
        assert gc.sha(getattr(sc, k)) == g["inputs"][k], "camera " + k
    assert gc.sha(sc.neighbors.astype(np.int32)) == g["inputs"]["neighbors"]
    assert [float(x) for x in sc.dmin] == g["inputs"]["dmin"] and [float(x) for x in sc.dmax] == g["inputs"]["dmax"]
    p = default_params(seed=c["seed"], nEstimationGeometricIters=c["geo_iters"])
    allv = list(range(c["n_views"]))
    # the scene interface with every sweep kernel a batch can get, chosen through pmhip_set_tuning: pm_sweep2_kernel<4 lanes, 2 views per lane> alone; <8,1> (26-79 views);
    # the engine's own choice for nine views (the two-wide speculative kernel); and the MIX bench.py's timed 100-view batch runs -- <4,2> on the long diagonals and the
    # two-wide kernel on the short ones of the same sweep (threshold scaled to this batch: 20000 pixels / 50 views per group = 400 pixels of diagonal, x 4.5 views per
    # group here), with two and with three view groups
    for what, tun in (("pm_sweep2_kernel<4,2>", dict(wideMaxViews=-1, sweepLanes=4)), ("pm_sweep2_kernel<8,1>", dict(wideMaxViews=-1, sweepLanes=8)), ("engine default", {}),
                      ("timed mix, 2 groups", dict(wideMaxViews=-1, sweepLanes=4, widePixels=1800, viewGroups=2)),
                      ("timed mix, 3 groups", dict(wideMaxViews=-1, sweepLanes=4, widePixels=1200, viewGroups=3))):
        e = PatchMatchHIP(0); e.Init(True)
        if tun:
            e.tuning(**tun)
        e.scene_load(sc, n_levels=2)
        rounds = []
        for r in range(1 + c["geo_iters"]):
            if r:
                e.scene_commit_round()
            e.scene_estimate(allv, r - 1, p)
            rounds.append([e.scene_get_maps(v) for v in allv])
            for v in allv:
                gc.check_maps(rounds[r][v], g["rounds"][r][str(v)], "scene interface (%s), round %d, view %d" % (what, r, v))
        e.close()
    # the per-view boundary (layer 1): host buffers in and out, one blocking call per pass -- with the regular sweep kernel and with the
    # one-wave-per-pixel kernel the engine uses by default for a single depth map
    ref = c["ref"]
    ids = [ref] + list(sc.neighbors[ref])
    for what, tun in (("pm_sweep2_kernel", dict(wideMaxViews=-1)), ("pm_sweep_wide_kernel", dict(wideMaxViews=8))):
        e = PatchMatchHIP(0); e.Init(True); e.tuning(**tun)
        cur = e.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[ref], sc.dmax[ref], params=p)
        gc.check_maps(cur, g["rounds"][0][str(ref)], "one-call boundary (%s), photometric" % what)
        for r in range(1, 1 + c["geo_iters"]):
            cur = e.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[ref], sc.dmax[ref], depth=cur[0], normal=cur[1],
                                     src_depths={v: rounds[r - 1][v][0] for v in ids[1:]}, nGeometricIter=r - 1, params=p)
            gc.check_maps(cur, g["rounds"][r][str(ref)], "one-call boundary (%s), geometric round %d" % (what, r - 1))
        e.close()


def test_config5_resolution_estimate_filter_fuse():
    """BASELINE config 5's resolution (3840x2160), a 5-view slice of such a scene: photometric pass + one geometric round, speckle / gap filters, the cross-view filter
    and the fusion, all resident -- stage by stage against the digests the SEQUENTIAL oracle produced on the CPU (tests/golden/make_fullsize_golden.py c5, ~25 CPU-minutes;
    SceneDensify.cpp:616-805, :1049-1299, :809-1047, :1303-1646), bit-exact: every map of every stage and the fused cloud.  The sizes where 32-bit offsets or grid limits
    could bite (8.3 Mpix per map, 6 k diagonals) are exercised.  Then the same chain once more through densify.compute_depth_maps: the driver is the stage sequence, and
    the run repeats itself (schedule races would show)."""
    import time
    from openmvs_amd import densify
    from openmvs_amd.patchmatch import PatchMatchHIP
    from tests import golden_check as gc
    g = gc.load("pm_config5_3840x2160.json")
    c = g["case"]
    W, H, V = c["width"], c["height"], c["n_views"]
    sc = synth.make_scene(V, W, H, n_src=c["n_src"], device="cuda", exact=True)
    assert gc.sha(sc.gray) == g["inputs"]["gray"], "the exact scene generator did not reproduce the golden inputs on this machine (images differ)"
    assert gc.sha(np.stack([np.asarray(b) for b in sc.bgr])) == g["inputs"]["bgr"]
    for k in ("K", "R", "C"):
        assert gc.sha(getattr(sc, k)) == g["inputs"][k], "camera " + k
    allv = list(range(V))
    p = default_params(seed=c["seed"], nEstimationGeometricIters=c["geo_iters"])
    order = po.fuse_order([len(sc.neighbors[v]) for v in allv])
    e = PatchMatchHIP(0)
    e.scene_load(sc, n_levels=2)
    for v in allv:
        e.scene_set_color(v, sc.bgr[v])

    def stage(k):
        assert g["stages"][k]["name"]
        for v in allv:
            gc.check_maps(e.scene_get_maps(v), g["stages"][k][str(v)], "config-5 resolution, %s, view %d" % (g["stages"][k]["name"], v))
    t0 = time.time()
    e.Init(False)
    for v in allv:
        e.scene_reset_view(v)
    e.scene_estimate(allv, -1, p); stage(0)
    e.scene_commit_round(); e.Init(True)
    e.scene_estimate(allv, 0, p); stage(1)
    e.scene_remove_small_segments(allv, 100, 0.01); e.scene_gap_interpolation(allv, 7, 0.01); stage(2)
    e.scene_filter(allv, True, 2, 1, 0.01, commit=True); stage(3)
    first = [e.scene_get_maps(v) for v in allv]
    cloud = e.scene_fuse(order)
    gf = g["fuse"]
    assert cloud["nPoints"] == gf["nPoints"] and cloud["nDepths"] == gf["nDepths"], (cloud["nPoints"], gf["nPoints"])
    for k in ("points", "viewStart", "views", "weights", "projs", "colors", "normals"):
        assert gc.sha(cloud[k]) == gf[k], "config-5 resolution, fused cloud: %s differs from the sequential oracle's" % k
    print("config-5 resolution, %d views: whole chain incl. downloads %.1f s, %d points" % (V, time.time() - t0, cloud["nPoints"]))
    # the driver (Scene::ComputeDepthMaps order of operations) on a fresh engine: same maps, same cloud
    e.close()
    e = PatchMatchHIP(0)
    e.scene_load(sc, n_levels=2)
    for v in allv:
        e.scene_set_color(v, sc.bgr[v])
    t0 = time.time()
    densify.compute_depth_maps(e, allv, p)
    e.sync(); t1 = time.time()
    cloud2 = e.scene_fuse(order)
    t2 = time.time()
    print("config-5 resolution, %d views: estimate + filters %.2f s (%.1f Mpix/s), fuse %.2f s" % (V, t1 - t0, V * W * H / (t1 - t0) / 1e6, t2 - t1))
    for v in allv:
        for a, b, what in zip(first[v], e.scene_get_maps(v), ("depth", "normal", "conf")):
            _same(a, b, f"driver vs stage sequence v{v} {what}")
    assert cloud2["nPoints"] == cloud["nPoints"] and np.array_equal(cloud2["points"], cloud["points"]) and np.array_equal(cloud2["views"], cloud["views"])
    e.close()
    d = first[2][0]; m = d > 0
    gt = sc.gt_depth[2]
    rel = np.abs(d[m] - gt[m]) / gt[m]
    assert m.mean() > 0.8 and np.median(rel) < 1e-3 and (rel < 0.01).mean() > 0.95
    pts = cloud["points"]
    assert len(pts) > 1_000_000 and np.abs(pts[:, 2]).max() < 0.12     # the surface is a height field |z| <= 0.1 around z = 0


def test_full_size_properties():
    """BASELINE config 2 (1 ref x 8 src, 1920x1080): the oracle needs ~15 min here, so check
    size-independent properties: run-to-run determinism (race check of the diagonal schedule),
    accuracy against the analytic ground truth, and sane outputs."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    sc = synth.make_scene(9, 1920, 1080, n_src=8, device="cuda", gray_only=True)
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(sc, n_levels=2)
    p = default_params(seed=1, nEstimationGeometricIters=0)
    outs = []
    for _ in range(2):
        e.scene_reset_view(4)
        e.scene_estimate([4], -1, p)
        outs.append(e.scene_get_maps(4))
    for a, b, w in zip(outs[0], outs[1], ("depth", "normal", "conf")):
        _same(a, b, f"determinism {w}")
    d, n, c = outs[0]
    m = d > 0
    gt = sc.gt_depth[4]
    rel = np.abs(d[m] - gt[m]) / gt[m]
    assert m.mean() > 0.9 and np.median(rel) < 1e-3 and (rel < 0.01).mean() > 0.95
    assert np.allclose(np.linalg.norm(n[m], axis=-1), 1, atol=1e-4)
    assert (c[m] > 0).all() and (c <= 1).all() and (d[~m] == 0).all()
    assert (d[:4] == 0).all() and (d[:, :4] == 0).all()       # 4-px border never processed
    e.close()


@pytest.mark.parametrize("kernel", ["sweep2", "speculative"])
def test_tiled_sweeps_equal_the_tiled_oracle(nine_scene, small_scene, kernel, tiles=((24, 16), (9, 40))):
    """The opt-in tiled sweeps (pmhip_set_sweep_tiles; NOT the reference's estimator): sweeps run inside tiles, a neighbour across a tile border is read as the previous sweep
    left it.  Deterministic by construction, so the device must equal the oracle's restatement of the same definition (Opt::tileW / tileH) bit for bit: tiles that do not divide
    the map, tiles larger than a coarse level, 8 and 4 sources, the pyramid, a geometric round of a scene batch -- and 0 x 0 switches back to the reference's sweep."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    e = PatchMatchHIP(0)
    e.tuning(wideMaxViews=-1 if kernel == "sweep2" else 64, wideHyps=-1)
    for (tw, th) in tiles:
        e.set_sweep_tiles(tw, th)
        sc = nine_scene
        e.Init(False)
        p = default_params(seed=21)
        ids = [4] + list(sc.neighbors[4])
        d, n, c = e.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[4], sc.dmax[4], params=p)
        od, on, oc = _oracle(sc, 4, 21, tileW=tw, tileH=th)
        _same(d, od, f"tiles {tw}x{th}: depth"); _same(n, on, "normal"); _same(c, oc, "conf")
        xd, _, _ = _oracle(sc, 4, 21)
        assert not np.array_equal(od, xd) and (od > 0).mean() > 0.7          # it IS another estimator: the sequential sweep's maps differ
        sc = small_scene                                                     # 4 sources, scene batch with a geometric round
        p = default_params(seed=5, nEstimationGeometricIters=1)
        e.Init(False); e.scene_load(sc, n_levels=2)
        allv = list(range(sc.n_views))
        e.scene_estimate(allv, -1, p)
        photo = [e.scene_get_maps(v) for v in allv]
        e.scene_commit_round(); e.Init(True)
        e.scene_estimate([1, 3], 0, p)
        for v in (1, 3):
            od, on, oc = _oracle(sc, v, 5, nEstimationGeometricIters=1, tileW=tw, tileH=th)
            _same(photo[v][0], od, f"tiles {tw}x{th}: photometric depth v{v}")
            gd, gn, gc = _oracle(sc, v, 5, geo_iter=0, depth=od, normal=on, src={u: photo[u][0] for u in allv}, nEstimationGeometricIters=1, tileW=tw, tileH=th)
            dd, nn, cc = e.scene_get_maps(v)
            _same(dd, gd, f"tiles {tw}x{th}: geometric depth v{v}"); _same(nn, gn, "normal"); _same(cc, gc, "conf")
    e.set_sweep_tiles(0, 0)
    e.Init(False)
    sc = nine_scene
    ids = [4] + list(sc.neighbors[4])
    d, n, c = e.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[4], sc.dmax[4], params=default_params(seed=21))
    od, on, oc = _oracle(sc, 4, 21)
    _same(d, od, "tiles off: depth"); _same(n, on, "normal"); _same(c, oc, "conf")
    with pytest.raises(Exception):
        e.set_sweep_tiles(4, 4)
    e.close()

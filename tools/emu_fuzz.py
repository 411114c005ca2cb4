"""Randomised differential test of the device code under the CPU emulator against the oracle (not part of the test-suite: run it for as long as you like).

    OPENMVS_AMD_TEST_EMULATOR=1 PMHIP_LIB=tests/cpp/hipemu/_build/libpmhip_emu.so SGMHIP_LIB=tests/cpp/hipemu/_build/libsgmhip_emu.so \
        python tools/emu_fuzz.py --minutes 30 --seed 1

Each trial draws a size, a number of source views, options, masks and ranges at random and checks bit-exact equality for: the estimator (photometric pass,
sometimes a geometric round, sometimes masked), the post-filters, the fusion, whole scenes whose views differ in size, SGM Match with both kernel mappings, and the resident tSGM loop.  A failure prints
the trial's seed (re-run with --only <seed>) and the harness goes on."""
import argparse, os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvs_amd import synth, sgm, tsgm
from openmvs_amd.patchmatch import PatchMatchHIP, default_params
from oracle import pyoracle as po
from tests import sgm_cases as sc_, fuse_cases as fc
from tests.tsgm_backends import OracleBackend


def same(a, b, what):
    if not np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8)):
        raise AssertionError("%s differs (%d of %d bytes)" % (what, int((np.asarray(a).view(np.uint8) != np.asarray(b).view(np.uint8)).sum()), np.asarray(a).nbytes))


def trial_estimator(r, eng):
    w, h = int(r.randint(28, 110)), int(r.randint(24, 90))
    nsrc = int(r.choice([1, 2, 3, 4, 5, 7, 8, 11, 16]))
    sc = synth.make_scene(nsrc + 1, w, h, n_src=nsrc, seed=int(r.randint(1 << 30)))
    ref = int(r.randint(nsrc + 1))
    ids = [ref] + list(sc.neighbors[ref][:nsrc])
    lv = int(r.randint(0, 3))
    while lv and (min(w, h) >> lv) < 12:
        lv -= 1
    kw = dict(nSubResolutionLevels=lv, nEstimationIters=int(r.randint(1, 4)), nRandomIters=int(r.randint(1, 8)))
    if r.rand() < 0.5:
        kw.update(fRandomSmoothBonus=float(r.uniform(0.7, 1.0)), fNCCThresholdKeep=float(r.uniform(0.5, 0.95)), fRandomDepthRatio=float(r.uniform(0.002, 0.02)))
    seed = int(r.randint(1 << 20))
    mask = None
    if r.rand() < 0.35:
        mask = (r.rand(h, w) > 0.2).astype(np.uint8) * 255; y, x = r.randint(h // 2), r.randint(w // 2); mask[y:y + h // 3, x:x + w // 3] = 0
    eng.Init(False)
    got = eng.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[ref], sc.dmax[ref], params=default_params(seed=seed, **kw), mask=mask)
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
    opt = po.default_opt(seed=seed, viewID=ref, **kw)
    want = po.estimate_depth_map(views, len(ids), float(sc.dmin[ref]), float(sc.dmax[ref]), opt) if mask is None else \
        po.estimate_depth_map_masked(views, len(ids), float(sc.dmin[ref]), float(sc.dmax[ref]), opt, mask, mask_mode=True)
    for a, b, t in zip(got, want, "dnc"):
        same(a, b, "estimator %s (w %d h %d nsrc %d %s mask %s)" % (t, w, h, nsrc, kw, mask is not None))
    if r.rand() < 0.4 and mask is None:
        src = {i: (want[0] * float(r.uniform(0.97, 1.03))).astype(np.float32) for i in ids[1:]}
        kw2 = dict(kw, nEstimationGeometricIters=1, fEstimationGeometricWeight=float(r.uniform(0.05, 0.4)))
        eng.Init(True)
        g = eng.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[ref], sc.dmax[ref], depth=want[0], normal=want[1], src_depths=src, nGeometricIter=0,
                                 params=default_params(seed=seed, **kw2))
        views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps=src)
        og = po.estimate_depth_map(views, len(ids), float(sc.dmin[ref]), float(sc.dmax[ref]), po.default_opt(seed=seed, viewID=ref, **kw2), geo_iter=0, depth=want[0], normal=want[1])
        for a, b, t in zip(g, og, "dnc"):
            same(a, b, "geometric %s" % t)
        eng.Init(False)


def trial_filters_and_fusion(r):
    nv = int(r.randint(3, 8)); w, h = int(r.randint(40, 100)), int(r.randint(32, 80))
    sc = synth.make_scene(nv, w, h, n_src=min(4, nv - 1), seed=int(r.randint(1 << 30)))
    maps = fc.make_maps(sc, seed=int(r.randint(1 << 20)))
    e = PatchMatchHIP(0)
    e.scene_load(sc, n_levels=0)
    d, n, c = maps
    allv = list(range(nv))

    def restore():
        for v in allv:
            e.scene_set_maps(v, d[v], n[v]); e.scene_set_conf(v, c[v]); e.scene_set_color(v, sc.bgr[v])
    restore()
    kw = dict(nMinViewsFuse=int(r.randint(1, 5)), fDepthDiffThreshold=float(r.choice([0.003, 0.01, 0.05])), fNormalDiffThreshold=float(r.choice([10.0, 25.0, 70.0])),
              bEstimateColor=bool(r.rand() < 0.5), bEstimateNormal=bool(r.rand() < 0.5))
    got = e.scene_fuse(po.fuse_order([len(x) for x in sc.neighbors]), **kw)
    ref = po.fuse_depth_maps(*maps, list(sc.bgr), sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], **kw)
    fc.same_cloud(got, ref, "fuse %s" % kw)
    size, th = int(r.choice([0, 3, 20, 100])), float(r.choice([0.004, 0.01, 0.05]))
    e.scene_remove_small_segments(allv, nSpeckleSize=size, fDepthDiffThreshold=th)
    for v in allv:
        for a, b, t in zip(e.scene_get_maps(v), po.remove_small_segments(d[v], n[v], c[v], nSpeckleSize=size, fDepthDiffThreshold=th), "dnc"):
            same(a, b, "segments %s v%d" % (t, v))
    restore()
    gap = int(r.choice([1, 3, 7, 15]))
    e.scene_gap_interpolation(allv, nIpolGapSize=gap, fDepthDiffThreshold=th)
    for v in allv:
        for a, b, t in zip(e.scene_get_maps(v), po.gap_interpolation(d[v], n[v], c[v], nIpolGapSize=gap, fDepthDiffThreshold=th), "dnc"):
            same(a, b, "gap %s v%d" % (t, v))
    restore()
    adj, mv = bool(r.rand() < 0.5), int(r.randint(1, 4))
    e.scene_filter(allv, bAdjust=adj, nMinViewsFilter=mv, fDepthDiffThreshold=th)
    dep = np.stack(d); cnf = np.stack(c)
    for v in allv:
        gd, gn, gc = e.scene_get_maps(v)
        rc, od, oc = po.filter_depth_map(dep, cnf, sc.K, sc.R, sc.C, v, list(sc.neighbors[v]), sc.dmin[v], sc.dmax[v], bAdjust=adj, nMinViewsFilter=mv, fDepthDiffThreshold=th)
        same(gd, od, "filter depth v%d" % v); same(gc, oc, "filter conf v%d" % v)
    e.close()


def trial_mixed_sizes(r):
    """A scene whose views differ in size (each rendered at its own scale with its own K): estimation of all views in one call (one sweep per size class), geometric round,
    per-map filters, cross-view filter and fusion -- every result against the oracle run with each map at its own size."""
    nv = int(r.randint(3, 6)); W, H = int(r.randint(48, 90)) // 4 * 4, int(r.randint(40, 72)) // 4 * 4
    nsrc = min(int(r.randint(2, 5)), nv - 1)
    scales = [(1, 1), (3, 4), (5, 4), (3, 2)]
    seed_sc = int(r.randint(1 << 30))
    scs = {}

    def scene(k):
        if k not in scs:
            a, b = scales[k]
            scs[k] = synth.make_scene(nv, W * a // b, H * a // b, n_src=nsrc, seed=seed_sc)
        return scs[k]
    pick = [0 if v == 0 else int(r.randint(len(scales))) for v in range(nv)]
    base = scene(0)
    gray = {v: scene(pick[v]).gray[v] for v in range(nv)}; K = {v: scene(pick[v]).K[v] for v in range(nv)}; bgr = [scene(pick[v]).bgr[v] for v in range(nv)]
    nbs = [[int(x) for x in base.neighbors[v]] for v in range(nv)]
    seed = int(r.randint(1 << 20)); lv = int(r.randint(0, 2))
    kw = dict(nSubResolutionLevels=lv, nEstimationIters=int(r.randint(1, 3)), nEstimationGeometricIters=1)
    p = default_params(seed=seed, **kw)
    e = PatchMatchHIP(0); e.Init(False)
    e.scene_load(base, n_levels=2)
    for v in range(nv):
        if pick[v]:
            e.scene_set_view_sized(v, gray[v], K[v], base.R[v], base.C[v], float(base.dmin[v]), float(base.dmax[v]), nbs[v])
    allv = list(range(nv))
    e.scene_estimate(allv, -1, p)
    cur = {}
    for v in allv:
        ids = [v] + nbs[v]
        views, keep = po.make_views(gray, K, base.R, base.C, ids)
        cur[v] = po.estimate_depth_map(views, len(ids), float(base.dmin[v]), float(base.dmax[v]), po.default_opt(seed=seed, viewID=v, **kw))
        for a, b, t in zip(e.scene_get_maps(v), cur[v], "dnc"):
            same(a, b, "mixed sizes %s, photometric %s v%d (%s)" % ([scales[k] for k in pick], t, v, kw))
    e.scene_commit_round(); e.Init(True); e.scene_estimate(allv, 0, p)
    geo = {}
    for v in allv:
        ids = [v] + nbs[v]
        src = {i: cur[i][0] for i in ids[1:]}; cams = {i: (K[i], base.R[i], base.C[i]) for i in ids[1:]}
        views, keep = po.make_views(gray, K, base.R, base.C, ids, depth_maps=src, depth_cams=cams)
        geo[v] = po.estimate_depth_map(views, len(ids), float(base.dmin[v]), float(base.dmax[v]), po.default_opt(seed=seed, viewID=v, **kw), geo_iter=0, depth=cur[v][0], normal=cur[v][1])
        for a, b, t in zip(e.scene_get_maps(v), geo[v], "dnc"):
            same(a, b, "mixed sizes, geometric %s v%d" % (t, v))
    size, th = int(r.choice([3, 20, 60])), float(r.choice([0.004, 0.01, 0.03]))
    e.scene_remove_small_segments(allv, size, th); e.scene_gap_interpolation(allv, 7, th)
    flt = {}
    for v in allv:
        flt[v] = po.gap_interpolation(*po.remove_small_segments(*geo[v], nSpeckleSize=size, fDepthDiffThreshold=th), nIpolGapSize=7, fDepthDiffThreshold=th)
        for a, b, t in zip(e.scene_get_maps(v), flt[v], "dnc"):
            same(a, b, "mixed sizes, speckle + gap %s v%d" % (t, v))
    for v in allv:
        e.scene_set_color(v, bgr[v])
    fkw = dict(nMinViewsFuse=int(r.randint(1, 4)), fDepthDiffThreshold=float(r.choice([0.01, 0.05])), bEstimateColor=bool(r.rand() < 0.5), bEstimateNormal=bool(r.rand() < 0.5))
    deps = [flt[v][0] for v in allv]; nrms = [flt[v][1] for v in allv]; cnfs = [flt[v][2] for v in allv]
    got = e.scene_fuse(po.fuse_order([len(x) for x in nbs]) if fkw["nMinViewsFuse"] >= 2 else allv, **fkw)
    ref = po.fuse_depth_maps(deps, nrms, cnfs, bgr, [K[v] for v in allv], base.R, base.C, nbs, **fkw)
    if fkw["nMinViewsFuse"] < 2:
        got = dict(got); ref = dict(ref); got["weights"] = None; ref["weights"] = None
    fc.same_cloud(got, ref, "mixed sizes, fuse %s" % fkw)
    adj = bool(r.rand() < 0.5)
    e.scene_filter(allv, bAdjust=adj, fDepthDiffThreshold=th)
    D = {v: flt[v][0] for v in allv}; Cf = {v: flt[v][2] for v in allv}
    for v in allv:
        rc, od, oc = po.filter_depth_map(D, Cf, K, base.R, base.C, v, nbs[v], float(base.dmin[v]), float(base.dmax[v]), bAdjust=adj, fDepthDiffThreshold=th)
        gd, gn, gc = e.scene_get_maps(v)
        if rc == 0:
            same(gd, od, "mixed sizes, filter depth v%d" % v); same(gc, oc, "mixed sizes, filter conf v%d" % v)
    e.close()


def trial_sgm(r, m):
    w, h = int(r.randint(12, 240)), int(r.randint(10, 120))
    kind = str(r.choice(["uniform", "ragged", "ragged", "holes"]))
    if kind == "holes" and (w < 120 or h < 100):
        kind = "ragged"
    lo = int(r.randint(-20, 5)); hi = lo + int(r.choice([4, 6, 9, 17, 33, 70, 130]))
    lb, lg, rg = sc_.stereo_pair(w, h, int(r.randint(0, 8)), seed=int(r.randint(1 << 20)))
    px, n, mx = sc_.ranges(w, h, kind, lo, hi, seed=int(r.randint(1 << 20)))
    if n == 0:
        return
    od, oc, ocosts, oacc = po.sgm_match(lb, lg, rg, px, n, mx, m.P1, m.P2s)
    for sub in (0, int(r.choice([8, 16, 32]))):
        m.set_sub_group_kernels(sub)
        try:
            m.set_problem(lb, lg, rg, px, n, mx); m.Match()
            dd, c, costs, acc = m.results(volumes=True)
        finally:
            m.set_sub_group_kernels(False)
        t = "sgm %s %dx%d [%d,%d) sub=%s" % (kind, w, h, lo, hi, sub)
        same(costs, ocosts, t + " costs"); same(acc, oacc, t + " sums"); same(dd, od, t + " disp"); same(c, oc, t + " cost")


def trial_tsgm(r, m):
    k = int(r.choice([1, 2])); f = 1 << k
    w, h = int(r.randint(14, 40)) * f * 2, int(r.randint(12, 30)) * f * 2
    d0 = int(r.randint(2, 10))
    lb, lg, rg = sc_.stereo_pair(w, h, d0, seed=int(r.randint(1 << 20)))
    rb = np.roll(lb, d0, axis=1)
    mask = np.full((h, w), 255, np.uint8)
    if r.rand() < 0.6:
        mask[:, :int(r.randint(1, 9))] = 0; y, x = r.randint(h // 2), r.randint(w // 2); mask[y:y + h // 5, x:x + w // 4] = 0
    mr = max(w, h) >> k
    if tsgm.compute_scale(w, h, mr) != k:
        return
    kw = dict(n_speckle_size=int(r.choice([0, 20, 100])), subpixel_mode=int(r.randint(0, 7)), subpixel_steps=int(r.choice([1, 2, 4, 8])))
    dev = m.tsgm_match(lb, rb, lg, rg, mask, mask, min_resolution=mr, **kw)
    ref = tsgm.tsgm_match(OracleBackend(), lb, lg, rb, rg, mask, mask, min_resolution=mr, **kw)
    same(dev[0], ref[0], "tsgm disparity %dx%d k%d %s" % (w, h, k, kw)); same(dev[1], ref[1], "tsgm cost")


def trial_sgm_steps(r, m):
    """The step-by-step tSGM functions on random sizes and seeds: the bodies of the device tests are size-generic."""
    from tests import test_gpu_sgm_post as g
    w, h, seed = int(r.randint(14, 160)), int(r.randint(12, 100)), int(r.randint(1 << 16))
    g.test_map_steps_match_the_oracle(m, w, h, seed)
    g.test_range_map_matches_the_oracle(m, w, h, seed)
    g.test_disparity_depth_conversions_match_the_oracle(m, w, h, seed)
    g.test_projection_and_pair_fusion_match_the_oracle(m, w, h, seed)
    g.test_resident_fuse_equals_the_stepwise_fuse(m, w, h, seed)
    d = g.pc.smooth_disparity(w, h, seed)
    rr = np.random.RandomState(seed + 9)
    noisy = d.copy(); o = rr.rand(h, w) < 0.08; noisy[o] = rr.randint(-60, 60, int(o.sum())).astype(np.int16)
    mx, df = int(r.choice([0, 10, 100, 5000])), int(r.choice([0, 1, 5, 50]))
    same(m.FilterSpeckles(noisy, mx, df), po.sgm_filter_speckles(noisy, mx, df), "speckles %dx%d %d %d" % (w, h, mx, df))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--minutes", type=float, default=10); ap.add_argument("--seed", type=int, default=1); ap.add_argument("--only", type=int); ap.add_argument("--kind", type=str, default=None)
    a = ap.parse_args()
    eng = PatchMatchHIP(0); m = sgm.SemiGlobalMatcherHIP(0)
    t0 = time.time(); n = fails = 0
    kinds = [("estimator", lambda r: trial_estimator(r, eng)), ("filters+fusion", trial_filters_and_fusion), ("mixed sizes", trial_mixed_sizes), ("sgm", lambda r: trial_sgm(r, m)),
             ("tsgm", lambda r: trial_tsgm(r, m)), ("sgm steps", lambda r: trial_sgm_steps(r, m))]
    if a.kind:
        kinds = [k for k in kinds if k[0] == a.kind]
    counts = {k: 0 for k, _ in kinds}
    s = a.seed * 1000003
    while time.time() - t0 < a.minutes * 60:
        ts = a.only if a.only is not None else s + n
        r = np.random.RandomState(ts % (1 << 32))
        name, fn = kinds[int(r.randint(len(kinds)))]
        try:
            fn(r); counts[name] += 1
        except Exception as ex:
            fails += 1
            print("FAIL trial seed %d (%s): %s" % (ts, name, ex), flush=True)
            if not isinstance(ex, AssertionError):
                traceback.print_exc()
        n += 1
        if a.only is not None:
            break
        if n % 20 == 0:
            print("%d trials, %d failures, %.0f s  %s" % (n, fails, time.time() - t0, counts), flush=True)
    print("done: %d trials, %d failures, %s" % (n, fails, counts), flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())

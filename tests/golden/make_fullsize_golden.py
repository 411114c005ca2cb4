"""Full-size golden digests from the SEQUENTIAL CPU oracle at BASELINE.json's own sizes (CPU minutes, no GPU):

    python tests/golden/make_fullsize_golden.py pm      # config 2: 9-view 1920x1080 scene, every view 1 ref x 8 src,
                                                         #           photometric pass + 2 geometric rounds  (~10 min on 8 cores)
    python tests/golden/make_fullsize_golden.py sgm     # config 4: 2048x1536, D = 64, D = 128, ragged D <= 64   (~10 min)
    python tests/golden/make_fullsize_golden.py c5      # config 5's resolution: 5-view 3840x2160 scene, photometric pass + 1 geometric round,
                                                         #           speckle / gap filters, FilterDepthMap, FuseDepthMaps   (~25 min on 8 cores)

The maps themselves are too large to commit (27 x 41 MB), so the files hold, per view and round, the SHA-256 of the depth, normal and
confidence maps, per-row CRC-32s of the reference view's maps (a mismatch on the device then names the rows), the count of valid pixels and
a strided sample of the depth map.  Inputs are NOT stored: the scene is rebuilt by `synth.make_scene(..., exact=True)` (correctly rounded IEEE
operations only, the same bits on any host and on the GPU) and its SHA-256 is stored here, so that a test first proves it has the same inputs.

Reference semantics restated by the oracle: SceneDensify.cpp:616-805 (EstimateDepthMap), SemiGlobalMatcher.cpp:863-1302 (Match).
These digests pin the oracle (and through the -m gpu tests the HIP engine) at full size; the oracle itself is pinned to the reference's own code
by tests/test_ref_pinning.py (oracle/_ref).
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

PM_CASE = dict(n_views=9, width=1920, height=1080, n_src=8, seed=1, geo_iters=2, ref=4)
C5_CASE = dict(n_views=5, width=3840, height=2160, n_src=4, seed=3, geo_iters=1, ref=2)
SGM_CASE = dict(width=2048, height=1536, shift=21, seed=9, cases=[["uniform", 0, 64], ["uniform", 0, 128], ["ragged", 0, 64]])


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def row_crcs(a):
    a = np.ascontiguousarray(a)
    return [zlib.crc32(a[y].tobytes()) for y in range(a.shape[0])]


def digest_maps(d, n, c, rows=False):
    out = {"depth": sha(d), "normal": sha(n), "conf": sha(c), "valid": int((d > 0).sum())}
    if rows:
        out["depth_rows"] = row_crcs(d); out["normal_rows"] = row_crcs(n); out["conf_rows"] = row_crcs(c)
        out["depth_sample_step"] = 40
        out["depth_sample"] = [float(x) for x in d[::40, ::40].ravel()]
    return out


_SC = None
_MAPS = None


def _pm_task(args):
    from oracle import pyoracle as po
    v, rnd = args
    sc = _SC
    ids = [v] + list(sc.neighbors[v])
    opt = po.default_opt(seed=PM_CASE["seed"], viewID=v, nThreads=1, nEstimationGeometricIters=PM_CASE["geo_iters"])
    t = time.time()
    if rnd == 0:
        views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
        out = po.estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), opt)
    else:
        prev = {u: _MAPS[u][0] for u in range(sc.n_views)}
        views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps=prev)
        out = po.estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), opt, geo_iter=rnd - 1, depth=_MAPS[v][0], normal=_MAPS[v][1])
    return v, out, time.time() - t


def make_pm():
    global _SC, _MAPS
    from openmvs_amd import synth
    c = PM_CASE
    _SC = synth.make_scene(c["n_views"], c["width"], c["height"], n_src=c["n_src"], gray_only=True, exact=True)
    gold = dict(case=c, inputs=dict(gray=sha(_SC.gray), K=sha(_SC.K), R=sha(_SC.R), C=sha(_SC.C), neighbors=sha(_SC.neighbors.astype(np.int32)),
                                    dmin=[float(x) for x in _SC.dmin], dmax=[float(x) for x in _SC.dmax], diameter=_SC.diameter),
                rounds=[], oracle_seconds=[])
    for rnd in range(1 + c["geo_iters"]):
        with mp.get_context("fork").Pool(min(c["n_views"], os.cpu_count() or 1)) as pool:   # forked: _SC / _MAPS are inherited
            res = pool.map(_pm_task, [(v, rnd) for v in range(c["n_views"])], chunksize=1)
        _MAPS = {v: out for v, out, _ in res}
        gold["rounds"].append({str(v): digest_maps(*_MAPS[v], rows=(v == c["ref"])) for v in range(c["n_views"])})
        gold["oracle_seconds"].append(round(max(t for _, _, t in res), 1))
        print("round", rnd, "done:", gold["oracle_seconds"][-1], "s per view (sequential oracle); valid in ref:", gold["rounds"][-1][str(c["ref"])]["valid"], flush=True)
    json.dump(gold, open(os.path.join(HERE, "pm_config2_1920x1080.json"), "w"), separators=(",", ":"))


def _c5_task(args):
    from oracle import pyoracle as po
    v, rnd = args
    sc = _SC; c = C5_CASE
    ids = [v] + list(sc.neighbors[v])
    opt = po.default_opt(seed=c["seed"], viewID=v, nThreads=1, nEstimationGeometricIters=c["geo_iters"])
    t = time.time()
    if rnd == 0:
        views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
        out = po.estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), opt)
    elif rnd == 1:
        prev = {u: _MAPS[u][0] for u in range(sc.n_views)}
        views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps=prev)
        out = po.estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), opt, geo_iter=0, depth=_MAPS[v][0], normal=_MAPS[v][1])
    elif rnd == 2:     # per-map post-filters after the last round (SceneDensify.cpp:2069-2093): RemoveSmallSegments, then GapInterpolation
        out = po.gap_interpolation(*po.remove_small_segments(*_MAPS[v]))
    else:              # FilterDepthMap against the neighbours' unfiltered maps (SceneDensify.cpp:1955-1980)
        dep = [_MAPS[u][0] for u in range(sc.n_views)]; cnf = [_MAPS[u][2] for u in range(sc.n_views)]
        rc, fd, fc = po.filter_depth_map(dep, cnf, sc.K, sc.R, sc.C, v, list(sc.neighbors[v]), sc.dmin[v], sc.dmax[v])
        assert rc == 0
        out = (fd, _MAPS[v][1], fc)
    return v, out, time.time() - t


def make_c5():
    """Config 5's resolution through the whole resident chain of densify.compute_depth_maps + scene_fuse (tests/test_gpu_patchmatch.py::
    test_config5_resolution_estimate_filter_fuse): SceneDensify.cpp:616-805 (estimate), :1049-1299 (speckles, gaps), :809-1047 (FilterDepthMap), :1303-1646 (fuse)."""
    global _SC, _MAPS
    from openmvs_amd import synth
    from oracle import pyoracle as po
    c = C5_CASE
    _SC = synth.make_scene(c["n_views"], c["width"], c["height"], n_src=c["n_src"], exact=True)
    sc = _SC
    gold = dict(case=c, inputs=dict(gray=sha(sc.gray), bgr=sha(np.stack([np.asarray(b) for b in sc.bgr])), K=sha(sc.K), R=sha(sc.R), C=sha(sc.C),
                                    neighbors=sha(np.asarray(sc.neighbors).astype(np.int32)), diameter=sc.diameter), stages=[], oracle_seconds=[])
    names = ["photometric", "geometric 0", "speckles + gaps", "FilterDepthMap"]
    for rnd in range(4):
        with mp.get_context("fork").Pool(min(c["n_views"], os.cpu_count() or 1)) as pool:
            res = pool.map(_c5_task, [(v, rnd) for v in range(c["n_views"])], chunksize=1)
        _MAPS = {v: out for v, out, _ in res}
        gold["stages"].append({"name": names[rnd], **{str(v): digest_maps(*_MAPS[v], rows=(v == c["ref"])) for v in range(c["n_views"])}})
        gold["oracle_seconds"].append(round(max(t for _, _, t in res), 1))
        print(names[rnd], "done:", gold["oracle_seconds"][-1], "s (slowest view, sequential oracle); valid in ref:", gold["stages"][-1][str(c["ref"])]["valid"], flush=True)
    allv = list(range(c["n_views"]))
    t = time.time()
    cl = po.fuse_depth_maps([_MAPS[v][0] for v in allv], [_MAPS[v][1] for v in allv], [_MAPS[v][2] for v in allv], [sc.bgr[v] for v in allv],
                            sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], order=po.fuse_order([len(sc.neighbors[v]) for v in allv]))
    gold["fuse"] = dict(nPoints=int(cl["nPoints"]), nDepths=int(cl["nDepths"]), **{k: sha(cl[k]) for k in ("points", "viewStart", "views", "weights", "projs", "colors", "normals")},
                        points_sample=[float(x) for x in cl["points"][::100003].ravel()], oracle_seconds=round(time.time() - t, 1))
    print("fuse done:", gold["fuse"]["nPoints"], "points,", gold["fuse"]["oracle_seconds"], "s", flush=True)
    json.dump(gold, open(os.path.join(HERE, "pm_config5_3840x2160.json"), "w"), separators=(",", ":"))


def _sgm_task(case):
    from oracle import pyoracle as po
    from tests import sgm_cases as scs
    c = SGM_CASE
    kind, lo, hi = case
    lb, lg, rg = scs.stereo_pair(c["width"], c["height"], c["shift"], seed=c["seed"])
    px, n, mx = scs.ranges(c["width"], c["height"], kind, lo, hi)
    P2s = po.sgm_generate_p2s()
    t = time.time()
    d, cst, costs, acc = po.sgm_match(lb, lg, rg, px, n, mx, 3, P2s)
    dt = time.time() - t
    out = dict(kind=kind, lo=lo, hi=hi, num_costs=int(n), max_num_disp=int(mx), inputs=dict(left_bgr=sha(lb), left_gray=sha(lg), right_gray=sha(rg), pixels=sha(px)),
               disparity=sha(d), cost=sha(cst), costs=sha(costs), accums=sha(acc), disparity_rows=row_crcs(d), cost_rows=row_crcs(cst), oracle_seconds=round(dt, 1),
               costs_sum=int(costs.astype(np.uint64).sum()), accums_sum=int(acc.astype(np.uint64).sum()))
    print(kind, lo, hi, "done in %.1f s" % dt, flush=True)
    return out


def make_sgm():
    with mp.get_context("fork").Pool(3) as pool:
        res = pool.map(_sgm_task, SGM_CASE["cases"], chunksize=1)
    json.dump(dict(case=SGM_CASE, results=res), open(os.path.join(HERE, "sgm_config4_2048x1536.json"), "w"), separators=(",", ":"))


if __name__ == "__main__":
    from oracle import pyoracle
    pyoracle.build()
    {"pm": make_pm, "sgm": make_sgm, "c5": make_c5}[sys.argv[1]]()

"""ctypes binding of oracle/_ref -- the reference's own estimator sources compiled against oracle/ref/shim (see oracle/ref/build_ref.py).

TEST INFRASTRUCTURE ONLY.  Used by tests/test_ref_pinning.py to pin oracle/pm_oracle.cpp to the reference's code: same arrays in, compared bit
for bit.  The libraries are built in the build container (where /root/reference exists) and travel to the GPU box as files."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from oracle import pyoracle as po

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def path(kind: str = "pm_math") -> str:
    return os.path.join(_HERE, "_ref", "libref_pm.so" if kind == "pm_math" else "libref_pm_libm.so")


def available(kind: str = "pm_math") -> bool:
    if os.path.exists(path(kind)):
        return True
    try:
        from oracle.ref import build_ref
        build_ref.build()
    except Exception:
        return False
    return os.path.exists(path(kind))


def lib(kind: str = "pm_math") -> C.CDLL:
    if kind not in _LIBS:
        if not available(kind):
            raise RuntimeError("oracle/_ref is not built and /root/reference is not here to build it from")
        l = C.CDLL(path(kind))
        l.ref_run_level.restype = C.c_int
        l.ref_math_kind.restype = C.c_char_p
        _LIBS[kind] = l
    return _LIBS[kind]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _run(fn, views, n_views, depth, normal, conf, prior, dmin, dmax, opt, do_init, iter_begin, iter_end, th_end, mask):
    h, w = views[0].h, views[0].w
    depth = np.ascontiguousarray(depth, np.float32).copy(); normal = np.ascontiguousarray(normal, np.float32).copy(); conf = np.ascontiguousarray(conf, np.float32).copy()
    assert depth.shape == (h, w) and normal.shape == (h, w, 3) and conf.shape == (h, w)
    pr = None if prior is None else np.ascontiguousarray(prior, np.float32)
    mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    rc = fn(views, C.c_int(n_views), _fp(depth), _fp(normal), _fp(conf), _fp(pr) if pr is not None else None, C.c_float(dmin), C.c_float(dmax), C.byref(opt),
            C.c_int(1 if do_init else 0), C.c_uint(iter_begin), C.c_uint(iter_end), C.c_float(th_end), mk.ctypes.data_as(C.POINTER(C.c_ubyte)) if mk is not None else None)
    if rc:
        raise RuntimeError("run_level failed: %d" % rc)
    return depth, normal, conf


def ref_run_level(views, n_views, depth, normal, conf, dmin, dmax, opt, do_init=True, iter_begin=0, iter_end=0, th_end=-1.0, prior=None, mask=None, kind="pm_math"):
    """One pyramid level through the reference's own code (oracle/ref/ref_harness.cpp)."""
    return _run(lib(kind).ref_run_level, views, n_views, depth, normal, conf, prior, dmin, dmax, opt, do_init, iter_begin, iter_end, th_end, mask)


def orc_run_level(views, n_views, depth, normal, conf, dmin, dmax, opt, do_init=True, iter_begin=0, iter_end=0, th_end=-1.0, prior=None, mask=None):
    """The same unit through oracle/pm_oracle.cpp."""
    l = po.lib(); l.orc_run_level.restype = C.c_int
    return _run(l.orc_run_level, views, n_views, depth, normal, conf, prior, dmin, dmax, opt, do_init, iter_begin, iter_end, th_end, mask)


# ---- DepthMapsData::RemoveSmallSegments / GapInterpolation through the reference's own code (SceneDensify.cpp:809-1045, cut verbatim) ----
def _filter(name, depth, normal, conf, arg, th):
    d = np.ascontiguousarray(depth, np.float32).copy(); n = np.ascontiguousarray(normal, np.float32).copy(); c = np.ascontiguousarray(conf, np.float32).copy()
    h, w = d.shape
    assert n.shape == (h, w, 3) and c.shape == (h, w)
    fn = getattr(lib(), name); fn.restype = None
    fn(_fp(d), _fp(n), _fp(c), C.c_int(w), C.c_int(h), C.c_uint(arg), C.c_float(th))
    return d, n, c


def ref_remove_small_segments(depth, normal, conf, nSpeckleSize=100, fDepthDiffThreshold=0.01):
    return _filter("ref_remove_small_segments", depth, normal, conf, nSpeckleSize, fDepthDiffThreshold)


def ref_gap_interpolation(depth, normal, conf, nIpolGapSize=7, fDepthDiffThreshold=0.01):
    return _filter("ref_gap_interpolation", depth, normal, conf, nIpolGapSize, fDepthDiffThreshold)


def ref_filter_depth_map(depths, confs, K, R, Cc, ref, nbs, dmin, dmax, bAdjust=True, nMinViewsFilter=2, nMinViewsFilterAdjust=1, nCalibratedImages=None,
                         fDepthDiffThreshold=0.01):
    """DepthMapsData::FilterDepthMap (SceneDensify.cpp:1049-1299, cut verbatim) of view `ref` against neighbour views `nbs`; same arguments and return value as
    oracle.pyoracle.filter_depth_map."""
    keep = []

    def mk(i):
        v = po.FltView(); d = np.ascontiguousarray(depths[i], np.float32); c = np.ascontiguousarray(confs[i], np.float32); keep.extend([d, c])
        v.depth = _fp(d); v.conf = _fp(c); v.h, v.w = d.shape
        v.K[:] = np.asarray(K[i], np.float64).ravel(); v.R[:] = np.asarray(R[i], np.float64).ravel(); v.C[:] = np.asarray(Cc[i], np.float64).ravel()
        return v
    rv = mk(ref); arr = (po.FltView * max(1, len(nbs)))(*[mk(i) for i in nbs])
    h, w = depths[ref].shape
    nd = np.zeros((h, w), np.float32); nc = np.zeros((h, w), np.float32)
    fn = lib().ref_filter_depth_map; fn.restype = C.c_int
    rc = fn(C.byref(rv), arr, C.c_int(len(nbs)), C.c_int(w), C.c_int(h), C.c_float(dmin), C.c_float(dmax), C.c_int(1 if bAdjust else 0),
            C.c_uint(nMinViewsFilter), C.c_uint(nMinViewsFilterAdjust), C.c_uint(nCalibratedImages or len(depths)), C.c_float(fDepthDiffThreshold), _fp(nd), _fp(nc))
    return rc, nd, nc


# ---- SemiGlobalMatcher::Match through the reference's own code (oracle/ref/ref_sgm_harness.cpp) ----------------------------------------------
def sgm_available() -> bool:
    return available() and os.path.exists(os.path.join(_HERE, "_ref", "libref_sgm.so"))


def _sgm_lib():
    if "sgm" not in _LIBS:
        available()
        _LIBS["sgm"] = C.CDLL(os.path.join(_HERE, "_ref", "libref_sgm.so"))
    return _LIBS["sgm"]


def ref_sgm_generate_p2s(P2=4, alpha=14.0, beta=38.0):
    out = np.zeros(256, np.uint16)
    _sgm_lib().ref_sgm_generate_p2s(C.c_uint16(P2), C.c_float(alpha), C.c_float(beta), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def ref_sgm_match(left_bgr, left_gray, right_gray, pixels, num_costs, max_num_disp, P1, P2s):
    """Same arguments and results as pyoracle.sgm_match."""
    lb = np.ascontiguousarray(left_bgr, np.uint8); lg = np.ascontiguousarray(left_gray, np.float32); rg = np.ascontiguousarray(right_gray, np.float32)
    h, w = lg.shape
    px = np.ascontiguousarray(pixels); p2 = np.ascontiguousarray(P2s, np.uint16)
    d = np.zeros((h - 6, w - 6), np.int16); c = np.zeros((h - 6, w - 6), np.uint16)
    costs = np.zeros(num_costs, np.uint8); acc = np.zeros(num_costs, np.uint16)
    l = _sgm_lib(); l.ref_sgm_match.restype = C.c_int
    rc = l.ref_sgm_match(lb.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(lg), _fp(rg), C.c_int(w), C.c_int(h), px.ctypes.data_as(C.c_void_p),
                         C.c_uint64(num_costs), C.c_int(max_num_disp), C.c_uint16(P1), p2.ctypes.data_as(C.POINTER(C.c_uint16)),
                         d.ctypes.data_as(C.POINTER(C.c_int16)), c.ctypes.data_as(C.POINTER(C.c_uint16)),
                         costs.ctypes.data_as(C.POINTER(C.c_uint8)), acc.ctypes.data_as(C.POINTER(C.c_uint16)))
    assert rc == 0
    return d, c, costs, acc


def sgm_post_lib():
    """libref_sgm.so as the `impl` of oracle.pyoracle's sgm_* step wrappers (prefix "ref_sgm_"): the reference's ConsistencyCrossCheck, FilterByCost, ExtractMask, UpscaleMask
    and FlipDirection (SemiGlobalMatcher.cpp:1446-1691, cut verbatim) behind the oracle functions' own signatures."""
    return _sgm_lib()


def ref_sgm_refine(disp, pixels, accums, mode=6, steps=4):
    """SemiGlobalMatcher::RefineDisparityMap (SemiGlobalMatcher.cpp:1693-1811) on a valid-grid disparity map with the pixel table and 8-path sums of the Match that produced it."""
    a = np.ascontiguousarray(disp, np.int16).copy(); px = np.ascontiguousarray(pixels); ac = np.ascontiguousarray(accums, np.uint16)
    vh, vw = a.shape
    fn = _sgm_lib().ref_sgm_refine_wh; fn.restype = None
    fn(a.ctypes.data_as(C.POINTER(C.c_int16)), px.ctypes.data_as(C.c_void_p), ac.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_uint64(ac.size), C.c_int(vw), C.c_int(vh), C.c_int(mode), C.c_int(steps))
    return a


# ---- DepthMapsData::EstimateDepthMap + ScaleDepthData through the reference's own code (oracle/ref/ref_driver_harness.cpp: SceneDensify.cpp:578-601, :616-805 verbatim) ----
def driver_path(kind: str = "pm_math") -> str:
    return os.path.join(_HERE, "_ref", "libref_driver.so" if kind == "pm_math" else "libref_driver_libm.so")


def driver_available(kind: str = "pm_math") -> bool:
    if os.path.exists(driver_path(kind)):
        return True
    available()
    return os.path.exists(driver_path(kind))


def _driver(kind="pm_math"):
    key = "driver_" + kind
    if key not in _LIBS:
        if not driver_available(kind):
            raise RuntimeError("oracle/_ref/libref_driver*.so is not built and /root/reference is not here to build it from")
        po.lib()                                           # libpm_oracle.so (the cv::resize stand-in's resamplers) must exist before the dependent library is loaded
        l = C.CDLL(driver_path(kind))
        l.ref_estimate_depth_map.restype = C.c_int
        l.ref_scale_view.restype = C.c_int
        _LIBS[key] = l
    return _LIBS[key]


def ref_estimate_depth_map(views, n_views, dmin, dmax, opt, geo_iter=-1, depth=None, normal=None, mask=None, mask_mode=False, kind="pm_math"):
    """One DepthMapsData::EstimateDepthMap call in the reference's own text (level loop, hand-off, thresholds, threads = opt.nThreads).  Same arguments and results as
    pyoracle.estimate_depth_map / estimate_depth_map_masked.  The estimators draw from the reference's std::mt19937 (non-release seeding): compare with the oracle at rngMode 2."""
    h, w = views[0].h, views[0].w
    depth = np.zeros((h, w), np.float32) if depth is None else np.ascontiguousarray(depth, np.float32).copy()
    normal = np.zeros((h, w, 3), np.float32) if normal is None else np.ascontiguousarray(normal, np.float32).copy()
    conf = np.zeros((h, w), np.float32)
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask) != 0, np.uint8)
    rc = _driver(kind).ref_estimate_depth_map(views, C.c_int(n_views), _fp(depth), _fp(normal), _fp(conf), C.c_float(dmin), C.c_float(dmax), C.byref(opt), C.c_int(geo_iter),
                                              None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(1 if (mask_mode or m is not None) else 0))
    if rc:
        raise RuntimeError("ref_estimate_depth_map failed: %d" % rc)
    return depth, normal, conf


def ref_scale_view(view, f):
    """DepthMapsData::ScaleDepthData of one view by 1 / f: (image, K, depth map or None, Kd or None)."""
    h, w = view.h, view.w
    img = np.zeros((h, w), np.float32); dep = np.zeros((h, w), np.float32); K = np.zeros(9); Kd = np.zeros(9); wh = (C.c_int * 2)()
    rc = _driver().ref_scale_view(C.byref(view), C.c_int(f), _fp(img), K.ctypes.data_as(C.POINTER(C.c_double)), _fp(dep), Kd.ctypes.data_as(C.POINTER(C.c_double)), wh)
    assert rc == 0
    nw, nh = wh[0], wh[1]
    has = bool(view.depth)
    return img.ravel()[:nw * nh].reshape(nh, nw).copy(), K.reshape(3, 3), (dep.ravel()[:nw * nh].reshape(nh, nw).copy() if has else None), (Kd.reshape(3, 3) if has else None)


# ---- Scene::SelectNeighborViews / FilterNeighborViews through the reference's own code (oracle/ref/ref_scene_harness.cpp: Scene.cpp:801-934, :953-968 verbatim) ----
VIEW_SCORE = np.dtype([("ID", np.uint32), ("points", np.uint32), ("scale", np.float32), ("angle", np.float32), ("area", np.float32), ("score", np.float32)])


def scene_available() -> bool:
    available()
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_scene.so"))


def _scene_lib():
    if "scene" not in _LIBS:
        if not scene_available():
            raise RuntimeError("oracle/_ref/libref_scene.so is not built and /root/reference is not here to build it from")
        l = C.CDLL(os.path.join(_HERE, "_ref", "libref_scene.so"))
        l.ref_select_neighbor_views.restype = C.c_int; l.ref_filter_neighbor_views.restype = C.c_int
        _LIBS["scene"] = l
    return _LIBS["scene"]


def ref_select_neighbor_views(cams, sizes, points, point_views, ID, nMinViews=2, nMinPointViews=2, fOptimAngleDeg=12.0, nInsideROI=1):
    """cams: list of (K, R, C) at each image's working resolution; sizes: list of (w, h); points: [N,3] float32; point_views: list of ascending image-index arrays.
    Returns (ok, neighbours as VIEW_SCORE records in the reference's order, kept point indices, average depth)."""
    n = len(cams)
    cam = np.zeros((n, 21), np.float64)
    for i, (K, R, Cc) in enumerate(cams):
        cam[i, :9] = np.asarray(K, np.float64).ravel(); cam[i, 9:18] = np.asarray(R, np.float64).ravel(); cam[i, 18:] = np.asarray(Cc, np.float64).ravel()
    sz = np.ascontiguousarray(np.asarray(sizes, np.int32)); valid = np.ones(n, np.uint8)
    pts = np.ascontiguousarray(points, np.float32)
    start = np.zeros(len(point_views) + 1, np.uint32); start[1:] = np.cumsum([len(v) for v in point_views])
    flat = np.ascontiguousarray(np.concatenate([np.asarray(v, np.uint32) for v in point_views]) if len(point_views) else np.zeros(0, np.uint32))
    nb = np.zeros(max(1, n), VIEW_SCORE); keep = np.zeros(max(1, len(pts)), np.uint32)
    nn = C.c_int(0); npk = C.c_int(0); avg = C.c_float(0)
    ok = _scene_lib().ref_select_neighbor_views(C.c_int(n), cam.ctypes.data_as(C.POINTER(C.c_double)), sz.ctypes.data_as(C.POINTER(C.c_int)), valid.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                                C.c_int(len(pts)), pts.ctypes.data_as(C.POINTER(C.c_float)), start.ctypes.data_as(C.POINTER(C.c_uint32)), flat.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                C.c_uint32(ID), C.c_uint(nMinViews), C.c_uint(nMinPointViews), C.c_float(np.float32(np.deg2rad(np.float32(fOptimAngleDeg)))), C.c_uint(nInsideROI),
                                                nb.ctypes.data_as(C.c_void_p), C.c_int(len(nb)), C.byref(nn), keep.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int(len(keep)), C.byref(npk), C.byref(avg))
    return bool(ok), nb[:nn.value].copy(), keep[:npk.value].copy(), float(avg.value)


def ref_pixel_camera(K, Rc, Cc, cam_size, Rp, Cp, size):
    """The archive's platform camera (K, Rc, Cc, stored resolution cam_size or (0, 0) if already normalised) and an image's pose (Rp, Cp) -> that image's pixel camera
    at `size`: Scene::LoadInterface's normalisation, Platform::GetCamera and Camera::GetK in the reference's own text (libref_scene.so)."""
    lib = _scene_lib()
    d = lambda a: np.ascontiguousarray(np.asarray(a, np.float64).ravel())
    k, rc, cc, rp, cp = d(K), d(Rc), d(Cc), d(Rp), d(Cp)
    oK, oR, oC = np.zeros(9), np.zeros(9), np.zeros(3)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    lib.ref_pixel_camera.restype = None
    lib.ref_pixel_camera(p(k), p(rc), p(cc), C.c_uint32(int(cam_size[0])), C.c_uint32(int(cam_size[1])), p(rp), p(cp), C.c_uint32(int(size[0])), C.c_uint32(int(size[1])), p(oK), p(oR), p(oC))
    return oK.reshape(3, 3), oR.reshape(3, 3), oC


def ref_filter_neighbor_views(neighbors, fMinArea, fMinScale, fMaxScale, fMinAngle, fMaxAngle, nMaxViews):
    a = np.ascontiguousarray(neighbors, VIEW_SCORE).copy()
    n = _scene_lib().ref_filter_neighbor_views(a.ctypes.data_as(C.c_void_p), C.c_int(len(a)), C.c_float(fMinArea), C.c_float(fMinScale), C.c_float(fMaxScale), C.c_float(fMinAngle), C.c_float(fMaxAngle), C.c_uint(nMaxViews))
    return a[:n]


# ---- DepthMapsData::FuseDepthMaps / MergeDepthMaps through the reference's own code (oracle/ref/ref_fuse_harness.cpp: SceneDensify.cpp:1303-1646 verbatim) ----
def fuse_available() -> bool:
    available()
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_fuse.so"))


def _fuse_lib():
    if "fuse" not in _LIBS:
        if not fuse_available():
            raise RuntimeError("oracle/_ref/libref_fuse.so is not built and /root/reference is not here to build it from")
        l = C.CDLL(os.path.join(_HERE, "_ref", "libref_fuse.so"))
        l.ref_fuse_depth_maps.restype = C.c_int; l.ref_merge_depth_maps.restype = C.c_int
        _LIBS["fuse"] = l
    return _LIBS["fuse"]


def ref_fuse_depth_maps(depths, normals, confs, bgrs, K, R, Cc, neighbors, nMinViewsFuse=2, fDepthDiffThreshold=0.01, fNormalDiffThreshold=25.0,
                        bEstimateColor=True, bEstimateNormal=True):
    """Same inputs as pyoracle.fuse_depth_maps.  Returns (cloud dict, order): the cloud WITHOUT `projs` (the reference keeps them in a local) and the order in which the
    reference's own sort processed the images; nMinViewsFuse < 2 runs MergeDepthMaps (order = None)."""
    n = len(depths)
    first = next(d for d in depths if d is not None)
    h, w = first.shape
    keep = []
    arr = (po.OrcFuseView * n)()

    def ptr(a, dt, ct):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dt); keep.append(a)
        return a.ctypes.data_as(C.POINTER(ct))
    for i in range(n):
        v = arr[i]
        v.depth = ptr(depths[i], np.float32, C.c_float)
        if depths[i] is not None:
            v.h, v.w = np.asarray(depths[i]).shape[:2]
        v.normal = ptr(None if normals is None or depths[i] is None else normals[i], np.float32, C.c_float)
        v.conf = ptr(None if confs is None or depths[i] is None else confs[i], np.float32, C.c_float)
        v.bgr = ptr(None if bgrs is None else bgrs[i], np.uint8, C.c_uint8)
        v.K[:] = np.asarray(K[i], np.float64).ravel(); v.R[:] = np.asarray(R[i], np.float64).ravel(); v.C[:] = np.asarray(Cc[i], np.float64).ravel()
        nb = np.ascontiguousarray(neighbors[i], np.uint32); keep.append(nb)
        v.neighbors = nb.ctypes.data_as(C.POINTER(C.c_uint32)); v.nNeighbors = len(nb)
    out = po.OrcFuseCloud()
    L = _fuse_lib()
    if nMinViewsFuse < 2:
        rc = L.ref_merge_depth_maps(arr, C.c_int(n), C.c_int(w), C.c_int(h), C.c_int(1 if bEstimateColor else 0), C.c_int(1 if bEstimateNormal else 0), C.byref(out))
        return po._cloud(out, rc, L.ref_fuse_free), None
    order = np.zeros(n, np.uint32); no = C.c_int(0)
    rc = L.ref_fuse_depth_maps(arr, C.c_int(n), C.c_int(w), C.c_int(h), order.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(no), C.c_uint(nMinViewsFuse),
                               C.c_float(fDepthDiffThreshold), C.c_float(fNormalDiffThreshold), C.c_int(1 if bEstimateColor else 0), C.c_int(1 if bEstimateNormal else 0), C.byref(out))
    return po._cloud(out, rc, L.ref_fuse_free), [int(x) for x in order[:no.value]]


def ref_estimate_normal_map(K, depth):
    """MVS::EstimateNormalMap (libs/MVS/DepthMap.cpp:1522-1613, verbatim in oracle/_ref/libref_fuse.so) -> [h, w, 3] float32."""
    d = np.ascontiguousarray(depth, np.float32); h, w = d.shape
    Kf = np.ascontiguousarray(np.asarray(K, np.float64).astype(np.float32).ravel())
    out = np.zeros((h, w, 3), np.float32)
    lib = _fuse_lib()
    rc = lib.ref_estimate_normal_map(Kf.ctypes.data_as(C.POINTER(C.c_float)), d.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(w), C.c_int(h), out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != 0:
        raise RuntimeError("ref_estimate_normal_map: %d" % rc)
    return out


# ---- the two text files in front of the path through the reference's own readers (oracle/ref/ref_text_harness.cpp: SML.cpp, ConfigTable.cpp, the OPTDENSE list of
# DepthMap.cpp:50-115, Scene::LoadViewNeighbors / SaveViewNeighbors, Util::CommandLineToArgvA -- all verbatim) ----
def text_available() -> bool:
    available()
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_text.so"))


def _text_lib():
    if "text" not in _LIBS:
        if not text_available():
            raise RuntimeError("oracle/_ref/libref_text.so is not built and /root/reference is not here to build it from")
        _LIBS["text"] = C.CDLL(os.path.join(_HERE, "_ref", "libref_text.so"))
    return _LIBS["text"]


def ref_optdense_load(path: str, save_path: str | None = None):
    """OPTDENSE::init(); oConfig.Load(path); OPTDENSE::update() [; oConfig.Save(save_path)] -> (bValidConfig, the 46 variables in list order as float64)."""
    v = np.zeros(64, np.float64)
    n = _text_lib().ref_optdense_load(path.encode(), v.ctypes.data_as(C.POINTER(C.c_double)), 64, save_path.encode() if save_path else None)
    return n > 0, v[:abs(n)].copy()


def ref_load_view_neighbors(path: str, n_images: int, save_path: str | None = None):
    """Scene::LoadViewNeighbors on a scene of n_images images -> (neighbour ID lists per image, (points, scale, angle, area, score) of the first ViewScore made)."""
    counts = np.zeros(n_images, np.int32); ids = np.zeros(4096, np.uint32); first = np.zeros(5, np.float32)
    n = _text_lib().ref_load_view_neighbors(path.encode(), n_images, counts.ctypes.data_as(C.POINTER(C.c_int)), ids.ctypes.data_as(C.POINTER(C.c_uint32)), len(ids),
                                            first.ctypes.data_as(C.POINTER(C.c_float)), save_path.encode() if save_path else None)
    if n < 0:
        raise RuntimeError("ref_load_view_neighbors: %d" % n)
    out, o = [], 0
    for c in counts:
        out.append([int(x) for x in ids[o:o + c]]); o += c
    return out, first


def ref_sml_root(path: str):
    """(what SML::Load returned, {name: value} of the root entries as it files them -- unnamed entries are called Item<n>)."""
    got = {}
    CB = C.CFUNCTYPE(None, C.c_char_p, C.c_char_p, C.c_void_p)
    cb = CB(lambda name, value, ctx: got.__setitem__(name.decode("latin-1"), value.decode("latin-1")))
    n = _text_lib().ref_sml_root(path.encode(), cb, None)
    return n >= 0, got


def ref_split_words(line: str):
    buf = C.create_string_buffer(len(line) + 8)
    n = _text_lib().ref_split_words(line.encode("latin-1"), buf, len(buf))
    assert n >= 0
    words = buf.raw.split(b"\0")[:n]
    return [w.decode("latin-1") for w in words]


def ref_image_size(width, height, level, min_size, max_size):
    """TImage::computeMaxResolution then Image::ResizeImage (Types.inl:2457-2477, Image.cpp:139-155, verbatim in libref_text.so) -> (w, h, effective level, resolution, scale)."""
    w, h, lv, res = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint(); sc = C.c_float()
    _text_lib().ref_image_size(C.c_uint(width), C.c_uint(height), C.c_uint(level), C.c_uint(min_size), C.c_uint(max_size), C.byref(w), C.byref(h), C.byref(lv), C.byref(res), C.byref(sc))
    return w.value, h.value, lv.value, res.value, sc.value


def ref_to_gray_bgr(bgr, srgb=False):
    """TImage<Pixel8U>::toGray(COLOR_BGR2GRAY, bNormalize = true, bSRGB) with the reference's CONVERT helpers (Types.inl:1590-1659 verbatim) -> float32 image."""
    a = np.ascontiguousarray(bgr, np.uint8)
    out = np.zeros(a.shape[:-1], np.float32)
    _text_lib().ref_to_gray_bgr(a.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(out.size), C.c_int(1 if srgb else 0), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def ref_raster_faces(w, h, projs, z, normals, faces):
    """The face loop of TriangulatePoints2DepthMap with the reference's own rasteriser and functor (libref_fuse.so) -> (depth [h, w], normal [h, w, 3] or None)."""
    lib = _fuse_lib()
    pj = np.ascontiguousarray(projs, np.float32); zz = np.ascontiguousarray(z, np.float32); fc = np.ascontiguousarray(faces, np.uint32)
    nr = None if normals is None else np.ascontiguousarray(normals, np.float32)
    d = np.zeros((h, w), np.float32); n = np.zeros((h, w, 3), np.float32)
    fp = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))
    lib.ref_raster_faces.restype = None
    lib.ref_raster_faces(C.c_int(w), C.c_int(h), C.c_int(len(zz)), fp(pj), fp(zz), fp(nr), C.c_int(len(fc)), fc.ctypes.data_as(C.POINTER(C.c_uint32)), fp(d), fp(n) if nr is not None else None)
    return d, (n if nr is not None else None)


def ref_init_views_splat(K, R, Cc, size, X, pts):
    """The nMinViewsTrustPoint < 2 branch of DepthMapsData::InitViews (SceneDensify.cpp:418-451 verbatim in libref_fuse.so) -> (depth, normal, dMin, dMax)."""
    w, h = size
    d = np.zeros((h, w), np.float32); n = np.ones((h, w, 3), np.float32); mn = C.c_float(); mx = C.c_float()
    Xf = np.ascontiguousarray(X, np.float32); p = np.ascontiguousarray(pts, np.uint32)
    dd = lambda a: np.ascontiguousarray(np.asarray(a, np.float64).ravel()).ctypes.data_as(C.POINTER(C.c_double))
    k, r, c = (np.ascontiguousarray(np.asarray(a, np.float64).ravel()) for a in (K, R, Cc))
    lib = _fuse_lib(); lib.ref_init_views_splat.restype = None
    lib.ref_init_views_splat(k.ctypes.data_as(C.POINTER(C.c_double)), r.ctypes.data_as(C.POINTER(C.c_double)), c.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(w), C.c_int(h),
                             Xf.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(len(Xf)), p.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int(len(p)),
                             d.ctypes.data_as(C.POINTER(C.c_float)), n.ctypes.data_as(C.POINTER(C.c_float)), C.byref(mn), C.byref(mx))
    return d, n, mn.value, mx.value

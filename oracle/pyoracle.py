"""ctypes binding of the CPU oracle (oracle/libpm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (openmvs_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcView(C.Structure):
    _fields_ = [("image", C.POINTER(C.c_float)), ("w", C.c_int), ("h", C.c_int),
                ("K", C.c_double * 9), ("R", C.c_double * 9), ("C", C.c_double * 3),
                ("depth", C.POINTER(C.c_float)),
                ("Kd", C.c_double * 9), ("Rd", C.c_double * 9), ("Cd", C.c_double * 3), ("dw", C.c_int), ("dh", C.c_int)]


class OrcOpt(C.Structure):
    _fields_ = [("nSubResolutionLevels", C.c_uint32), ("nEstimationIters", C.c_uint32),
                ("nEstimationGeometricIters", C.c_uint32), ("nRandomIters", C.c_uint32),
                ("fEstimationGeometricWeight", C.c_float), ("fRandomDepthRatio", C.c_float),
                ("fRandomAngle1Range", C.c_float), ("fRandomAngle2Range", C.c_float),
                ("fRandomSmoothDepth", C.c_float), ("fRandomSmoothNormal", C.c_float),
                ("fRandomSmoothBonus", C.c_float), ("fNCCThresholdKeep", C.c_float),
                ("fDescriptorMinMagnitudeThreshold", C.c_float),
                ("seed", C.c_uint32), ("viewID", C.c_uint32), ("rngMode", C.c_int32), ("nThreads", C.c_int32),
                ("tileW", C.c_int32), ("tileH", C.c_int32)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libpm_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pm_oracle.cpp", "sgm_oracle.cpp", "filter_oracle.cpp", "fuse_oracle.cpp", "sgm_post_oracle.cpp")] + \
           [os.path.join(_HERE, "..", "openmvs_amd", "csrc", "pm_math.h")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s))
    if stale and all(os.path.exists(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpm_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_estimate_depth_map.restype = C.c_int
        _LIB.orc_score_pixel.restype = C.c_int
    return _LIB


def default_opt(**kw) -> OrcOpt:
    o = OrcOpt()
    lib().orc_default_opt(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_views(gray, K, R, Cc, ids, depth_maps=None, depth_cams=None):
    """ids[0] = reference view, ids[1:] = sources.  depth_maps: optional dict id -> [H,W] float32 (any size);
    depth_cams: optional dict id -> (Kd, Rd, Cd), the camera stored with that depth map (default: the view's own)."""
    keep = []
    arr = (OrcView * len(ids))()
    for n, i in enumerate(ids):
        img = np.ascontiguousarray(gray[i], np.float32); keep.append(img)
        v = arr[n]
        v.image = _fp(img); v.h, v.w = img.shape
        v.K[:] = np.asarray(K[i], np.float64).ravel(); v.R[:] = np.asarray(R[i], np.float64).ravel(); v.C[:] = np.asarray(Cc[i], np.float64).ravel()
        if n > 0 and depth_maps is not None:
            d = np.ascontiguousarray(depth_maps[i], np.float32); keep.append(d)
            v.depth = _fp(d); v.dh, v.dw = d.shape
            if depth_cams is not None and i in depth_cams:
                kd, rd, cd = depth_cams[i]
                v.Kd[:] = np.asarray(kd, np.float64).ravel(); v.Rd[:] = np.asarray(rd, np.float64).ravel(); v.Cd[:] = np.asarray(cd, np.float64).ravel()
            else:
                v.Kd[:] = v.K[:]; v.Rd[:] = v.R[:]; v.Cd[:] = v.C[:]
    return arr, keep


def estimate_depth_map(views, n_views, dmin, dmax, opt: OrcOpt, geo_iter: int = -1,
                       depth=None, normal=None, stages: bool = False):
    """One DepthMapsData::EstimateDepthMap call.  Returns (depth, normal, conf[, stage list])."""
    h, w = views[0].h, views[0].w
    depth = np.zeros((h, w), np.float32) if depth is None else np.ascontiguousarray(depth, np.float32).copy()
    normal = np.zeros((h, w, 3), np.float32) if normal is None else np.ascontiguousarray(normal, np.float32).copy()
    conf = np.zeros((h, w), np.float32)
    dump = None; cap = 0; used = C.c_size_t(0)
    if stages:
        cap = int((4 + h * w * 5) * 24)
        dump = np.zeros(cap, np.float32)
    rc = lib().orc_estimate_depth_map(views, C.c_int(n_views), _fp(depth), _fp(normal), _fp(conf),
                                      C.c_float(dmin), C.c_float(dmax), C.byref(opt), C.c_int(geo_iter),
                                      _fp(dump) if stages else None, C.c_size_t(cap), C.byref(used))
    if rc != 0:
        raise RuntimeError(f"orc_estimate_depth_map failed: {rc}")
    if not stages:
        return depth, normal, conf
    out = []; p = 0
    while p < used.value:
        lvl, it, sw, sh = (int(dump[p + k]) for k in range(4)); n = sw * sh
        out.append(dict(level=lvl, iter=it, depth=dump[p + 4:p + 4 + n].reshape(sh, sw).copy(),
                        normal=dump[p + 4 + n:p + 4 + 4 * n].reshape(sh, sw, 3).copy(),
                        cost=dump[p + 4 + 4 * n:p + 4 + 5 * n].reshape(sh, sw).copy()))
        p += 4 + 5 * n
    return depth, normal, conf, out


def estimate_depth_map_masked(views, n_views, dmin, dmax, opt: OrcOpt, mask, geo_iter: int = -1, depth=None, normal=None, mask_mode=True):
    """EstimateDepthMap with --ignore-mask-label: `mask` (h, w), 0 = ignored pixel, or None (then only the NEAREST hand-off of mask_mode)."""
    h, w = views[0].h, views[0].w
    depth = np.zeros((h, w), np.float32) if depth is None else np.ascontiguousarray(depth, np.float32).copy()
    normal = np.zeros((h, w, 3), np.float32) if normal is None else np.ascontiguousarray(normal, np.float32).copy()
    conf = np.zeros((h, w), np.float32)
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask) != 0, np.uint8)
    lib().orc_estimate_depth_map_masked.restype = C.c_int
    rc = lib().orc_estimate_depth_map_masked(views, C.c_int(n_views), _fp(depth), _fp(normal), _fp(conf), C.c_float(dmin), C.c_float(dmax),
                                             C.byref(opt), C.c_int(geo_iter), None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8)),
                                             C.c_int(1 if mask_mode else 0))
    if rc != 0:
        raise RuntimeError(f"orc_estimate_depth_map_masked failed: {rc}")
    return depth, normal, conf


def score_pixel(views, n_views, opt: OrcOpt, x, y, depth, normal, prior=None):
    sc = np.zeros(n_views - 1, np.float32); agg = C.c_float(0)
    nrm = np.ascontiguousarray(normal, np.float32)
    pr = None if prior is None else np.ascontiguousarray(prior, np.float32)
    rc = lib().orc_score_pixel(views, C.c_int(n_views), C.byref(opt), C.c_int(x), C.c_int(y), C.c_float(depth), _fp(nrm),
                               _fp(pr) if pr is not None else None, _fp(sc), C.byref(agg))
    return rc, sc, agg.value


def pixel_helpers(views, n_views, opt: OrcOpt, x, y, dmin, dmax, nx, ny, ndepth, nnormal, hyp_depth, hyp_normal):
    """(InterpolatePixel depth, CorrectNormal(nnormal), smoothness factor of the hypothesis plane w.r.t. the neighbour) at pixel (x, y)."""
    nn = np.ascontiguousarray(nnormal, np.float32); hn = np.ascontiguousarray(hyp_normal, np.float32)
    di = C.c_float(0); sf = C.c_float(0); cn = np.zeros(3, np.float32)
    lib().orc_pixel_helpers.restype = C.c_int
    rc = lib().orc_pixel_helpers(views, C.c_int(n_views), C.byref(opt), C.c_int(x), C.c_int(y), C.c_float(dmin), C.c_float(dmax), C.c_int(nx), C.c_int(ny),
                                 C.c_float(ndepth), _fp(nn), C.c_float(hyp_depth), _fp(hn), C.byref(di), _fp(cn), C.byref(sf))
    return rc, di.value, cn, sf.value


def zigzag(w, h, raw_stride=64):
    out = np.zeros((w * h, 2), np.uint16)
    lib().orc_zigzag(C.c_int(w), C.c_int(h), C.c_int(raw_stride), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def scaled_size(n, f):
    lib().orc_scaled_size.restype = C.c_int
    return int(lib().orc_scaled_size(C.c_int(n), C.c_int(f)))


def resize_nearest_down(img, f):
    img = np.ascontiguousarray(img, np.float32); h, w = img.shape
    o = np.zeros((scaled_size(h, f), scaled_size(w, f)), np.float32)
    lib().orc_resize_nearest_down(_fp(img), C.c_int(w), C.c_int(h), C.c_int(f), _fp(o)); return o


def resize_area(img, f):
    img = np.ascontiguousarray(img, np.float32); h, w = img.shape
    o = np.zeros((scaled_size(h, f), scaled_size(w, f)), np.float32)
    lib().orc_resize_area(_fp(img), C.c_int(w), C.c_int(h), C.c_int(f), _fp(o)); return o


def resize_linear(img, nw, nh):
    img = np.ascontiguousarray(img, np.float32); h, w = img.shape
    o = np.zeros((nh, nw), np.float32)
    lib().orc_resize_linear(_fp(img), C.c_int(w), C.c_int(h), C.c_int(nw), C.c_int(nh), _fp(o)); return o


def resize_nearest(img, nw, nh):
    img = np.ascontiguousarray(img, np.float32); h, w = img.shape
    o = np.zeros((nh, nw), np.float32)
    lib().orc_resize_nearest(_fp(img), C.c_int(w), C.c_int(h), C.c_int(nw), C.c_int(nh), _fp(o)); return o


def math_eval(kind, a, b=None):
    a = np.ascontiguousarray(a, np.float32); b = a if b is None else np.ascontiguousarray(b, np.float32)
    o = np.zeros_like(a)
    lib().orc_math_eval(C.c_int(kind), _fp(a), _fp(b), _fp(o), C.c_size_t(a.size)); return o


# ---- SGM oracle (oracle/sgm_oracle.cpp) -----------------------------------------------------
def sgm_generate_p2s(P2=4, alpha=14.0, beta=38.0):
    out = np.zeros(256, np.uint16)
    lib().orc_sgm_generate_p2s(C.c_uint16(P2), C.c_float(alpha), C.c_float(beta), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def sgm_match(left_bgr, left_gray, right_gray, pixels, num_costs, max_num_disp, P1, P2s):
    lb = np.ascontiguousarray(left_bgr, np.uint8); lg = np.ascontiguousarray(left_gray, np.float32); rg = np.ascontiguousarray(right_gray, np.float32)
    h, w = lg.shape
    px = np.ascontiguousarray(pixels); p2 = np.ascontiguousarray(P2s, np.uint16)
    d = np.zeros((h - 6, w - 6), np.int16); c = np.zeros((h - 6, w - 6), np.uint16)
    costs = np.zeros(num_costs, np.uint8); acc = np.zeros(num_costs, np.uint16)
    lib().orc_sgm_match(lb.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(lg), _fp(rg), C.c_int(w), C.c_int(h), px.ctypes.data_as(C.c_void_p),
                        C.c_uint64(num_costs), C.c_int(max_num_disp), C.c_uint16(P1), p2.ctypes.data_as(C.POINTER(C.c_uint16)),
                        d.ctypes.data_as(C.POINTER(C.c_int16)), c.ctypes.data_as(C.POINTER(C.c_uint16)),
                        costs.ctypes.data_as(C.POINTER(C.c_uint8)), acc.ctypes.data_as(C.POINTER(C.c_uint16)))
    return d, c, costs, acc


def sgm_step_forms_agree(Lp, pmin, pmax, costs, smin, smax, P1, P2) -> bool:
    Lp = np.ascontiguousarray(Lp, np.uint16); costs = np.ascontiguousarray(costs, np.uint8)
    lib().orc_sgm_step_forms_agree.restype = C.c_int
    return bool(lib().orc_sgm_step_forms_agree(Lp.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_int(pmin), C.c_int(pmax),
                                               costs.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(smin), C.c_int(smax), C.c_uint16(P1), C.c_uint16(P2)))


# ---- FilterDepthMap oracle (oracle/filter_oracle.cpp) ------------------------------------------
class FltView(C.Structure):
    _fields_ = [("depth", C.POINTER(C.c_float)), ("conf", C.POINTER(C.c_float)), ("K", C.c_double * 9), ("R", C.c_double * 9), ("C", C.c_double * 3),
                ("w", C.c_int), ("h", C.c_int)]       # size of this view's maps (0 = the reference view's): the reference sizes every depth map on its own


def filter_depth_map(depths, confs, K, R, Cc, ref, nbs, dmin, dmax, bAdjust=True, nMinViewsFilter=2, nMinViewsFilterAdjust=1,
                     nCalibratedImages=None, fDepthDiffThreshold=0.01):
    """One DepthMapsData::FilterDepthMap call: view `ref` against neighbour views `nbs` (ids into the arrays)."""
    keep = []

    def mk(i):
        v = FltView(); d = np.ascontiguousarray(depths[i], np.float32); c = np.ascontiguousarray(confs[i], np.float32); keep.extend([d, c])
        v.depth = _fp(d); v.conf = _fp(c); v.h, v.w = d.shape
        v.K[:] = np.asarray(K[i], np.float64).ravel(); v.R[:] = np.asarray(R[i], np.float64).ravel(); v.C[:] = np.asarray(Cc[i], np.float64).ravel()
        return v
    rv = mk(ref); arr = (FltView * max(1, len(nbs)))(*[mk(i) for i in nbs])
    h, w = depths[ref].shape
    nd = np.zeros((h, w), np.float32); nc = np.zeros((h, w), np.float32)
    lib().orc_filter_depth_map.restype = C.c_int
    rc = lib().orc_filter_depth_map(C.byref(rv), arr, C.c_int(len(nbs)), C.c_int(w), C.c_int(h), C.c_float(dmin), C.c_float(dmax), C.c_int(1 if bAdjust else 0),
                                    C.c_uint(nMinViewsFilter), C.c_uint(nMinViewsFilterAdjust), C.c_uint(nCalibratedImages or len(depths)), C.c_float(fDepthDiffThreshold),
                                    _fp(nd), _fp(nc))
    return rc, nd, nc


def gap_interpolation(depth, normal, conf, nIpolGapSize=7, fDepthDiffThreshold=0.01):
    d = np.ascontiguousarray(depth, np.float32).copy(); n = np.ascontiguousarray(normal, np.float32).copy(); c = np.ascontiguousarray(conf, np.float32).copy()
    h, w = d.shape
    lib().orc_gap_interpolation(_fp(d), _fp(n), _fp(c), C.c_int(w), C.c_int(h), C.c_uint(nIpolGapSize), C.c_float(fDepthDiffThreshold))
    return d, n, c


def remove_small_segments(depth, normal, conf, nSpeckleSize=100, fDepthDiffThreshold=0.01):
    d = np.ascontiguousarray(depth, np.float32).copy(); n = np.ascontiguousarray(normal, np.float32).copy(); c = np.ascontiguousarray(conf, np.float32).copy()
    h, w = d.shape
    lib().orc_remove_small_segments(_fp(d), _fp(n), _fp(c), C.c_int(w), C.c_int(h), C.c_uint(nSpeckleSize), C.c_float(fDepthDiffThreshold))
    return d, n, c


# ---- FuseDepthMaps oracle (oracle/fuse_oracle.cpp) ---------------------------------------------
class OrcFuseView(C.Structure):
    _fields_ = [("depth", C.POINTER(C.c_float)), ("normal", C.POINTER(C.c_float)), ("conf", C.POINTER(C.c_float)), ("bgr", C.POINTER(C.c_uint8)),
                ("K", C.c_double * 9), ("R", C.c_double * 9), ("C", C.c_double * 3), ("neighbors", C.POINTER(C.c_uint32)), ("nNeighbors", C.c_uint32),
                ("w", C.c_int), ("h", C.c_int)]       # this view's own map size (0 = the call's)


class OrcFuseCloud(C.Structure):
    _fields_ = [("nPoints", C.c_uint64), ("nDepths", C.c_uint64), ("nViews", C.c_uint64), ("points", C.POINTER(C.c_float)),
                ("viewStart", C.POINTER(C.c_uint32)), ("views", C.POINTER(C.c_uint32)), ("weights", C.POINTER(C.c_float)),
                ("projs", C.POINTER(C.c_uint16)), ("colors", C.POINTER(C.c_uint8)), ("normals", C.POINTER(C.c_float))]


def fuse_order(neighbor_counts, valid=None):
    """Processing order of FuseDepthMaps: images with a depth map, by decreasing neighbour count (SceneDensify.cpp:1423-1450); ties by index."""
    idx = [i for i in range(len(neighbor_counts)) if (valid is None or valid[i]) and neighbor_counts[i] > 0]
    return sorted(idx, key=lambda i: (-neighbor_counts[i], i))


def fuse_depth_maps(depths, normals, confs, bgrs, K, R, Cc, neighbors, order=None, nMinViewsFuse=2, fDepthDiffThreshold=0.01,
                    fNormalDiffThreshold=25.0, bEstimateColor=True, bEstimateNormal=True, fn=None):
    """One DepthMapsData::FuseDepthMaps call over all images.  depths[i] may be None (no depth map).  Returns a dict of arrays.
    `fn` overrides the C entry point (same signature), used to run the host emulation of the device kernels through the same wrapper."""
    n = len(depths)
    first = next(d for d in depths if d is not None)
    h, w = first.shape
    keep = []
    arr = (OrcFuseView * n)()

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
    if order is None:
        order = fuse_order([len(x) for x in neighbors], [d is not None for d in depths])
    od = np.ascontiguousarray(order, np.uint32)
    out = OrcFuseCloud()
    if nMinViewsFuse < 2 and fn is None:           # Scene::DenseReconstruction picks MergeDepthMaps then (SceneDensify.cpp:1695-1698)
        lib().orc_merge_depth_maps.restype = C.c_int
        rc = lib().orc_merge_depth_maps(arr, C.c_int(n), C.c_int(w), C.c_int(h), C.c_int(1 if bEstimateColor else 0), C.c_int(1 if bEstimateNormal else 0), C.byref(out))
        return _cloud(out, rc, lib().orc_fuse_free)
    f = fn or lib().orc_fuse_depth_maps
    f.restype = C.c_int
    # COS(FD2R(x)) in float with the C library's cosf, the same call the engine makes (numpy's float32 cos may differ in the last bit)
    libm = C.CDLL("libm.so.6"); libm.cosf.restype = C.c_float; libm.cosf.argtypes = [C.c_float]
    normalError = libm.cosf(float(np.float32(fNormalDiffThreshold) * (np.float32(3.14159265358979323846) / np.float32(180))))
    rc = f(arr, C.c_int(n), C.c_int(w), C.c_int(h), od.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_int(len(od)), C.c_uint(nMinViewsFuse),
           C.c_float(fDepthDiffThreshold), C.c_float(float(normalError)), C.c_int(1 if bEstimateColor else 0), C.c_int(1 if bEstimateNormal else 0), C.byref(out))
    return _cloud(out, rc, lib().orc_fuse_free if fn is None else getattr(fn, "_free", lib().orc_fuse_free))


def _cloud(out, rc, free):
    if rc != 0:
        raise RuntimeError(f"fuse failed: {rc}")
    P, V = int(out.nPoints), int(out.nViews)

    def take(p, cnt, dt):
        return np.ctypeslib.as_array(p, shape=(max(cnt, 1),))[:cnt].astype(dt, copy=True) if cnt >= 0 and bool(p) else None
    res = dict(nPoints=P, nDepths=int(out.nDepths), points=take(out.points, 3 * P, np.float32).reshape(P, 3),
               viewStart=take(out.viewStart, P + 1, np.uint32), views=take(out.views, V, np.uint32), weights=take(out.weights, V, np.float32),
               projs=take(out.projs, 2 * V, np.uint16).reshape(V, 2),
               colors=None if not out.colors else take(out.colors, 3 * P, np.uint8).reshape(P, 3),
               normals=None if not out.normals else take(out.normals, 3 * P, np.float32).reshape(P, 3))
    free(C.byref(out))
    return res


# ---- tSGM steps around Match (oracle/sgm_post_oracle.cpp); `impl` = a CDLL with the same entry points under another prefix ---------
def _sgm_post(prefix, impl=None):
    L = impl or lib()
    return lambda name: getattr(L, prefix + name)


def sgm_cross_check(l2r, r2l, thCross=1, impl=None, prefix="orc_sgm_"):
    a = np.ascontiguousarray(l2r, np.int16).copy(); b = np.ascontiguousarray(r2l, np.int16)
    _sgm_post(prefix, impl)("cross_check")(a.ctypes.data_as(C.POINTER(C.c_int16)), b.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int(a.shape[1]), C.c_int(a.shape[0]), C.c_int(b.shape[1]), C.c_int(thCross))
    return a


def sgm_filter_by_cost(disp, cost, th, impl=None, prefix="orc_sgm_"):
    a = np.ascontiguousarray(disp, np.int16).copy(); c = np.ascontiguousarray(cost, np.uint16)
    _sgm_post(prefix, impl)("filter_by_cost")(a.ctypes.data_as(C.POINTER(C.c_int16)), c.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_int(a.shape[1]), C.c_int(a.shape[0]), C.c_uint16(th))
    return a


def sgm_extract_mask(disp, mask=None, thValid=3, impl=None, prefix="orc_sgm_"):
    a = np.ascontiguousarray(disp, np.int16)
    m = np.zeros(a.shape, np.uint8) if mask is None else np.ascontiguousarray(mask, np.uint8).copy()
    _sgm_post(prefix, impl)("extract_mask")(a.ctypes.data_as(C.POINTER(C.c_int16)), m.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(a.shape[1]), C.c_int(a.shape[0]), C.c_int(thValid), C.c_int(1 if mask is None else 0))
    return m


def sgm_upscale_mask(mask, size2x, impl=None, prefix="orc_sgm_"):
    m = np.ascontiguousarray(mask, np.uint8); w2, h2 = size2x
    o = np.zeros((h2, w2), np.uint8)
    _sgm_post(prefix, impl)("upscale_mask")(m.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(m.shape[1]), C.c_int(m.shape[0]), o.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(w2), C.c_int(h2))
    return o


def sgm_flip_direction(l2r, impl=None, prefix="orc_sgm_"):
    a = np.ascontiguousarray(l2r, np.int16); o = np.zeros_like(a)
    _sgm_post(prefix, impl)("flip_direction")(a.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int(a.shape[1]), C.c_int(a.shape[0]), o.ctypes.data_as(C.POINTER(C.c_int16)))
    return o


def sgm_refine(disp, pixels, accums, mode=6, steps=4, impl=None, prefix="orc_sgm_"):
    """RefineDisparityMap: disp (valid-grid int16), pixels = the SGMHipPixelData table, accums = the 8-path sums (uint16)."""
    a = np.ascontiguousarray(disp, np.int16).copy(); px = np.ascontiguousarray(pixels); ac = np.ascontiguousarray(accums, np.uint16)
    _sgm_post(prefix, impl)("refine")(a.ctypes.data_as(C.POINTER(C.c_int16)), px.ctypes.data_as(C.c_void_p), ac.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_long(a.size), C.c_int(mode), C.c_int(steps))
    return a


_SGM_PIXEL = np.dtype([("idx", np.uint64), ("minDisp", np.int16), ("maxDisp", np.int16), ("pad", np.int32)])


def sgm_disparity2range_map(disp, mask2x, minNumDisp=5, minNumDispInvalid=7, impl=None, prefix="orc_sgm_"):
    a = np.ascontiguousarray(disp, np.int16); m = np.ascontiguousarray(mask2x, np.uint8)
    px = np.zeros(m.size, _SGM_PIXEL); mx = C.c_int(0)
    f = _sgm_post(prefix, impl)("disparity2range_map"); f.restype = C.c_ulonglong
    n = f(a.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int(a.shape[1]), C.c_int(a.shape[0]), m.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(m.shape[1]), C.c_int(m.shape[0]),
          C.c_int(minNumDisp), C.c_int(minNumDispInvalid), px.ctypes.data_as(C.c_void_p), C.byref(mx))
    return px, int(n), int(mx.value)


def _dp(a):
    a = np.ascontiguousarray(a, np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def sgm_depth2disparity_map(depth, invH, invQ, steps, size, impl=None, prefix="orc_sgm_"):
    d = np.ascontiguousarray(depth, np.float32); w, h = size
    o = np.zeros((h, w), np.int16); kh, ph = _dp(invH); kq, pq = _dp(invQ)
    _sgm_post(prefix, impl)("depth2disparity_map")(d.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(d.shape[1]), C.c_int(d.shape[0]), ph, pq, C.c_int(steps),
                                                   o.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int(w), C.c_int(h))
    return o


def sgm_disparity2depth_map(disp, cost, H, Q, steps, size, impl=None, prefix="orc_sgm_"):
    a = np.ascontiguousarray(disp, np.int16); c = None if cost is None else np.ascontiguousarray(cost, np.uint16); dw, dh = size
    dep = np.zeros((dh, dw), np.float32); cf = np.zeros((dh, dw), np.float32); kh, ph = _dp(H); kq, pq = _dp(Q)
    _sgm_post(prefix, impl)("disparity2depth_map")(a.ctypes.data_as(C.POINTER(C.c_int16)), None if c is None else c.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_int(a.shape[1]), C.c_int(a.shape[0]),
                                                   ph, pq, C.c_int(steps), dep.ctypes.data_as(C.POINTER(C.c_float)), cf.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(dw), C.c_int(dh))
    return dep, (None if c is None else cf)


def sgm_project_disparity2depth_map(disp, cost, Q, steps, size, impl=None, prefix="orc_sgm_"):
    a = np.ascontiguousarray(disp, np.int16); c = None if cost is None else np.ascontiguousarray(cost, np.uint16); dw, dh = size
    dep = np.zeros((dh, dw), np.float32); rg = np.zeros((dh, dw, 2), np.float32); cf = np.zeros((dh, dw), np.float32); kq, pq = _dp(Q)
    f = _sgm_post(prefix, impl)("project_disparity2depth_map"); f.restype = C.c_int
    ok = f(a.ctypes.data_as(C.POINTER(C.c_int16)), None if c is None else c.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_int(a.shape[1]), C.c_int(a.shape[0]), pq, C.c_int(steps),
           dep.ctypes.data_as(C.POINTER(C.c_float)), rg.ctypes.data_as(C.POINTER(C.c_float)), None if c is None else cf.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(dw), C.c_int(dh))
    return bool(ok), dep, rg, (None if c is None else cf)


def _fptrs(arrs, dt=np.float32):
    keep = [np.ascontiguousarray(a, dt) for a in arrs]
    return keep, (C.POINTER(C.c_float) * max(1, len(keep)))(*[k.ctypes.data_as(C.POINTER(C.c_float)) for k in keep])


def sgm_fuse_pairs(depths, ranges, confs, minViews=2, impl=None, prefix="orc_sgm_"):
    dh, dw = depths[0].shape
    kd, pd = _fptrs(depths); kr, pr = _fptrs(ranges); kc, pc = _fptrs(confs)
    dep = np.zeros((dh, dw), np.float32); cf = np.zeros((dh, dw), np.float32)
    _sgm_post(prefix, impl)("fuse_pairs")(pd, pr, pc, C.c_int(len(depths)), C.c_int(dw), C.c_int(dh), C.c_uint(minViews), dep.ctypes.data_as(C.POINTER(C.c_float)), cf.ctypes.data_as(C.POINTER(C.c_float)))
    return dep, cf


def sgm_filter_speckles(disp, maxSpeckleSize=100, maxDiff=5, impl=None, prefix="orc_sgm_"):
    a = np.ascontiguousarray(disp, np.int16).copy()
    _sgm_post(prefix, impl)("filter_speckles")(a.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int(a.shape[1]), C.c_int(a.shape[0]), C.c_int16(32767), C.c_int(maxSpeckleSize), C.c_int(maxDiff))
    return a

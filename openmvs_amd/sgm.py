"""ctypes binding of libsgmhip.so: host-side mirror of SemiGlobalMatcher::Match(ViewData, ViewData,
DisparityMap&, AccumCostMap&) (libs/MVS/SemiGlobalMatcher.cpp:863-1302).  No CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

PIXEL_DTYPE = np.dtype([("idx", np.uint64), ("minDisp", np.int16), ("maxDisp", np.int16), ("pad", np.int32)])
EXPORTS = ["sgmhip_create", "sgmhip_destroy", "sgmhip_last_error", "sgmhip_generate_p2s", "sgmhip_set_problem",
           "sgmhip_match", "sgmhip_get_results", "sgmhip_sync", "sgmhip_stats_reset", "sgmhip_stats_get",
           "sgmhip_consistency_cross_check", "sgmhip_filter_by_cost", "sgmhip_extract_mask", "sgmhip_upscale_mask", "sgmhip_flip_direction",
           "sgmhip_refine_disparity", "sgmhip_disparity2range_map", "sgmhip_depth2disparity_map", "sgmhip_disparity2depth_map",
           "sgmhip_project_disparity2depth_map", "sgmhip_fuse_pairs", "sgmhip_filter_speckles", "sgmhip_set_disparity", "sgmhip_tsgm_match", "sgmhip_fuse_disparities", "sgmhip_set_sub_group_kernels"]
NO_DISP = 32767          # SemiGlobalMatcher::NO_DISP
INVALID, VALID = 0, 255  # MaskMap values
SUBPIXEL_NA, SUBPIXEL_LINEAR, SUBPIXEL_POLY4, SUBPIXEL_PARABOLA, SUBPIXEL_SINE, SUBPIXEL_COSINE, SUBPIXEL_LC_BLEND = range(7)


class SGMHipStats(C.Structure):
    _fields_ = [("costMs", C.c_double), ("aggrMs", C.c_double), ("wtaMs", C.c_double), ("calls", C.c_uint64), ("aggrLaunches", C.c_uint64)]


_LIB = None


def load_library() -> C.CDLL:
    global _LIB
    if _LIB is None:
        path = os.environ.get("SGMHIP_LIB") or _build.build_lib("libsgmhip.so")
        if path is None or not os.path.exists(path):
            raise RuntimeError("libsgmhip.so is not built (python -m openmvs_amd.build)")
        lib = C.CDLL(path)
        if hasattr(lib, "hipemu_counters") and os.environ.get("OPENMVS_AMD_TEST_EMULATOR") != "1":
            # tests/cpp/hipemu builds of the kernels exist for the CPU test-suite only; the product never computes on the host
            raise RuntimeError("%s is a CPU-emulated test build, not a device library: refusing to load it outside the test-suite" % path)
        lib.sgmhip_last_error.restype = C.c_char_p
        lib.sgmhip_destroy.restype = None
        for n in EXPORTS:
            getattr(lib, n)
        _LIB = lib
    return _LIB


def generate_p2s(P2=4, alpha=14.0, beta=38.0) -> np.ndarray:
    """SemiGlobalMatcher::GenerateP2s with the ctor defaults (SemiGlobalMatcher.h:151)."""
    out = np.zeros(256, np.uint16)
    load_library().sgmhip_generate_p2s(C.c_uint16(P2), C.c_float(alpha), C.c_float(beta), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def make_pixels(ranges_min: np.ndarray, ranges_max: np.ndarray):
    """Build the PixelData table (SemiGlobalMatcher.h:79-82) for a valid grid: idx = running sum of numDisp."""
    mn = np.ascontiguousarray(ranges_min, np.int16); mx = np.ascontiguousarray(ranges_max, np.int16)
    nd = np.maximum(mx.astype(np.int64) - mn.astype(np.int64), 0).ravel()
    idx = np.concatenate([[0], np.cumsum(nd)[:-1]]).astype(np.uint64)
    px = np.zeros(nd.size, PIXEL_DTYPE)
    px["idx"] = idx; px["minDisp"] = mn.ravel(); px["maxDisp"] = mx.ravel()
    return px, int(nd.sum()), int(nd.max())


class SGMError(RuntimeError):
    pass


class SemiGlobalMatcherHIP:
    def __init__(self, device: int = 0, P1: int = 3, P2: int = 4, P2alpha: float = 14.0, P2beta: float = 38.0):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.sgmhip_create(C.c_int(device), C.byref(self._h))
        if rc != 0:
            raise SGMError(f"sgmhip_create failed ({rc}): no usable HIP device")
        self.P1 = P1
        self.P2s = generate_p2s(P2, P2alpha, P2beta)

    def _chk(self, rc):
        if rc != 0:
            raise SGMError(f"sgmhip error {rc}: {self._lib.sgmhip_last_error(self._h).decode()}")

    def set_problem(self, left_bgr, left_gray, right_gray, pixels, num_costs, max_num_disp):
        lb = np.ascontiguousarray(left_bgr, np.uint8); lg = np.ascontiguousarray(left_gray, np.float32); rg = np.ascontiguousarray(right_gray, np.float32)
        h, w = lg.shape
        px = np.ascontiguousarray(pixels)
        assert px.dtype == PIXEL_DTYPE and px.size == (w - 6) * (h - 6)
        self._shape = (h - 6, w - 6); self._num = num_costs
        self._chk(self._lib.sgmhip_set_problem(self._h, lb.ctypes.data_as(C.POINTER(C.c_uint8)), lg.ctypes.data_as(C.POINTER(C.c_float)),
                                               rg.ctypes.data_as(C.POINTER(C.c_float)), w, h, px.ctypes.data_as(C.c_void_p), C.c_uint64(num_costs), max_num_disp))

    def set_sub_group_kernels(self, lanes):
        """Match with sub-groups of `lanes` (8, 16, 32; True = 16) lanes per pixel / pair / line (narrow tSGM ranges) instead of one wavefront each
        (False / 0); same results."""
        self._chk(self._lib.sgmhip_set_sub_group_kernels(self._h, 16 if lanes is True else int(lanes)))

    def Match(self, sync=True):
        self._chk(self._lib.sgmhip_match(self._h, C.c_uint16(self.P1), self.P2s.ctypes.data_as(C.POINTER(C.c_uint16)), 1 if sync else 0))

    def results(self, volumes=False):
        d = np.zeros(self._shape, np.int16); c = np.zeros(self._shape, np.uint16)
        costs = np.zeros(self._num, np.uint8) if volumes else None
        acc = np.zeros(self._num, np.uint16) if volumes else None
        self._chk(self._lib.sgmhip_get_results(self._h, d.ctypes.data_as(C.POINTER(C.c_int16)), c.ctypes.data_as(C.POINTER(C.c_uint16)),
                                               costs.ctypes.data_as(C.POINTER(C.c_uint8)) if volumes else None,
                                               acc.ctypes.data_as(C.POINTER(C.c_uint16)) if volumes else None))
        return (d, c, costs, acc) if volumes else (d, c)

    # ---- the tSGM steps around Match (libs/MVS/SemiGlobalMatcher.cpp:1449-1811) -------------------------------------------
    def ConsistencyCrossCheck(self, l2r, r2l, thCross=1):
        a = np.ascontiguousarray(l2r, np.int16).copy(); b = np.ascontiguousarray(r2l, np.int16)
        self._chk(self._lib.sgmhip_consistency_cross_check(self._h, a.ctypes.data_as(C.POINTER(C.c_int16)), b.ctypes.data_as(C.POINTER(C.c_int16)),
                                                           a.shape[1], a.shape[0], b.shape[1], thCross))
        return a

    def FilterByCost(self, disparity, cost, th):
        a = np.ascontiguousarray(disparity, np.int16).copy(); c = np.ascontiguousarray(cost, np.uint16)
        self._chk(self._lib.sgmhip_filter_by_cost(self._h, a.ctypes.data_as(C.POINTER(C.c_int16)), c.ctypes.data_as(C.POINTER(C.c_uint16)), a.shape[1], a.shape[0], C.c_uint16(th)))
        return a

    def ExtractMask(self, disparity, mask=None, thValid=3):
        a = np.ascontiguousarray(disparity, np.int16)
        m = np.zeros(a.shape, np.uint8) if mask is None else np.ascontiguousarray(mask, np.uint8).copy()
        self._chk(self._lib.sgmhip_extract_mask(self._h, a.ctypes.data_as(C.POINTER(C.c_int16)), m.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[0], thValid, 1 if mask is None else 0))
        return m

    def UpscaleMask(self, mask, size2x):
        m = np.ascontiguousarray(mask, np.uint8); w2, h2 = size2x
        o = np.zeros((h2, w2), np.uint8)
        self._chk(self._lib.sgmhip_upscale_mask(self._h, m.ctypes.data_as(C.POINTER(C.c_uint8)), m.shape[1], m.shape[0], o.ctypes.data_as(C.POINTER(C.c_uint8)), w2, h2))
        return o

    def FlipDirection(self, l2r):
        a = np.ascontiguousarray(l2r, np.int16); o = np.zeros_like(a)
        self._chk(self._lib.sgmhip_flip_direction(self._h, a.ctypes.data_as(C.POINTER(C.c_int16)), a.shape[1], a.shape[0], o.ctypes.data_as(C.POINTER(C.c_int16))))
        return o

    def set_disparity(self, disparity):
        a = np.ascontiguousarray(disparity, np.int16)
        assert a.shape == self._shape
        self._chk(self._lib.sgmhip_set_disparity(self._h, a.ctypes.data_as(C.POINTER(C.c_int16))))

    def RefineDisparityMap(self, subpixelMode=SUBPIXEL_LC_BLEND, subpixelSteps=4):
        """In place on the device, on the result of the last Match(); read it back with results()."""
        self._chk(self._lib.sgmhip_refine_disparity(self._h, subpixelMode, subpixelSteps))

    def Disparity2RangeMap(self, disparity, mask2x, minNumDisp=5, minNumDispInvalid=7):
        """-> (pixel table of the 2x level, numCosts, maxNumDisp), ready for set_problem."""
        a = np.ascontiguousarray(disparity, np.int16); m = np.ascontiguousarray(mask2x, np.uint8)
        px = np.zeros(m.size, PIXEL_DTYPE); n = C.c_uint64(0); mx = C.c_int(0)
        self._chk(self._lib.sgmhip_disparity2range_map(self._h, a.ctypes.data_as(C.POINTER(C.c_int16)), a.shape[1], a.shape[0], m.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                       m.shape[1], m.shape[0], minNumDisp, minNumDispInvalid, px.ctypes.data_as(C.c_void_p), C.byref(n), C.byref(mx)))
        return px, int(n.value), int(mx.value)

    def Depth2DisparityMap(self, depth, invH, invQ, subpixelSteps, size):
        d = np.ascontiguousarray(depth, np.float32); w, h = size
        o = np.zeros((h, w), np.int16)
        ih = np.ascontiguousarray(invH, np.float64); iq = np.ascontiguousarray(invQ, np.float64)
        self._chk(self._lib.sgmhip_depth2disparity_map(self._h, d.ctypes.data_as(C.POINTER(C.c_float)), d.shape[1], d.shape[0], ih.ctypes.data_as(C.POINTER(C.c_double)),
                                                       iq.ctypes.data_as(C.POINTER(C.c_double)), subpixelSteps, o.ctypes.data_as(C.POINTER(C.c_int16)), w, h))
        return o

    def Disparity2DepthMap(self, disparity, cost, H, Q, subpixelSteps, size):
        a = np.ascontiguousarray(disparity, np.int16); c = None if cost is None else np.ascontiguousarray(cost, np.uint16); dw, dh = size
        dep = np.zeros((dh, dw), np.float32); cf = np.zeros((dh, dw), np.float32)
        hh = np.ascontiguousarray(H, np.float64); qq = np.ascontiguousarray(Q, np.float64)
        self._chk(self._lib.sgmhip_disparity2depth_map(self._h, a.ctypes.data_as(C.POINTER(C.c_int16)), None if c is None else c.ctypes.data_as(C.POINTER(C.c_uint16)), a.shape[1], a.shape[0],
                                                       hh.ctypes.data_as(C.POINTER(C.c_double)), qq.ctypes.data_as(C.POINTER(C.c_double)), subpixelSteps,
                                                       dep.ctypes.data_as(C.POINTER(C.c_float)), cf.ctypes.data_as(C.POINTER(C.c_float)), dw, dh))
        return dep, (None if c is None else cf)

    def ProjectDisparity2DepthMap(self, disparity, cost, Q, subpixelSteps, size):
        """-> (any depth produced, depthMap, depthRangeMap (h, w, 2), confMap or None)"""
        a = np.ascontiguousarray(disparity, np.int16); c = None if cost is None else np.ascontiguousarray(cost, np.uint16); dw, dh = size
        dep = np.zeros((dh, dw), np.float32); rg = np.zeros((dh, dw, 2), np.float32); cf = np.zeros((dh, dw), np.float32); anyd = C.c_int(0)
        qq = np.ascontiguousarray(Q, np.float64)
        self._chk(self._lib.sgmhip_project_disparity2depth_map(self._h, a.ctypes.data_as(C.POINTER(C.c_int16)), None if c is None else c.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                               a.shape[1], a.shape[0], qq.ctypes.data_as(C.POINTER(C.c_double)), subpixelSteps, dep.ctypes.data_as(C.POINTER(C.c_float)),
                                                               rg.ctypes.data_as(C.POINTER(C.c_float)), None if c is None else cf.ctypes.data_as(C.POINTER(C.c_float)), dw, dh, C.byref(anyd)))
        return bool(anyd.value), dep, rg, (None if c is None else cf)

    def FusePairs(self, depths, ranges, confs, minViews=2):
        """The per-pixel cluster fusion of SemiGlobalMatcher::Fuse over the maps of ProjectDisparity2DepthMap."""
        dh, dw = depths[0].shape
        keep = [[np.ascontiguousarray(a, np.float32) for a in arrs] for arrs in (depths, ranges, confs)]
        ptrs = [(C.POINTER(C.c_float) * len(k))(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in k]) for k in keep]
        dep = np.zeros((dh, dw), np.float32); cf = np.zeros((dh, dw), np.float32)
        self._chk(self._lib.sgmhip_fuse_pairs(self._h, ptrs[0], ptrs[1], ptrs[2], len(depths), dw, dh, C.c_uint(minViews), dep.ctypes.data_as(C.POINTER(C.c_float)), cf.ctypes.data_as(C.POINTER(C.c_float))))
        return dep, cf

    def tsgm_match(self, left_bgr, right_bgr, left_gray, right_gray, left_mask, right_mask, min_resolution=320, init_left_disparity=None,
                   n_speckle_size=100, subpixel_mode=SUBPIXEL_LC_BLEND, subpixel_steps=4):
        """The whole coarse-to-fine loop for a rectified pair in one call, resident on the device (sgmhip_tsgm_match; the same result as
        openmvs_amd.tsgm.tsgm_match driving the steps).  -> (disparity, cost, levels) on the valid grid."""
        lb = np.ascontiguousarray(left_bgr, np.uint8); rb = np.ascontiguousarray(right_bgr, np.uint8)
        lg = np.ascontiguousarray(left_gray, np.float32); rg = np.ascontiguousarray(right_gray, np.float32)
        lm = np.ascontiguousarray(left_mask, np.uint8); rm = np.ascontiguousarray(right_mask, np.uint8)
        h, w = lg.shape
        assert lb.shape == (h, w, 3) and rb.shape == (h, w, 3) and rg.shape == (h, w) and lm.shape == (h, w) and rm.shape == (h, w)
        init = None if init_left_disparity is None else np.ascontiguousarray(init_left_disparity, np.int16)
        d = np.zeros((h - 6, w - 6), np.int16); c = np.zeros((h - 6, w - 6), np.uint16); lv = C.c_int(0)
        u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8)); f32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        self._chk(self._lib.sgmhip_tsgm_match(self._h, u8(lb), u8(rb), f32(lg), f32(rg), u8(lm), u8(rm), w, h, C.c_uint(min_resolution),
                                              None if init is None else init.ctypes.data_as(C.POINTER(C.c_int16)), n_speckle_size, subpixel_mode, subpixel_steps,
                                              C.c_uint16(self.P1), self.P2s.ctypes.data_as(C.POINTER(C.c_uint16)), d.ctypes.data_as(C.POINTER(C.c_int16)),
                                              c.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(lv)))
        self._shape = d.shape
        return d, c, int(lv.value)

    def fuse_disparities(self, pairs, size, minViews=2):
        """SemiGlobalMatcher::Fuse in one resident call.  pairs: dicts with disparity, cost, Q, subpixel_steps (the .dimap content); size = (w, h) of
        the reference image.  -> (depthMap, confMap, number of pairs that produced depths)."""
        n = len(pairs); dw, dh = size
        ds = [np.ascontiguousarray(p["disparity"], np.int16) for p in pairs]; cs = [np.ascontiguousarray(p["cost"], np.uint16) for p in pairs]
        dp = (C.POINTER(C.c_int16) * max(n, 1))(*[a.ctypes.data_as(C.POINTER(C.c_int16)) for a in ds])
        cp = (C.POINTER(C.c_uint16) * max(n, 1))(*[a.ctypes.data_as(C.POINTER(C.c_uint16)) for a in cs])
        ws = (C.c_int * max(n, 1))(*[a.shape[1] for a in ds]); hs = (C.c_int * max(n, 1))(*[a.shape[0] for a in ds])
        qs = np.ascontiguousarray(np.stack([np.asarray(p["Q"], np.float64).reshape(16) for p in pairs]) if n else np.zeros((1, 16)), np.float64)
        st = (C.c_int * max(n, 1))(*[int(p["subpixel_steps"]) for p in pairs])
        dep = np.zeros((dh, dw), np.float32); cf = np.zeros((dh, dw), np.float32); used = C.c_int(0)
        self._chk(self._lib.sgmhip_fuse_disparities(self._h, n, dp, cp, ws, hs, qs.ctypes.data_as(C.POINTER(C.c_double)), st, dw, dh, C.c_uint(minViews),
                                                    dep.ctypes.data_as(C.POINTER(C.c_float)), cf.ctypes.data_as(C.POINTER(C.c_float)), C.byref(used)))
        return dep, cf, int(used.value)

    def FilterSpeckles(self, disparity, maxSpeckleSize=100, maxDiff=5):
        a = np.ascontiguousarray(disparity, np.int16).copy()
        self._chk(self._lib.sgmhip_filter_speckles(self._h, a.ctypes.data_as(C.POINTER(C.c_int16)), a.shape[1], a.shape[0], maxSpeckleSize, maxDiff))
        return a

    def sync(self):
        self._chk(self._lib.sgmhip_sync(self._h))

    def stats_reset(self, enable=True):
        self._chk(self._lib.sgmhip_stats_reset(self._h, 1 if enable else 0))

    def stats_get(self) -> SGMHipStats:
        s = SGMHipStats(); self._chk(self._lib.sgmhip_stats_get(self._h, C.byref(s))); return s

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sgmhip_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

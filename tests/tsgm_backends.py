"""Backends for openmvs_amd.tsgm.tsgm_match used by the tests: the CPU oracle (every step from oracle/) and the device (SemiGlobalMatcherHIP
plus the PatchMatch engine's float INTER_AREA resampler).  The loop itself is product code and is the same for both."""
import numpy as np

from oracle import pyoracle as po


class OracleBackend:
    def __init__(self, P1=3):
        self.P1 = P1; self.P2s = po.sgm_generate_p2s()

    def resize_area_f32(self, img, f):
        return po.resize_area(img, f)

    def set_problem(self, bgr, gl, gr, px, n, mx):
        self._prob = (bgr, gl, gr, px, n, mx)

    def Match(self):
        bgr, gl, gr, px, n, mx = self._prob
        self._d, self._c, self._costs, self._acc = po.sgm_match(bgr, gl, gr, px, n, mx, self.P1, self.P2s)

    def results(self):
        return self._d.copy(), self._c.copy()

    def set_disparity(self, d):
        self._d = np.ascontiguousarray(d, np.int16).copy()

    def RefineDisparityMap(self, mode, steps):
        self._d = po.sgm_refine(self._d, self._prob[3], self._acc, mode, steps)

    ConsistencyCrossCheck = staticmethod(lambda a, b, th=1: po.sgm_cross_check(a, b, th))
    FilterSpeckles = staticmethod(lambda d, n, df: po.sgm_filter_speckles(d, n, df))
    ExtractMask = staticmethod(lambda d, m=None, tv=3: po.sgm_extract_mask(d, m, tv))
    UpscaleMask = staticmethod(lambda m, size: po.sgm_upscale_mask(m, size))
    FlipDirection = staticmethod(lambda d: po.sgm_flip_direction(d))
    Disparity2RangeMap = staticmethod(lambda d, m, a, b: po.sgm_disparity2range_map(d, m, a, b))
    Depth2DisparityMap = staticmethod(lambda depth, invH, invQ, steps, size: po.sgm_depth2disparity_map(depth, invH, invQ, steps, size))
    ProjectDisparity2DepthMap = staticmethod(lambda d, c, Q, steps, size: po.sgm_project_disparity2depth_map(d, c, Q, steps, size))
    FusePairs = staticmethod(lambda deps, rgs, cfs, mv=2: po.sgm_fuse_pairs(deps, rgs, cfs, mv))


class DeviceBackend:
    """SemiGlobalMatcherHIP already has every method of the interface except the float resampler, which the PatchMatch library provides."""

    def __init__(self, matcher, engine):
        self.m, self.e = matcher, engine

    def resize_area_f32(self, img, f):
        return self.e.resize(0, img, f)

    def __getattr__(self, name):
        return getattr(self.m, name)

"""ctypes binding of libmvsfront.so (include/mvsfront.h): the C++ scene front end -- MVSI reader, pixel cameras, neighbour-view selection,
sparse depth initialisation.  Same results as the numpy implementation in mvsi.py / views.py (tests/test_mvsfront.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import build as _build

VIEW_SCORE_DTYPE = np.dtype([("ID", "<u4"), ("points", "<u4"), ("scale", "<f4"), ("angle", "<f4"), ("area", "<f4"), ("score", "<f4")])
EXPORTS = ["mvsf_default_options", "mvsf_load", "mvsf_free", "mvsf_version", "mvsf_num_images", "mvsf_num_points", "mvsf_image_info", "mvsf_point",
           "mvsf_camera", "mvsf_select_views", "mvsf_select_neighbor_views", "mvsf_init_depth_map", "mvsf_init_depth_map_dense", "mvsf_triangulate_depth_map",
           "mvsf_get_neighbors", "mvsf_set_neighbors", "mvsf_load_view_neighbors", "mvsf_save_view_neighbors", "mvsf_image_depths", "mvsf_estimate_normal_map"]


class MVSFOptions(C.Structure):
    _fields_ = [("nMinViews", C.c_uint32), ("nMaxViews", C.c_uint32), ("nMinViewsTrustPoint", C.c_uint32), ("nNumViews", C.c_uint32), ("nPointInsideROI", C.c_uint32),
                ("fViewMinScore", C.c_float), ("fViewMinScoreRatio", C.c_float), ("fMinArea", C.c_float), ("fMinAngle", C.c_float), ("fOptimAngle", C.c_float), ("fMaxAngle", C.c_float)]


_LIB = None


def load_library() -> C.CDLL:
    global _LIB
    if _LIB is None:
        lib = C.CDLL(_build.build_host_lib("libmvsfront.so"))
        lib.mvsf_free.restype = None; lib.mvsf_default_options.restype = None
        for n in EXPORTS:
            getattr(lib, n)
        _LIB = lib
    return _LIB


def default_options(**kw) -> MVSFOptions:
    o = MVSFOptions(); load_library().mvsf_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class SceneFront:
    def __init__(self, path: str):
        self._lib = load_library(); self._h = C.c_void_p()
        rc = self._lib.mvsf_load(path.encode(), C.byref(self._h))
        if rc != 0:
            raise ValueError("mvsf_load(%s) failed: %d" % (path, rc))
        self.version = self._lib.mvsf_version(self._h); self.n_images = self._lib.mvsf_num_images(self._h); self.n_points = self._lib.mvsf_num_points(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mvsf_free(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def image_info(self, i):
        name = C.create_string_buffer(1024); w = C.c_int(); h = C.c_int(); v = C.c_int()
        assert self._lib.mvsf_image_info(self._h, i, name, 1024, C.byref(w), C.byref(h), C.byref(v)) == 0
        return name.value.decode(), w.value, h.value, bool(v.value)

    def point(self, i):
        X = np.zeros(3, np.float32); n = C.c_int(); views = np.zeros(64, np.uint32)
        assert self._lib.mvsf_point(self._h, i, X.ctypes.data_as(C.POINTER(C.c_float)), views.ctypes.data_as(C.POINTER(C.c_uint32)), 64, C.byref(n)) == 0
        return X, views[:n.value].copy()

    def camera(self, i, size=(0, 0)):
        K = np.zeros((3, 3)); R = np.zeros((3, 3)); Cc = np.zeros(3)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        rc = self._lib.mvsf_camera(self._h, i, int(size[0]), int(size[1]), dp(K), dp(R), dp(Cc))
        if rc != 0:
            raise ValueError("mvsf_camera failed: %d" % rc)
        return K, R, Cc

    def neighbors(self, i):
        """The neighbour list image i carries (archive view scores or a view-neighbours file); empty = to be selected from the sparse points."""
        n = C.c_int()
        assert self._lib.mvsf_get_neighbors(self._h, i, None, 0, C.byref(n)) == 0
        nb = np.zeros(n.value, VIEW_SCORE_DTYPE)
        assert self._lib.mvsf_get_neighbors(self._h, i, nb.ctypes.data_as(C.c_void_p), len(nb), C.byref(n)) == 0
        return nb

    def set_neighbors(self, i, nb):
        nb = np.ascontiguousarray(nb, VIEW_SCORE_DTYPE)
        if self._lib.mvsf_set_neighbors(self._h, i, nb.ctypes.data_as(C.c_void_p), len(nb)) != 0:
            raise ValueError("mvsf_set_neighbors(%d)" % i)

    def load_view_neighbors(self, path: str):
        """Scene::LoadViewNeighbors (`--view-neighbors-file`)."""
        rc = self._lib.mvsf_load_view_neighbors(self._h, path.encode())
        if rc != 0:
            raise ValueError("mvsf_load_view_neighbors(%s) failed: %d" % (path, rc))

    def save_view_neighbors(self, path: str):
        if self._lib.mvsf_save_view_neighbors(self._h, path.encode()) != 0:
            raise OSError("cannot write %s" % path)

    def image_depths(self, i):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        assert self._lib.mvsf_image_depths(self._h, i, C.byref(a), C.byref(b), C.byref(c)) == 0
        return a.value, b.value, c.value

    def _sizes(self, sizes):
        if sizes is None:
            return None, None
        a = np.ascontiguousarray(sizes, np.int32).reshape(-1)
        return a, a.ctypes.data_as(C.POINTER(C.c_int))

    def select_neighbor_views(self, i, nMinViews=2, nMinPointViews=2, fOptimAngle=12.0, nInsideROI=1, sizes=None):
        nb = np.zeros(self.n_images, VIEW_SCORE_DTYPE); pts = np.zeros(self.n_points, np.uint32); nn = C.c_int(); npts = C.c_int(); avg = C.c_float()
        keep, ps = self._sizes(sizes)
        rc = self._lib.mvsf_select_neighbor_views(self._h, i, ps, nMinViews, nMinPointViews, C.c_float(fOptimAngle), nInsideROI, nb.ctypes.data_as(C.c_void_p), len(nb), C.byref(nn),
                                                  pts.ctypes.data_as(C.POINTER(C.c_uint32)), len(pts), C.byref(npts), C.byref(avg))
        return rc == 0, nb[:nn.value].copy(), pts[:npts.value].copy(), float(avg.value)

    def select_views(self, i, opt=None, sizes=None):
        opt = opt or default_options()
        nb = np.zeros(self.n_images, VIEW_SCORE_DTYPE); pts = np.zeros(self.n_points, np.uint32); nn = C.c_int(); npts = C.c_int(); avg = C.c_float()
        keep, ps = self._sizes(sizes)
        rc = self._lib.mvsf_select_views(self._h, i, ps, C.byref(opt), nb.ctypes.data_as(C.c_void_p), len(nb), C.byref(nn), pts.ctypes.data_as(C.POINTER(C.c_uint32)), len(pts), C.byref(npts), C.byref(avg))
        if rc != 0:
            return None
        return nb[:nn.value].copy(), pts[:npts.value].copy(), float(avg.value)

    def init_depth_map(self, i, points, size, nMinViewsTrustPoint=2, dense=False):
        """dense: OPTDENSE::bInitSparse = 0 (rasterised triangles instead of 2x2 splats)."""
        w, h = size
        d = np.zeros((h, w), np.float32); n = np.zeros((h, w, 3), np.float32); dmin = C.c_float(); dmax = C.c_float()
        p = np.ascontiguousarray(points, np.uint32)
        out = (d.ctypes.data_as(C.POINTER(C.c_float)), n.ctypes.data_as(C.POINTER(C.c_float)), C.byref(dmin), C.byref(dmax))
        if dense:
            rc = self._lib.mvsf_init_depth_map_dense(self._h, i, w, h, p.ctypes.data_as(C.POINTER(C.c_uint32)), len(p), *out)
        else:
            rc = self._lib.mvsf_init_depth_map(self._h, i, w, h, p.ctypes.data_as(C.POINTER(C.c_uint32)), len(p), nMinViewsTrustPoint, *out)
        if rc != 0:
            raise ValueError("mvsf_init_depth_map failed: %d" % rc)
        return d, n, float(dmin.value), float(dmax.value)

    def triangulate_depth_map(self, i, points, size, avg_depth=None, sparse=False):
        """mvsf_triangulate_depth_map: avg_depth not None = with the image corners.  -> (depthMap, dMin, dMax)."""
        w, h = size
        d = np.zeros((h, w), np.float32); dmin = C.c_float(); dmax = C.c_float()
        p = np.ascontiguousarray(points, np.uint32)
        rc = self._lib.mvsf_triangulate_depth_map(self._h, i, w, h, p.ctypes.data_as(C.POINTER(C.c_uint32)), len(p), int(avg_depth is not None),
                                                  C.c_float(avg_depth or 0.0), int(sparse), d.ctypes.data_as(C.POINTER(C.c_float)), C.byref(dmin), C.byref(dmax))
        if rc != 0:
            raise ValueError("mvsf_triangulate_depth_map failed: %d" % rc)
        return d, float(dmin.value), float(dmax.value)


def estimate_normal_map(K, depth):
    """mvsf_estimate_normal_map (MVS::EstimateNormalMap): K 3x3, depth [h, w] -> normals [h, w, 3] float32."""
    d = np.ascontiguousarray(depth, np.float32); h, w = d.shape
    k = np.ascontiguousarray(np.asarray(K, np.float64).ravel())
    out = np.zeros((h, w, 3), np.float32)
    rc = load_library().mvsf_estimate_normal_map(k.ctypes.data_as(C.POINTER(C.c_double)), d.ctypes.data_as(C.POINTER(C.c_float)), w, h, out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc != 0:
        raise ValueError("mvsf_estimate_normal_map: %d" % rc)
    return out

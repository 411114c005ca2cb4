"""ctypes binding of libpmhip.so -- the host-side mirror of the reference's GPU plug-in.

`PatchMatchHIP` has the same four-method surface as the reference's `PatchMatchCUDA`
(libs/MVS/PatchMatchCUDA.inl:102-108: ctor(device), Init(bGeomConsistency), Release(),
EstimateDepthMap(DepthData&)), plus the HBM-resident scene interface of include/pmhip.h.
There is no CPU fallback: if the HIP library or a GPU is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

MAX_SOURCES = 16


class PMHipParams(C.Structure):
    _fields_ = [("nSubResolutionLevels", C.c_uint32), ("nEstimationIters", C.c_uint32),
                ("nEstimationGeometricIters", C.c_uint32), ("nRandomIters", C.c_uint32),
                ("fEstimationGeometricWeight", C.c_float), ("fRandomDepthRatio", C.c_float),
                ("fRandomAngle1Range", C.c_float), ("fRandomAngle2Range", C.c_float),
                ("fRandomSmoothDepth", C.c_float), ("fRandomSmoothNormal", C.c_float),
                ("fRandomSmoothBonus", C.c_float), ("fNCCThresholdKeep", C.c_float),
                ("fDescriptorMinMagnitudeThreshold", C.c_float), ("seed", C.c_uint32)]


class PMHipView(C.Structure):
    _fields_ = [("image", C.POINTER(C.c_float)), ("w", C.c_int32), ("h", C.c_int32),
                ("K", C.c_double * 9), ("R", C.c_double * 9), ("C", C.c_double * 3),
                ("depth", C.POINTER(C.c_float)),
                ("Kd", C.c_double * 9), ("Rd", C.c_double * 9), ("Cd", C.c_double * 3),
                ("id", C.c_uint32), ("dw", C.c_int32), ("dh", C.c_int32)]


class PMHipDepthData(C.Structure):
    _fields_ = [("views", C.POINTER(PMHipView)), ("nViews", C.c_int32),
                ("depthMap", C.POINTER(C.c_float)), ("normalMap", C.POINTER(C.c_float)),
                ("confMap", C.POINTER(C.c_float)), ("dMin", C.c_float), ("dMax", C.c_float)]


class PMHipKernelStats(C.Structure):
    _fields_ = [("sweepLaunches", C.c_uint64), ("sweepMs", C.c_double), ("sweepBytes", C.c_double),
                ("sweepPixels", C.c_uint64), ("initLaunches", C.c_uint64), ("initMs", C.c_double), ("sweepWallMs", C.c_double), ("sweepHostMs", C.c_double)]


class PMHipFuseParams(C.Structure):
    _fields_ = [("nMinViewsFuse", C.c_uint32), ("fDepthDiffThreshold", C.c_float), ("fNormalDiffThreshold", C.c_float),
                ("bEstimateColor", C.c_int32), ("bEstimateNormal", C.c_int32)]


PMHIP_ABI_VERSION = 6      # include/pmhip.h


class PMHipTuning(C.Structure):
    _fields_ = [("viewGroups", C.c_int32), ("wideMaxViews", C.c_int32), ("wideHyps", C.c_int32), ("sweepLanes", C.c_int32), ("quadBuffer", C.c_int32), ("widePixels", C.c_int32), ("wide8Pixels", C.c_int32), ("reserved0", C.c_int32)]


EXPORTS = ["pmhip_get_tuning", "pmhip_set_tuning", "pmhip_scene_set_view_id", "pmhip_scene_set_view_sized", "pmhip_scene_set_source_depth", "pmhip_scene_set_mask", "pmhip_scene_set_mask_mode", "pmhip_scene_set_conf", "pmhip_scene_set_color", "pmhip_scene_fuse", "pmhip_scene_fuse_get", "pmhip_scene_fuse_rounds", "pmhip_default_params", "pmhip_create", "pmhip_destroy", "pmhip_init", "pmhip_release",
           "pmhip_estimate_depth_map", "pmhip_estimate_depth_map_masked", "pmhip_last_error", "pmhip_scene_create", "pmhip_scene_set_view",
           "pmhip_scene_estimate", "pmhip_scene_commit_round", "pmhip_scene_reset_view", "pmhip_scene_set_maps",
           "pmhip_scene_get_maps", "pmhip_scene_device_ptr", "pmhip_scene_copy", "pmhip_scene_filter", "pmhip_scene_filter_commit", "pmhip_scene_gap_interpolation", "pmhip_scene_remove_small_segments", "pmhip_scene_images_updated", "pmhip_scene_maps_updated", "pmhip_scene_bytes", "pmhip_sync",
           "pmhip_stream", "pmhip_stats_reset", "pmhip_stats_get", "pmhip_prof_get", "pmhip_math_eval", "pmhip_resize", "pmhip_set_sweep_tiles", "pmhip_abi_version"]

_LIB = None


def load_library() -> C.CDLL:
    """Load libpmhip.so (building it first when the sources are newer).  Fails loudly."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("PMHIP_LIB") or _build.build_lib("libpmhip.so")   # PMHIP_LIB: tuning experiments only
        if path is None or not os.path.exists(path):
            raise RuntimeError("libpmhip.so is not built (python -m openmvs_amd.build)")
        lib = C.CDLL(path)
        if hasattr(lib, "hipemu_counters") and os.environ.get("OPENMVS_AMD_TEST_EMULATOR") != "1":
            # tests/cpp/hipemu builds of the kernels exist for the CPU test-suite only; the product never computes on the host
            raise RuntimeError("%s is a CPU-emulated test build, not a device library: refusing to load it outside the test-suite" % path)
        lib.pmhip_last_error.restype = C.c_char_p
        lib.pmhip_scene_device_ptr.restype = C.c_void_p
        lib.pmhip_stream.restype = C.c_void_p
        lib.pmhip_destroy.restype = None
        lib.pmhip_scene_fuse_rounds.restype = C.c_uint64
        experiment = bool(os.environ.get("PMHIP_LIB"))        # (an older build of the library in an A/B run may lack the newest entry points)
        for n in EXPORTS:
            if not (experiment and n in ("pmhip_set_sweep_tiles", "pmhip_abi_version")):
                getattr(lib, n)  # raises AttributeError if a declared symbol is missing
        if hasattr(lib, "pmhip_abi_version"):
            lib.pmhip_abi_version.restype = C.c_uint32
        if hasattr(lib, "pmhip_abi_version") and lib.pmhip_abi_version() != PMHIP_ABI_VERSION and not experiment:
            raise RuntimeError("%s has struct layout version %d, these bindings are written for %d (include/pmhip.h)" % (path, lib.pmhip_abi_version(), PMHIP_ABI_VERSION))
        _LIB = lib
    return _LIB


def default_params(**kw) -> PMHipParams:
    p = PMHipParams()
    load_library().pmhip_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    """float64 array -> double* (the array is kept alive by ctypes for the duration of the call)."""
    a = np.ascontiguousarray(a, np.float64)
    p = a.ctypes.data_as(C.POINTER(C.c_double)); p._keep = a
    return p


class PatchMatchError(RuntimeError):
    pass


class PatchMatchHIP:
    """Mirror of `class PatchMatchCUDA` (libs/MVS/PatchMatchCUDA.inl:76-139)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.pmhip_create(C.c_int(device), C.byref(self._h))
        if rc != 0:
            raise PatchMatchError(f"pmhip_create failed ({rc}): no usable MI355X / HIP device")
        self.params = default_params()

    def _chk(self, rc):
        if rc != 0:
            raise PatchMatchError(f"pmhip error {rc}: {self._lib.pmhip_last_error(self._h).decode()}")

    # -- the four reference methods ---------------------------------------------------------
    def Init(self, bGeomConsistency: bool):
        self._chk(self._lib.pmhip_init(self._h, C.c_int(1 if bGeomConsistency else 0)))

    def Release(self):
        if self._h:
            self._chk(self._lib.pmhip_release(self._h))

    def EstimateDepthMap(self, gray, K, R, Cc, ids, dmin, dmax, depth=None, normal=None, src_depths=None,
                         nGeometricIter: int = -1, params: PMHipParams | None = None, mask=None, mask_option=False, src_depth_cams=None):
        """One depth map.  ids[0] = reference view, ids[1:] = sources (indices into gray/K/R/Cc; the source images may have any size).
        src_depths: dict id -> depth map (any size), required for a geometric round; src_depth_cams: dict id -> (Kd, Rd, Cd), the camera
        stored with that depth map (default: the view's own).  mask: (h, w) uint8, 0 = ignored pixel (DepthData::mask);
        mask_option: OPTDENSE::nIgnoreMaskLabel is set although this view has no mask.  Returns (depth, normal, conf)."""
        p = params or self.params
        n = len(ids)
        views = (PMHipView * n)()
        keep = []
        for k, i in enumerate(ids):
            img = np.ascontiguousarray(gray[i], np.float32); keep.append(img)
            v = views[k]
            v.image = _fp(img); v.h, v.w = img.shape
            v.K[:] = np.asarray(K[i], np.float64).ravel(); v.R[:] = np.asarray(R[i], np.float64).ravel(); v.C[:] = np.asarray(Cc[i], np.float64).ravel()
            v.id = int(i)
            if k > 0 and src_depths is not None:
                d = np.ascontiguousarray(src_depths[i], np.float32); keep.append(d)
                v.depth = _fp(d); v.dh, v.dw = d.shape
                if src_depth_cams is not None and i in src_depth_cams:
                    kd, rd, cd = src_depth_cams[i]
                    v.Kd[:] = np.asarray(kd, np.float64).ravel(); v.Rd[:] = np.asarray(rd, np.float64).ravel(); v.Cd[:] = np.asarray(cd, np.float64).ravel()
                else:
                    v.Kd[:] = v.K[:]; v.Rd[:] = v.R[:]; v.Cd[:] = v.C[:]
        h, w = keep[0].shape
        depth = np.zeros((h, w), np.float32) if depth is None else np.ascontiguousarray(depth, np.float32).copy()
        normal = np.zeros((h, w, 3), np.float32) if normal is None else np.ascontiguousarray(normal, np.float32).copy()
        conf = np.zeros((h, w), np.float32)
        dd = PMHipDepthData(views, n, _fp(depth), _fp(normal), _fp(conf), float(dmin), float(dmax))
        if mask is None and not mask_option:
            self._chk(self._lib.pmhip_estimate_depth_map(self._h, C.byref(dd), C.byref(p), C.c_int(nGeometricIter)))
        else:
            m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
            if m is not None and m.shape != (h, w):
                raise ValueError("mask must have the reference view's size")
            self._chk(self._lib.pmhip_estimate_depth_map_masked(self._h, C.byref(dd), None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                                1 if mask_option else 0, C.byref(p), C.c_int(nGeometricIter)))
        return depth, normal, conf

    # -- HBM-resident scene interface --------------------------------------------------------
    def scene_create(self, n_images, w, h, n_levels=2):
        self._chk(self._lib.pmhip_scene_create(self._h, n_images, w, h, n_levels))
        self._scene = (n_images, w, h); self._sizes = {}

    def scene_set_view(self, idx, gray, K, R, Cc, dmin, dmax, neighbors, device_ptr=None):
        K = np.ascontiguousarray(K, np.float64); R = np.ascontiguousarray(R, np.float64); Cc = np.ascontiguousarray(Cc, np.float64)
        nb = np.ascontiguousarray(neighbors, np.int32)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        if device_ptr is not None:
            src, ondev = C.cast(C.c_void_p(device_ptr), C.POINTER(C.c_float)), 1
        elif gray is None:
            src, ondev = None, 0
        else:
            g = np.ascontiguousarray(gray, np.float32); src, ondev = _fp(g), 0
        if src is not None:
            getattr(self, "_sizes", {}).pop(int(idx), None)       # (an image of the scene's size takes the view back into the scene arrays)
        self._chk(self._lib.pmhip_scene_set_view(self._h, idx, src, ondev, dp(K), dp(R), dp(Cc), C.c_float(dmin), C.c_float(dmax),
                                                 nb.ctypes.data_as(C.POINTER(C.c_int32)), len(nb)))

    def tuning(self, **kw):
        """pmhip_get_tuning / pmhip_set_tuning: how a batch is mapped onto the GPU (viewGroups, wideMaxViews, wideHyps, sweepLanes, quadBuffer, widePixels, wide8Pixels); returns the settings
        in force as a dict.  The results never depend on them."""
        t = PMHipTuning()
        if kw:
            names = {k for k, _ in PMHipTuning._fields_}
            for k, v in kw.items():
                if k not in names:
                    raise KeyError("PMHipTuning has no field %r" % k)
                setattr(t, k, int(v))
            self._chk(self._lib.pmhip_set_tuning(self._h, C.byref(t)))
        self._chk(self._lib.pmhip_get_tuning(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in PMHipTuning._fields_}

    def set_sweep_tiles(self, tile_w: int = 0, tile_h: int = 0):
        """OPT-IN, not the reference's estimator (pmhip_set_sweep_tiles): sweeps run inside tile_w x tile_h tiles, neighbours across a tile border are read as the previous
        sweep left them.  0, 0 = off (the reference's sweep, bit for bit).  The result equals the oracle's with tileW / tileH set."""
        self._chk(self._lib.pmhip_set_sweep_tiles(self._h, int(tile_w), int(tile_h)))

    def scene_set_view_id(self, idx, view_id):
        """The identity slot idx draws its random numbers under (pmhip_scene_set_view_id): the view's index in the whole scene when this engine holds a part of it."""
        self._chk(self._lib.pmhip_scene_set_view_id(self._h, int(idx), C.c_uint32(int(view_id))))

    def scene_load(self, scene, n_levels=2):
        """Upload a synth.Scene (or anything with the same attributes)."""
        self.scene_create(scene.n_views, scene.width, scene.height, n_levels)
        sizes = getattr(scene, "sizes", None) or []
        nbs = getattr(scene, "estimate_neighbors", None) or scene.neighbors       # (a densify.SceneViews lists resampled copies of neighbours here, ViewData::ScaleImage)
        for i in range(scene.n_views):
            own = i < len(sizes) and tuple(sizes[i]) != (scene.width, scene.height)
            (self.scene_set_view_sized if own else self.scene_set_view)(i, scene.gray[i], scene.K[i], scene.R[i], scene.C[i], float(scene.dmin[i]), float(scene.dmax[i]), nbs[i])
        # ignore masks of a scene loaded with --ignore-mask-label (densify.load_scene): per image where a mask file was found; the option alone already selects the
        # nearest-neighbour level hand-off (SceneDensify.cpp:661)
        for i, m in getattr(scene, "masks", {}).items():
            self.scene_set_mask(i, m)
        if getattr(scene, "mask_option", False):
            self.scene_set_mask_mode(1)

    def scene_set_view_sized(self, idx, gray, K, R, Cc, dmin, dmax, neighbors):
        """A view whose image -- and therefore its depth, normal and confidence maps -- has its own size (pmhip_scene_set_view_sized): a source view or a reference view."""
        g = np.ascontiguousarray(gray, np.float32)
        self._sizes = getattr(self, "_sizes", {}); self._sizes[int(idx)] = (g.shape[1], g.shape[0])
        nb = np.ascontiguousarray(neighbors, np.int32)
        self._chk(self._lib.pmhip_scene_set_view_sized(self._h, idx, _fp(g), g.shape[1], g.shape[0], 0, _dp(K), _dp(R), _dp(Cc), C.c_float(dmin), C.c_float(dmax),
                                                       nb.ctypes.data_as(C.POINTER(C.c_int32)), len(nb)))

    def scene_set_source_depth(self, idx, depth, Kd=None, Rd=None, Cd=None):
        """Known depth map of a source view with the camera stored next to it (pmhip_scene_set_source_depth); depth None removes it."""
        if depth is None:
            self._chk(self._lib.pmhip_scene_set_source_depth(self._h, idx, None, 0, 0, None, None, None)); return
        d = np.ascontiguousarray(depth, np.float32)
        self._chk(self._lib.pmhip_scene_set_source_depth(self._h, idx, _fp(d), d.shape[1], d.shape[0], _dp(Kd), _dp(Rd), _dp(Cd)))

    def scene_estimate(self, view_ids, nGeometricIter=-1, params=None, sync=True):
        ids = np.ascontiguousarray(view_ids, np.int32)
        p = params or self.params
        self._chk(self._lib.pmhip_scene_estimate(self._h, ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids), C.byref(p),
                                                 C.c_int(nGeometricIter), C.c_int(1 if sync else 0)))

    def scene_commit_round(self):
        self._chk(self._lib.pmhip_scene_commit_round(self._h))

    def scene_reset_view(self, idx):
        self._chk(self._lib.pmhip_scene_reset_view(self._h, idx))

    def scene_set_maps(self, idx, depth=None, normal=None):
        d = None if depth is None else np.ascontiguousarray(depth, np.float32)
        n = None if normal is None else np.ascontiguousarray(normal, np.float32)
        self._chk(self._lib.pmhip_scene_set_maps(self._h, idx, _fp(d) if d is not None else None, _fp(n) if n is not None else None))

    def scene_set_mask(self, idx, mask):
        """Ignore mask of a view: (h, w) array, 0 = ignore the pixel (--ignore-mask-label); None removes it."""
        if mask is None:
            self._chk(self._lib.pmhip_scene_set_mask(self._h, idx, None)); return
        m = np.ascontiguousarray(np.asarray(mask) != 0, np.uint8)
        w, h = self.view_size(idx)
        if m.shape != (h, w):
            raise ValueError("mask must be (h, w)")
        self._chk(self._lib.pmhip_scene_set_mask(self._h, idx, m.ctypes.data_as(C.POINTER(C.c_uint8))))

    def scene_set_mask_mode(self, mode):
        self._chk(self._lib.pmhip_scene_set_mask_mode(self._h, int(mode)))

    def scene_set_conf(self, idx, conf):
        c = np.ascontiguousarray(conf, np.float32)
        self._chk(self._lib.pmhip_scene_set_conf(self._h, idx, _fp(c)))

    def view_size(self, idx):
        """(w, h) of a view's maps: the scene's, or the view's own (scene_set_view_sized)."""
        _, w, h = self._scene
        return getattr(self, "_sizes", {}).get(int(idx), (w, h))

    def scene_get_maps(self, idx):
        w, h = self.view_size(idx)
        d = np.zeros((h, w), np.float32); n = np.zeros((h, w, 3), np.float32); c = np.zeros((h, w), np.float32)
        self._chk(self._lib.pmhip_scene_get_maps(self._h, idx, _fp(d), _fp(n), _fp(c)))
        return d, n, c

    def scene_device_ptr(self, what, idx=0) -> int:
        return int(self._lib.pmhip_scene_device_ptr(self._h, what, idx) or 0)

    def scene_filter(self, view_ids, bAdjust=True, nMinViewsFilter=2, nMinViewsFilterAdjust=1, fDepthDiffThreshold=0.01, commit=True):
        """DepthMapsData::FilterDepthMap for these views (Scene::DenseReconstructionFilter); commit installs the results."""
        ids = np.ascontiguousarray(view_ids, np.int32)
        self._chk(self._lib.pmhip_scene_filter(self._h, ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids), 1 if bAdjust else 0,
                                               C.c_uint32(nMinViewsFilter), C.c_uint32(nMinViewsFilterAdjust), C.c_float(fDepthDiffThreshold), 1))
        if commit:
            self._chk(self._lib.pmhip_scene_filter_commit(self._h))

    def scene_remove_small_segments(self, view_ids, nSpeckleSize=100, fDepthDiffThreshold=0.01):
        ids = np.ascontiguousarray(view_ids, np.int32)
        self._chk(self._lib.pmhip_scene_remove_small_segments(self._h, ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids), C.c_uint32(nSpeckleSize), C.c_float(fDepthDiffThreshold)))

    def scene_gap_interpolation(self, view_ids, nIpolGapSize=7, fDepthDiffThreshold=0.01):
        ids = np.ascontiguousarray(view_ids, np.int32)
        self._chk(self._lib.pmhip_scene_gap_interpolation(self._h, ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids), C.c_uint32(nIpolGapSize), C.c_float(fDepthDiffThreshold)))

    def scene_set_color(self, idx, bgr):
        """8-bit BGR image of a view at depth-map resolution (only fusion with bEstimateColor reads it)."""
        b = np.ascontiguousarray(bgr, np.uint8)
        w, h = self.view_size(idx)
        if b.shape != (h, w, 3):
            raise ValueError("bgr must be (h, w, 3) uint8")
        self._chk(self._lib.pmhip_scene_set_color(self._h, idx, b.ctypes.data_as(C.POINTER(C.c_uint8))))

    def scene_fuse(self, order, nMinViewsFuse=2, fDepthDiffThreshold=0.01, fNormalDiffThreshold=25.0, bEstimateColor=True, bEstimateNormal=True):
        """DepthMapsData::FuseDepthMaps over the resident maps (libs/MVS/SceneDensify.cpp:1372-1650); `order` = views, best connected
        first.  Returns a dict with the fields of MVS::PointCloud: points, viewStart/views/weights (CSR), projs, colors (BGR), normals."""
        od = np.ascontiguousarray(order, np.int32)
        prm = PMHipFuseParams(nMinViewsFuse, fDepthDiffThreshold, fNormalDiffThreshold, 1 if bEstimateColor else 0, 1 if bEstimateNormal else 0)
        nP, nV, nD = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self._lib.pmhip_scene_fuse(self._h, od.ctypes.data_as(C.POINTER(C.c_int32)), len(od), C.byref(prm), C.byref(nP), C.byref(nV), C.byref(nD)))
        P, V = int(nP.value), int(nV.value)
        pts = np.zeros((P, 3), np.float32); vs = np.zeros(P + 1, np.uint32); views = np.zeros(V, np.uint32); wts = np.zeros(V, np.float32)
        projs = np.zeros((V, 2), np.uint16)
        cols = np.zeros((P, 3), np.uint8) if bEstimateColor else None
        nrm = np.zeros((P, 3), np.float32) if bEstimateNormal else None
        vp = lambda a, t: None if a is None else a.ctypes.data_as(C.POINTER(t))
        self._chk(self._lib.pmhip_scene_fuse_get(self._h, vp(pts, C.c_float), vp(vs, C.c_uint32), vp(views, C.c_uint32), vp(wts, C.c_float),
                                                 vp(projs, C.c_uint16), vp(cols, C.c_uint8), vp(nrm, C.c_float)))
        return dict(nPoints=P, nDepths=int(nD.value), points=pts, viewStart=vs, views=views, weights=wts, projs=projs, colors=cols, normals=nrm,
                    rounds=int(self._lib.pmhip_scene_fuse_rounds(self._h)))

    def scene_copy(self, what, first, count, device_ptr, to_engine):
        self._chk(self._lib.pmhip_scene_copy(self._h, what, first, count, C.c_void_p(device_ptr), 1 if to_engine else 0))

    def scene_images_updated(self):
        self._chk(self._lib.pmhip_scene_images_updated(self._h))

    def scene_maps_updated(self, first, count):
        """Depth maps of these views were written through scene_device_ptr(1, ...): see include/pmhip.h."""
        self._chk(self._lib.pmhip_scene_maps_updated(self._h, int(first), int(count)))

    def sync(self):
        self._chk(self._lib.pmhip_sync(self._h))

    def scene_bytes(self) -> int:
        """Device memory the resident scene holds right now (pmhip_scene_bytes)."""
        self._lib.pmhip_scene_bytes.restype = C.c_uint64
        return int(self._lib.pmhip_scene_bytes(self._h))

    def stream(self) -> int:
        return int(self._lib.pmhip_stream(self._h) or 0)

    def stats_reset(self, enable=True):
        self._chk(self._lib.pmhip_stats_reset(self._h, 1 if enable else 0))

    def stats_get(self) -> PMHipKernelStats:
        s = PMHipKernelStats()
        self._chk(self._lib.pmhip_stats_get(self._h, C.byref(s)))
        return s

    def prof_get(self, reset=True):
        out = (C.c_ulonglong * 16)()
        self._chk(self._lib.pmhip_prof_get(self._h, out, 1 if reset else 0))
        return list(out)

    # -- self-test hooks -----------------------------------------------------------------------
    def math_eval(self, kind, a, b=None):
        a = np.ascontiguousarray(a, np.float32); b = a if b is None else np.ascontiguousarray(b, np.float32)
        o = np.zeros_like(a)
        self._chk(self._lib.pmhip_math_eval(self._h, kind, _fp(a), _fp(b), _fp(o), C.c_size_t(a.size)))
        return o

    def resize(self, kind, img, arg=2):
        img = np.ascontiguousarray(img, np.float32); h, w = img.shape
        o = np.zeros((int(np.rint(h / arg)), int(np.rint(w / arg))) if kind == 0 else (h * 2, w * 2), np.float32)
        self._chk(self._lib.pmhip_resize(self._h, kind, _fp(img), w, h, arg, _fp(o)))
        return o

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pmhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

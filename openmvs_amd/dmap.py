"""ctypes binding of libdmapio.so: the OpenMVS `.dmap` depth-data file (include/dmapio.h).
Mirrors DepthData::Save / DepthData::Load (libs/MVS/DepthMap.cpp:234-268)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import build as _build


class DMapHeader(C.Structure):
    _fields_ = [("imageWidth", C.c_uint32), ("imageHeight", C.c_uint32), ("depthWidth", C.c_uint32), ("depthHeight", C.c_uint32),
                ("dMin", C.c_float), ("dMax", C.c_float), ("type", C.c_uint32), ("nIDs", C.c_uint32), ("IDs", C.c_uint32 * 256),
                ("K", C.c_double * 9), ("R", C.c_double * 9), ("C", C.c_double * 3), ("imageFileName", C.c_char * 1024)]


EXPORTS = ["dmap_write", "dmap_read_header", "dmap_read", "dimap_write", "dimap_read"]
_LIB = None


def load_library():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(_build.build_host_lib("libdmapio.so"))
        for n in EXPORTS:
            getattr(_LIB, n)
    return _LIB


def depth_file_name(i: int, ext: str = "dmap") -> str:
    """ComposeDepthFilePath, libs/MVS/DepthMap.h:72."""
    return "depth%04u.%s" % (i, ext)


def save(path, image_name, ids, image_size, K, R, Cc, dmin, dmax, depth, normal=None, conf=None, views=None):
    h = DMapHeader()
    depth = np.ascontiguousarray(depth, np.float32)
    h.imageWidth, h.imageHeight = int(image_size[0]), int(image_size[1])
    h.depthHeight, h.depthWidth = depth.shape
    h.dMin, h.dMax = float(dmin), float(dmax)
    h.nIDs = len(ids); h.IDs[:len(ids)] = [int(i) for i in ids]
    h.K[:] = np.asarray(K, np.float64).ravel(); h.R[:] = np.asarray(R, np.float64).ravel(); h.C[:] = np.asarray(Cc, np.float64).ravel()
    h.imageFileName = image_name.encode()
    fp = lambda a, t: None if a is None else np.ascontiguousarray(a, t).ctypes.data_as(C.c_void_p)
    keep = [None if a is None else np.ascontiguousarray(a, t) for a, t in ((normal, np.float32), (conf, np.float32), (views, np.uint8))]
    rc = load_library().dmap_write(str(path).encode(), C.byref(h), depth.ctypes.data_as(C.c_void_p),
                                   *[None if k is None else k.ctypes.data_as(C.c_void_p) for k in keep])
    if rc != 0:
        raise IOError("dmap_write failed: %d" % rc)


def load(path, flags: int = 15) -> dict:
    lib = load_library()
    h = DMapHeader()
    rc = lib.dmap_read_header(str(path).encode(), C.byref(h))
    if rc != 0:
        raise IOError("invalid depth-data file '%s' (%d)" % (path, rc))
    hh, ww = h.depthHeight, h.depthWidth
    out = dict(image_width=h.imageWidth, image_height=h.imageHeight, depth_width=ww, depth_height=hh, depth_min=h.dMin, depth_max=h.dMax,
               file_name=h.imageFileName.decode(), reference_view_id=h.IDs[0], neighbor_view_ids=[h.IDs[i] for i in range(1, h.nIDs)],
               K=np.array(h.K[:]).reshape(3, 3), R=np.array(h.R[:]).reshape(3, 3), C=np.array(h.C[:]),
               has_normal=bool(h.type & 2), has_conf=bool(h.type & 4), has_views=bool(h.type & 8))
    d = np.zeros((hh, ww), np.float32) if flags & 1 else None
    n = np.zeros((hh, ww, 3), np.float32) if (h.type & 2 and flags & 2) else None
    c = np.zeros((hh, ww), np.float32) if (h.type & 4 and flags & 4) else None
    v = np.zeros((hh, ww, 4), np.uint8) if (h.type & 8 and flags & 8) else None
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib.dmap_read(str(path).encode(), C.byref(h), C.c_uint(flags), p(d), p(n), p(c), p(v))
    if rc != 0:
        raise IOError("dmap_read failed: %d" % rc)
    for k, a in (("depth_map", d), ("normal_map", n), ("confidence_map", c), ("views_map", v)):
        if a is not None:
            out[k] = a
    return out


# ---- .dimap (disparity data of the SGM path) ---------------------------------------------------------------------------------------
def save_dimap(path, image_size, H, Q, subpixel_steps, disparity, cost=None):
    """SemiGlobalMatcher::ExportDisparityDataRawFull (libs/MVS/SemiGlobalMatcher.cpp:2124-2138): valid-grid maps in, bordered maps on disk."""
    d = np.ascontiguousarray(disparity, np.int16); c = None if cost is None else np.ascontiguousarray(cost, np.uint16)
    hh = np.ascontiguousarray(H, np.float64); qq = np.ascontiguousarray(Q, np.float64)
    rc = load_library().dimap_write(str(path).encode(), int(image_size[0]), int(image_size[1]), hh.ctypes.data_as(C.POINTER(C.c_double)), qq.ctypes.data_as(C.POINTER(C.c_double)),
                                    C.c_int16(subpixel_steps), d.ctypes.data_as(C.c_void_p), None if c is None else c.ctypes.data_as(C.c_void_p), d.shape[1], d.shape[0])
    if rc != 0:
        raise IOError("dimap_write failed: %d" % rc)


def load_dimap(path) -> dict:
    """ImportDisparityDataRawFull (:2178-2188)."""
    lib = load_library()
    iw, ih, w, h, hc = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(); st = C.c_int16()
    H = np.zeros((3, 3)); Q = np.zeros((4, 4))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    args = (str(path).encode(), C.byref(iw), C.byref(ih), dp(H), dp(Q), C.byref(st), C.byref(w), C.byref(h), C.byref(hc))
    if lib.dimap_read(*args, None, None) != 0:
        raise IOError("invalid disparity-data file '%s'" % path)
    d = np.zeros((h.value, w.value), np.int16); c = np.zeros((h.value, w.value), np.uint16) if hc.value else None
    if lib.dimap_read(*args, d.ctypes.data_as(C.c_void_p), None if c is None else c.ctypes.data_as(C.c_void_p)) != 0:
        raise IOError("dimap_read failed")
    return dict(image_size=(iw.value, ih.value), H=H, Q=Q, subpixel_steps=int(st.value), disparity=d, cost=c)

"""Malformed / hostile .mvs archives must come back as an error code from the C ABI, never as a crash, an exception across `extern "C"`, or an
out-of-bounds index handed to view selection (mvs_front.cpp reader)."""
import ctypes as C
import os
import struct

import pytest

from openmvs_amd import mvsfront as mf

SCENE = os.path.join(os.path.dirname(__file__), "data", "scene", "scene.mvs")


def _load(path):
    lib = mf.load_library()
    h = C.c_void_p()
    lib.mvsf_load.restype = C.c_int
    rc = lib.mvsf_load(path.encode(), C.byref(h))
    if rc == 0:
        lib.mvsf_free(h)
    return rc


def test_truncations_are_rejected(tmp_path):
    raw = open(SCENE, "rb").read()
    assert _load(SCENE) == 0
    for cut in (5, 11, 40, 200, len(raw) // 3, len(raw) // 2, len(raw) - 9):
        p = tmp_path / ("cut%d.mvs" % cut)
        p.write_bytes(raw[:cut])
        assert _load(str(p)) == -2, cut


def test_huge_counts_do_not_wrap_or_throw(tmp_path):
    raw = bytearray(open(SCENE, "rb").read())
    hdr = 12 if raw[:4] == b"MVSI" else 0
    # number of platforms, then the first name length: 2^64-1 and values whose product with the record size wraps
    for off in (hdr, hdr + 8):
        for val in (0xFFFFFFFFFFFFFFFF, 0x2000000000000001, 0x1555555555555556, 1 << 40):
            bad = bytearray(raw); bad[off:off + 8] = struct.pack("<Q", val)
            p = tmp_path / "huge.mvs"; p.write_bytes(bad)
            assert _load(str(p)) == -2, (off, hex(val))


def test_out_of_range_view_index_is_rejected(tmp_path):
    """A point that names a view >= number of images used to index cams[] / scores[] out of bounds in selectNeighborViews."""
    from openmvs_amd import mvsi
    sc = mvsi.load(SCENE)
    n_img = len(sc.images)
    raw = bytearray(open(SCENE, "rb").read())
    import numpy as np
    X0 = np.asarray(sc.vertices[0], np.float32).tobytes()      # the vertex block starts with the first point's coordinates
    assert raw.find(X0) > 0
    o = raw.find(X0) + 12
    m = struct.unpack_from("<Q", raw, o)[0]
    assert 0 < m < 64
    struct.pack_into("<I", raw, o + 8, n_img + 7)
    p = tmp_path / "badview.mvs"; p.write_bytes(raw)
    assert _load(str(p)) == -2

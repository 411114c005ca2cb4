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


def test_stored_view_scores_with_huge_counts_or_foreign_ids(tmp_path):
    """Archives of version 7 carry every image's view scores, which the reader now keeps as neighbour lists: a count that does not fit the file, or an ID that is not an
    image of the scene, is a format error; truncations anywhere in the block are too.  (800 random text files and 600 randomly damaged version-7 archives went through an
    AddressSanitizer / UBSan build of the two sources without a report: tools/README.md.)"""
    import numpy as np
    from openmvs_amd import mvsi, views
    sc = mvsi.load(SCENE)
    cams = views.Cameras(sc)
    for i, im in enumerate(sc.images):
        ok, nb, pts, avg = views.select_neighbor_views(sc, cams, i)
        im.view_scores, im.avg_depth = nb, avg
    p7 = tmp_path / "v7.mvs"
    mvsi.save(str(p7), sc, version=7)
    assert _load(str(p7)) == 0
    raw = bytearray(p7.read_bytes())
    first = sc.images[0].view_scores[:1].tobytes()
    at = bytes(raw).find(first)
    assert at > 8 and struct.unpack_from("<Q", raw, at - 8)[0] == len(sc.images[0].view_scores)
    for val in (0xFFFFFFFFFFFFFFFF, 0x0AAAAAAAAAAAAAAB, 1 << 33, len(raw)):                      # the count of image 0's view scores
        bad = bytearray(raw); struct.pack_into("<Q", bad, at - 8, val)
        p = tmp_path / "count.mvs"; p.write_bytes(bad)
        assert _load(str(p)) == -2, hex(val)
    bad = bytearray(raw); struct.pack_into("<I", bad, at, len(sc.images))                       # a neighbour that is not an image of the scene
    p = tmp_path / "id.mvs"; p.write_bytes(bad)
    assert _load(str(p)) == -2
    for cut in (at - 4, at + 3, at + 24 * len(sc.images[0].view_scores) - 1):
        p = tmp_path / "cut.mvs"; p.write_bytes(raw[:cut])
        assert _load(str(p)) == -2, cut

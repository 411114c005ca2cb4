"""Writes tests/golden/depth0007.dmap with OUR writer and checks it with the REFERENCE's own reader
(scripts/python/MvsUtils.py:9-70, importable only in the build container).  Run from the repo root."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from openmvs_amd import dmap  # noqa: E402


def content():
    r = np.random.RandomState(42)
    h, w = 12, 20
    depth = r.rand(h, w).astype(np.float32) * 5; depth[::4] = 0
    normal = r.randn(h, w, 3).astype(np.float32); normal /= np.linalg.norm(normal, axis=-1, keepdims=True)
    conf = r.rand(h, w).astype(np.float32)
    views = r.randint(0, 255, (h, w, 4)).astype(np.uint8)
    K = np.array([[672.62, 0, 312.168], [0, 672.62, 226.712], [0, 0, 1]]); R = np.linalg.qr(r.randn(3, 3))[0]; C = r.randn(3)
    return dict(image_name="images/00007.jpg", ids=[7, 3, 9, 12], image_size=(40, 24), K=K, R=R, Cc=C, dmin=0.25, dmax=7.5,
                depth=depth, normal=normal, conf=conf, views=views)


if __name__ == "__main__":
    c = content()
    path = os.path.join(HERE, dmap.depth_file_name(7))
    dmap.save(path, **c)
    sys.path.insert(0, "/root/reference/scripts/python")
    from MvsUtils import loadDMAP  # the reference's independent statement of the format
    d = loadDMAP(path)
    assert d["file_name"] == c["image_name"] and d["reference_view_id"] == 7 and list(d["neighbor_view_ids"]) == [3, 9, 12]
    assert (d["image_width"], d["image_height"], d["depth_width"], d["depth_height"]) == (40, 24, 20, 12)
    assert np.array_equal(d["depth_map"], c["depth"]) and np.array_equal(d["normal_map"], c["normal"]) and np.array_equal(d["confidence_map"], c["conf"])
    assert np.array_equal(d["views_map"], c["views"]) and np.array_equal(d["K"], c["K"]) and np.array_equal(d["R"], c["R"]) and np.array_equal(d["C"], c["Cc"])
    assert abs(d["depth_min"] - 0.25) < 1e-7 and abs(d["depth_max"] - 7.5) < 1e-7
    print("reference reader parsed our file identically:", path, os.path.getsize(path), "bytes")

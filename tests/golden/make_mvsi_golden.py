"""Generates tests/golden/mvsi_golden.json by parsing tests/data/scene/scene.mvs with the REFERENCE's own reader
(`/root/reference/scripts/python/MvsUtils.py::loadMVSInterface`).  Runs only in the build container (the GPU box has no
/root/reference); the JSON it writes is the committed fixture that tests/test_mvsi_views.py checks `openmvs_amd.mvsi` against."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/scripts/python")
import MvsUtils  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
d = MvsUtils.loadMVSInterface(os.path.join(HERE, "..", "data", "scene", "scene.mvs"))


def sha(a, dt):
    return hashlib.sha256(np.ascontiguousarray(a, dt).tobytes()).hexdigest()


out = {
    "version": d["project_stream_version"],
    "platforms": [{"name": p["name"],
                   "cameras": [{"name": c["name"], "width": c["width"], "height": c["height"], "K": c["K"],
                                "poses": c["poses"]} for c in p["cameras"]]} for p in d["platforms"]],
    "images": d["images"],
    "n_vertices": len(d["vertices"]),
    "vertices_sha256": sha([v["X"] for v in d["vertices"]], "<f4"),
    "views_per_vertex_sha256": sha([len(v["views"]) for v in d["vertices"]], "<i8"),
    "view_image_ids_sha256": sha([w["image_id"] for v in d["vertices"] for w in v["views"]], "<u4"),
    "view_confidences_sha256": sha([w["confidence"] for v in d["vertices"] for w in v["views"]], "<f4"),
    "vertices_color_sha256": sha(d["vertices_color"], "u1"),
    "n_normals": len(d["vertices_normal"]), "n_lines": len(d["lines"]),
    "first_vertex": d["vertices"][0], "last_vertex": d["vertices"][-1],
    "transform": d["transform"], "obb": d["obb"],
}
with open(os.path.join(HERE, "mvsi_golden.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote mvsi_golden.json:", out["n_vertices"], "vertices")

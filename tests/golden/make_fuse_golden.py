"""Generates tests/golden/fuse_golden_96x64.npz: the fused cloud of the CPU oracle (oracle/fuse_oracle.cpp) over the cross-view-filtered maps of
filter_golden_96x64.npz (5 views, 96x64).  Colours come from a deterministic synthetic BGR image derived from the gray images.  Run from the repo root."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle as po  # noqa: E402

g = np.load(os.path.join(HERE, "pm_golden_96x64.npz"))
f = np.load(os.path.join(HERE, "filter_golden_96x64.npz"))
nv = int(g["n_views"])
gray = g["gray"]
bgr = np.stack([np.clip(np.stack([gray * 255, gray * 200 + 20, 255 - gray * 180], -1), 0, 255).astype(np.uint8)[v] for v in range(nv)])
depth, conf, normal = f["filt_depth"], f["filt_conf"], f["gap_normal"]
nbrs = [list(g["neighbors"][v]) for v in range(nv)]
out = {}
for tag, kw in (("fuse2", dict(nMinViewsFuse=2)), ("fuse3", dict(nMinViewsFuse=3, fNormalDiffThreshold=40.0))):
    r = po.fuse_depth_maps(list(depth), list(normal), list(conf), list(bgr), g["K"], g["R"], g["C"], nbrs, **kw)
    for k in ("points", "viewStart", "views", "weights", "projs", "colors", "normals"):
        out[tag + "_" + k] = r[k]
    out[tag + "_nDepths"] = r["nDepths"]
    print(tag, r["nPoints"], "points from", r["nDepths"], "depths")
np.savez_compressed(os.path.join(HERE, "fuse_golden_96x64.npz"), bgr=bgr, **out)

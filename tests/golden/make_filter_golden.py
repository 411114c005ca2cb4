"""Generates tests/golden/filter_golden_96x64.npz from the CPU oracles of the post-filters (run from the repo root).
Inputs: the photometric depth/normal/conf maps of the 5 views of pm_golden_96x64.npz recomputed by the PatchMatch oracle."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle as po  # noqa: E402

g = np.load(os.path.join(HERE, "pm_golden_96x64.npz"))
nv = int(g["n_views"])
maps = []
for v in range(nv):
    ids = [v] + list(g["neighbors"][v])
    views, keep = po.make_views(g["gray"], g["K"], g["R"], g["C"], ids)
    maps.append(po.estimate_depth_map(views, len(ids), float(g["dmin"][v]), float(g["dmax"][v]), po.default_opt(seed=int(g["seed"]), viewID=v)))
depth = np.stack([m[0] for m in maps]); normal = np.stack([m[1] for m in maps]); conf = np.stack([m[2] for m in maps])
assert np.array_equal(depth, g["depth_photo_all"])
sp = [po.remove_small_segments(depth[v], normal[v], conf[v], nSpeckleSize=30) for v in range(nv)]
gp = [po.gap_interpolation(*sp[v]) for v in range(nv)]
d2 = np.stack([x[0] for x in gp]); c2 = np.stack([x[2] for x in gp])
fl = [po.filter_depth_map(d2, c2, g["K"], g["R"], g["C"], v, list(g["neighbors"][v]), g["dmin"][v], g["dmax"][v]) for v in range(nv)]
fs = [po.filter_depth_map(d2, c2, g["K"], g["R"], g["C"], v, list(g["neighbors"][v]), g["dmin"][v], g["dmax"][v], bAdjust=False) for v in range(nv)]
np.savez_compressed(os.path.join(HERE, "filter_golden_96x64.npz"), normal=normal, conf=conf,
                    speckle_depth=np.stack([x[0] for x in sp]), gap_depth=d2, gap_normal=np.stack([x[1] for x in gp]), gap_conf=c2,
                    filt_depth=np.stack([x[1] for x in fl]), filt_conf=np.stack([x[2] for x in fl]),
                    strict_depth=np.stack([x[1] for x in fs]), strict_conf=np.stack([x[2] for x in fs]))
print("removed by speckle filter:", int(((depth > 0) & (np.stack([x[0] for x in sp]) == 0)).sum()), "filled gaps:", int(((d2 > 0) & (np.stack([x[0] for x in sp]) == 0)).sum()),
      "kept by cross-view filter:", int((np.stack([x[1] for x in fl]) > 0).sum()), "of", int((d2 > 0).sum()))

"""Generates tests/golden/pm_golden_96x64.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference ships no golden vectors for this path and
cannot be built or imported here, so these pin the *oracle* (and, through the -m gpu tests, the
HIP engine) against regressions; they are not outputs of the reference binary."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from openmvs_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

N_VIEWS, N_SRC, SEED = 5, 4, 7
sc = synth.make_scene(N_VIEWS, 96, 64, n_src=N_SRC)
out = dict(n_views=N_VIEWS, n_src=N_SRC, seed=SEED, gray=sc.gray, K=sc.K, R=sc.R, C=sc.C, neighbors=sc.neighbors,
           dmin=sc.dmin, dmax=sc.dmax, diameter=sc.diameter)
# photometric pass for every view (needed as inputs of the geometric round of view 0)
photo = {}
for v in range(N_VIEWS):
    ids = [v] + list(sc.neighbors[v])
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
    opt = po.default_opt(seed=SEED, viewID=v)
    photo[v] = po.estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), opt)
out["depth_photo"], out["normal_photo"], out["conf_photo"] = photo[0]
out["depth_photo_all"] = np.stack([photo[v][0] for v in range(N_VIEWS)])
ids = [0] + list(sc.neighbors[0])
views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps={v: photo[v][0] for v in range(N_VIEWS)})
opt = po.default_opt(seed=SEED, viewID=0)
d, n, c = po.estimate_depth_map(views, len(ids), float(sc.dmin[0]), float(sc.dmax[0]), opt, geo_iter=0, depth=photo[0][0], normal=photo[0][1])
out["depth_geo0"], out["normal_geo0"], out["conf_geo0"] = d, n, c
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pm_golden_96x64.npz"), **out)
print("written; valid photo %.3f geo %.3f" % ((photo[0][0] > 0).mean(), (d > 0).mean()))

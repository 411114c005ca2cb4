"""SGM timing probe (GPU box): BASELINE config 4, 2048x1536, D = 64 / 128, + ragged tSGM-like ranges."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvs_amd import sgm
from tests import sgm_cases as sc
w, h = 2048, 1536
lb, lg, rg = sc.stereo_pair(w, h, 21, seed=9)
m = sgm.SemiGlobalMatcherHIP(0)
for kind, lo, hi in (("uniform", 0, 64), ("uniform", 0, 128), ("ragged", 0, 64)):
    px, n, mx = sc.ranges(w, h, kind, lo, hi)
    m.set_problem(lb, lg, rg, px, n, mx)
    m.Match()
    m.stats_reset(True); t = time.time()
    reps = 4
    for _ in range(reps): m.Match(sync=False)
    m.sync(); dt = (time.time() - t) / reps
    s = m.stats_get()
    gb = 43.0 * n / 1e9
    print("%s D<=%d numCosts %.1fM: %.2f ms/match (cost %.2f aggr %.2f wta %.2f) -> %.1f GB/s on the 43 B/cost model (aggr alone: %.1f GB/s of 40 B/cost)" % (
        kind, hi - lo, n / 1e6, dt * 1e3, s.costMs / reps, s.aggrMs / reps, s.wtaMs / reps, gb / dt, 40.0 * n / 1e9 / (s.aggrMs / reps / 1e3)), flush=True)

# narrow tSGM-like ranges (3..12 disparities per pixel): the wide mapping against the 16-lane sub-group mapping
px, n, mx = sc.ranges(w, h, "ragged", 0, 12)
m.set_problem(lb, lg, rg, px, n, mx)
for sub in (0, 8, 16, 32):
    m.set_sub_group_kernels(sub)
    m.Match()
    d_ref = m.results()[0] if not sub else d_ref
    same = bool(np.array_equal(m.results()[0], d_ref))
    m.stats_reset(True); t = time.time()
    for _ in range(4): m.Match(sync=False)
    m.sync(); dt = (time.time() - t) / 4
    s = m.stats_get()
    print("narrow ranges numCosts %.1fM, %s kernels: %.2f ms/match (cost %.2f aggr %.2f wta %.2f)%s" % (
        n / 1e6, ("%d-lane sub-group" % sub) if sub else "wide", dt * 1e3, s.costMs / 4, s.aggrMs / 4, s.wtaMs / 4, "" if not sub else "  identical: %s" % same), flush=True)
m.set_sub_group_kernels(False)

# the whole coarse-to-fine loop for a rectified pair: one resident call vs the step-wise loop through host buffers
from openmvs_amd import tsgm
from openmvs_amd.patchmatch import PatchMatchHIP
from tests.tsgm_backends import DeviceBackend
rb = np.roll(lb, 21, axis=1)
mask = np.full((h, w), 255, np.uint8)
m.tsgm_match(lb, rb, lg, rg, mask, mask, min_resolution=320)
t = time.time(); d1, c1, lv = m.tsgm_match(lb, rb, lg, rg, mask, mask, min_resolution=320); t1 = time.time() - t
e = PatchMatchHIP(0)
t = time.time(); d2, c2, _ = tsgm.tsgm_match(DeviceBackend(m, e), lb, lg, rb, rg, mask, mask, min_resolution=320); t2 = time.time() - t
print("tSGM %dx%d, %d levels: resident call %.1f ms, step-wise loop %.1f ms, identical: %s, valid %.3f" % (
    w, h, lv, t1 * 1e3, t2 * 1e3, bool(np.array_equal(d1, d2) and np.array_equal(c1, c2)), float((d1 != 32767).mean())), flush=True)

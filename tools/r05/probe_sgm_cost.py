"""Round-5 SGM probe: SemiGlobalMatcher::Match at 2048x1536 (uniform D = 64 / 128, ragged D <= 64) with the library SGMHIP_LIB names (default: the tree's), HIP-event split per phase,
and a digest of the disparity map (equal digests = equal results across libraries)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    dig = hashlib.sha1(m.results()[0].tobytes()).hexdigest()[:10]
    best = None
    for rep in range(3):
        m.stats_reset(True); t = time.time()
        reps = 8
        for _ in range(reps): m.Match(sync=False)
        m.sync(); dt = (time.time() - t) / reps
        s = m.stats_get()
        if best is None or dt < best[0]: best = (dt, s.costMs / reps, s.aggrMs / reps, s.wtaMs / reps)
    dt, c, a, wt = best
    print("%-8s D<=%-3d numCosts %.1fM: %.3f ms/match (cost %.3f aggr %.3f wta %.3f) -> %.0f GB/s on the 43 B/cost model   disparity sha1 %s" % (kind, hi - lo, n / 1e6, dt * 1e3, c, a, wt, 43.0 * n / 1e9 / dt, dig), flush=True)

"""SGM aggregation rate against the size of the cost / sum volumes (GPU box): is the 8-path kernel bound by HBM traffic that a cache-resident band would avoid?
Same width and disparity range, heights from 192 rows (25 + 50 MB of volumes: inside the 256 MB MALL) to 1536 (200 + 400 MB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvs_amd import sgm
from tests import sgm_cases as sc
w = 2048
m = sgm.SemiGlobalMatcherHIP(0)
for h in (198, 390, 774, 1536):
    lb, lg, rg = sc.stereo_pair(w, h, 21, seed=9)
    px, n, mx = sc.ranges(w, h, "uniform", 0, 64)
    m.set_problem(lb, lg, rg, px, n, mx)
    m.Match()
    m.stats_reset(True)
    reps = 8
    for _ in range(reps): m.Match(sync=False)
    m.sync()
    s = m.stats_get()
    a = s.aggrMs / reps
    print("2048x%d D=64: volumes %.0f + %.0f MB, aggregation %.3f ms = %.2f ns per kilo-entry (%.0f GB/s of 40 B/entry), cost %.3f ms" % (
        h, n / 1e6, 2 * n / 1e6, a, a * 1e6 / (n / 1e3), 40.0 * n / 1e9 / (a / 1e3), s.costMs / reps), flush=True)

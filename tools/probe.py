"""Quick timing probe: photometric + geometric rounds for B views at WxH (GPU box)."""
import argparse, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvs_amd import synth
from openmvs_amd.patchmatch import PatchMatchHIP, default_params
ap = argparse.ArgumentParser(); ap.add_argument("--views", type=int, default=9); ap.add_argument("--batch", type=int, default=9)
ap.add_argument("--w", type=int, default=1920); ap.add_argument("--h", type=int, default=1080); ap.add_argument("--geo", type=int, default=2)
ap.add_argument("--reps", type=int, default=1)
a = ap.parse_args()
t = time.time(); sc = synth.make_scene(a.views, a.w, a.h, n_src=8, device="cuda", gray_only=True); print("scene %.1fs" % (time.time() - t), flush=True)
e = PatchMatchHIP(0); e.Init(True); e.scene_load(sc, 2)
p = default_params(seed=1, nEstimationGeometricIters=a.geo)
ids = list(range(a.batch))
for rep in range(a.reps + 1):
    for v in ids: e.scene_reset_view(v)
    e.stats_reset(True); e.sync(); t0 = time.time()
    e.scene_estimate(ids, -1, p); e.sync(); t1 = time.time()
    for g in range(a.geo):
        e.scene_commit_round(); e.scene_estimate(ids, g, p)
    e.sync(); t2 = time.time()
    s = e.stats_get()
    mp = a.batch * a.w * a.h / 1e6
    print("rep %d: photo %.3fs geo %.3fs total %.3fs -> %.2f Mpix/s | sweep %.1f ms over %d launches (%.1f us/launch), init %.1f ms, sweep GB/s %.1f" % (
        rep, t1 - t0, t2 - t1, t2 - t0, mp / (t2 - t0), s.sweepMs, s.sweepLaunches, 1e3 * s.sweepMs / max(1, s.sweepLaunches), s.initMs, s.sweepBytes / 1e6 / max(s.sweepMs, 1e-9)), flush=True)
d, n, c = e.scene_get_maps(ids[len(ids) // 2]); m = d > 0; gt = sc.gt_depth[ids[len(ids) // 2]]
print("valid %.3f median rel err %.2e" % (m.mean(), np.median(np.abs(d[m] - gt[m]) / gt[m])))

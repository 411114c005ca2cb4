"""Full schedule (photometric pass over the pyramid + 2 geometric rounds) for small batches of reference views of a 13-view 1920x1080 scene, with the
regular sweep kernel and with the speculative kernels (PMHipTuning::wideMaxViews): seconds per batch and Mpix/s.  Decides PMHIP_DEFAULT_WIDE.
    python tools/small_batch_probe.py [batch sizes ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmvs_amd import synth
from openmvs_amd.patchmatch import PatchMatchHIP, default_params
sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 13]
V, W, H = int(os.environ.get("PROBE_VIEWS", "13")), 1920, 1080   # PROBE_VIEWS=25: the batch sizes around the speculative kernels' upper limit
sc = synth.make_scene(V, W, H, n_src=8, device="cuda", gray_only=True)
p = default_params(seed=1)
ref = {}
for band, wide in (("0", "0"), ("0", "64")):
    e = PatchMatchHIP(0); e.tuning(wideMaxViews=-1 if wide == "0" else int(wide)); e.Init(True); e.scene_load(sc, 2)
    allv = list(range(V))
    for b in sizes:
        ids = [(4 + k) % V for k in range(b)]
        best = 1e9
        for rep in range(2):
            for v in allv: e.scene_reset_view(v)
            e.sync(); t = time.time()
            e.scene_estimate(ids, -1, p, sync=False)
            for g in range(2):
                e.scene_commit_round(); e.scene_estimate(ids, g, p, sync=False)
            e.sync(); best = min(best, time.time() - t)
        d = e.scene_get_maps(ids[0])[0]
        same = ""
        if wide == "0": ref[b] = d
        else: same = "  identical to the regular kernel: %s" % bool(np.array_equal(d, ref[b]))
        print("wideMaxViews=%-2s batch %2d views: %.3f s  -> %.2f Mpix/s%s" % (wide, b, best, b * W * H / best / 1e6, same), flush=True)
    e.close()

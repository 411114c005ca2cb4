"""In-kernel phase breakdown of the sweep kernel (needs libs built with -DPM_PROFILE; see tools/README)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvs_amd import synth
from openmvs_amd.patchmatch import PatchMatchHIP, default_params
views = int(sys.argv[1]) if len(sys.argv) > 1 else 32
only = [int(a) for a in sys.argv[2:]]          # estimate only these views (e.g. "9 4": BASELINE config 2, one depth map of a 9-view scene)
sc = synth.make_scene(views, 1920, 1080, n_src=8, device="cuda", gray_only=True)
e = PatchMatchHIP(0); e.Init(True); e.scene_load(sc, 2)
p = default_params(seed=1); ids = only or list(range(views))
for rep in range(2):
    for v in ids: e.scene_reset_view(v)
    e.prof_get(True); e.sync(); t0 = time.time()
    e.scene_estimate(ids, -1, p)
    for g in range(2): e.scene_commit_round(); e.scene_estimate(ids, g, p)
    e.sync(); dt = time.time() - t0
    c = e.prof_get(True)
    import ctypes
    hist = (ctypes.c_ulonglong * 17)()
    if hasattr(e._lib, 'pmhip_prof_hist'): e._lib.pmhip_prof_hist(e._h, hist, 1)
names = ["-", "hyp-gen", "smooth", "homography", "taps", "epilogue", "aggr+accept"]
tot = (sum(c[0:7]) + c[12]) or 1
print(os.environ.get("PMHIP_LIB", "default"), "views", len(ids), "of", views, "%.2f s -> %.2f Mpix/s" % (dt, len(ids) * 1920 * 1080 / dt / 1e6))
print("  wave-visits %d, cycles per wave-visit %.0f (s_memtime), hypotheses x active lanes per wave-visit %.1f" % (c[9], tot / max(1, c[9]), c[8] / max(1, c[9])))
print("  %-12s %5.1f %%   %8.0f cycles/wave-visit" % ("head", 100.0 * c[12] / tot, c[12] / max(1, c[9])))
print("  %-12s %5.1f %%   %8.0f cycles/wave-visit   (the speculative kernels' whole head; pm_sweep2_kernel: window staging, unused)" % ("setup", 100.0 * c[0] / tot, c[0] / max(1, c[9])))
for i, n in enumerate(names):
    if i: print("  %-12s %5.1f %%   %8.0f cycles/wave-visit" % (n, 100.0 * c[i] / tot, c[i] / max(1, c[9])))
if sum(hist):
    tot_t = sum(hist)
    print("  trips of a wave-visit by pixels taking part (of 16 with 4 lanes per pixel): " + " ".join("%d:%.1f%%" % (k, 100.0 * hist[k] / tot_t) for k in range(17) if hist[k]))
    print("  trips with <= 8 pixels taking part: %.1f %%, <= 4: %.1f %%; mean pixels per trip %.2f" % (100.0 * sum(hist[:9]) / tot_t, 100.0 * sum(hist[:5]) / tot_t, sum(k * hist[k] for k in range(17)) / tot_t))

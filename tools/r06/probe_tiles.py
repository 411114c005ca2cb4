"""Round-6 probe: full schedule (photometric pass + 2 geometric rounds) of the first B views of a V-view 1920x1080 scene with the reference's sweep and with the opt-in tiled
sweeps (pmhip_set_sweep_tiles), one fresh engine per configuration.
    python tools/r06/probe_tiles.py V B "NAME:tw=64,th=64,groups=2,lib=libpmhip.so" ...
Prints seconds per step, Mpix/s, and for the tiled configurations how view 0's depth map compares with the first configuration's (they are different estimators) and with the
ground truth."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from openmvs_amd import patchmatch, synth
from openmvs_amd.patchmatch import PatchMatchHIP, default_params

V, B = int(sys.argv[1]), int(sys.argv[2]); W, H = 1920, 1080
configs = []
for a in sys.argv[3:]:
    name, _, kv = a.partition(":")
    configs.append((name, dict(x.split("=") for x in kv.split(",") if x)))
dev = torch.device("cuda", 0)
sc = synth.make_scene_torch(V, W, H, n_src=8, device=dev, gt_views=1)
gray = sc["gray"]; torch.cuda.synchronize()
gt = sc["gt_depth"][0].cpu().numpy() if "gt_depth" in sc else None
diam = float(sc.get("diameter", 1.0))
p = default_params(seed=1, nEstimationGeometricIters=2)
ref = None
steps = int(os.environ.get("PROBE_STEPS", "2"))
for name, kv in configs:
    lib = kv.get("lib")
    if lib:
        os.environ["PMHIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(patchmatch.__file__)), lib); patchmatch._LIB = None
    e = PatchMatchHIP(0); e.Init(True); e.scene_create(V, W, H, 2)
    for i in range(V):
        e.scene_set_view(i, None, sc["K"][i], sc["R"][i], sc["C"][i], float(sc["dmin"][i]), float(sc["dmax"][i]), sc["neighbors"][i])
    e.scene_copy(0, 0, V, gray.data_ptr(), True); e.sync()
    if "groups" in kv: e.tuning(viewGroups=int(kv["groups"]))
    if "widepx" in kv: e.tuning(widePixels=int(kv["widepx"]))
    tw, th = int(kv.get("tw", 0)), int(kv.get("th", 0))
    if tw: e.set_sweep_tiles(tw, th)
    ids = list(range(B)); allv = list(range(V))
    best = 1e9
    for rep in range(1 + steps):
        for v in allv: e.scene_reset_view(v)
        if B < V:                                            # the views outside the batch are read as sources in the geometric rounds: give them maps once
            pass
        e.sync(); t = time.perf_counter()
        e.scene_estimate(ids, -1, p, sync=False)
        for g in range(2):
            e.scene_commit_round(); e.scene_estimate(ids, g, p, sync=False)
        e.sync(); dt = time.perf_counter() - t
        if rep and dt < best: best = dt
    d = e.scene_get_maps(0)[0]
    if ref is None: ref = d
    m = (d > 0) & (ref > 0)
    ad = np.abs(d[m].astype(np.float64) - ref[m]) / diam
    vs_first = "rmse %.2e med %.2e p95 %.2e only-one %d" % (float(np.sqrt(np.mean(ad ** 2))), float(np.median(ad)), float(np.percentile(ad, 95)), int(((d > 0) != (ref > 0)).sum())) if ad.size else "-"
    vs_gt = ""
    if gt is not None:
        mg = d > 0
        vs_gt = " | vs gt rmse/diam %.3e valid %.3f" % (float(np.sqrt(np.mean((d[mg].astype(np.float64) - gt[mg]) ** 2))) / diam, float(mg.mean()))
    print("%-22s %-40s %.3f s/step  %.2f Mpix/s | view 0 vs first config: %s%s" % (name, kv, best, B * W * H / best / 1e6, vs_first, vs_gt), flush=True)
    e.close()

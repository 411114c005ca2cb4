"""BASELINE config 5 at a slice of its scale on ONE GPU: V views of 3840x2160, full schedule (photometric pyramid + 2 geometric rounds), speckle / gap
filters, cross-view filter, then FuseDepthMaps on the same device -- per-stage seconds, HBM in use after each stage, points fused.
    python tools/config5_probe.py [views=32] [width=3840] [height=2160]
Feeds DESIGN.md's model of the 300-view run (the fuse step is single-rank and sequential over the scene: its time and memory at 300 x 4K are
extrapolated from here)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openmvs_amd import synth
from openmvs_amd.patchmatch import PatchMatchHIP, default_params
V = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160


def used_gb():
    free, total = torch.cuda.mem_get_info()
    return round((total - free) / 2 ** 30, 2)


out = {"views": V, "width": W, "height": H}
sc = synth.make_scene_torch(V, W, H, n_src=8, device="cuda", gt_views=0, want_bgr=False)
gray = sc["gray"]
e = PatchMatchHIP(0); e.Init(True); e.scene_create(V, W, H, 2)
for i in range(V):
    e.scene_set_view(i, None, sc["K"][i], sc["R"][i], sc["C"][i], float(sc["dmin"][i]), float(sc["dmax"][i]), sc["neighbors"][i])
e.scene_copy(0, 0, V, gray.data_ptr(), True); e.sync()
del gray; torch.cuda.empty_cache()
out["hbm_gb_after_upload"] = used_gb()
p = default_params(seed=1)
ids = list(range(V))
t = time.time(); e.scene_estimate(ids, -1, p, sync=False)
for g in range(2):
    e.scene_commit_round(); e.scene_estimate(ids, g, p, sync=False)
e.sync(); out["estimate_s"] = round(time.time() - t, 3); out["estimate_mpix_s"] = round(V * W * H / out["estimate_s"] / 1e6, 2)
out["hbm_gb_after_estimate"] = used_gb()
t = time.time(); e.scene_remove_small_segments(ids); e.scene_gap_interpolation(ids); e.sync(); out["speckle_gap_s"] = round(time.time() - t, 3)
t = time.time(); e.scene_filter(ids, True, 2, 1, 0.01, commit=True); e.sync(); out["cross_view_filter_s"] = round(time.time() - t, 3)
out["hbm_gb_after_filter"] = used_gb()
order = sorted(ids, key=lambda i: -len(sc["neighbors"][i]))
t = time.time(); pc = e.scene_fuse(order, bEstimateColor=False); out["fuse_s_incl_download"] = round(time.time() - t, 3)
out["fused_points"] = int(pc["nPoints"]); out["fused_depths"] = int(pc["nDepths"]); out["fuse_rounds"] = int(pc["rounds"])
out["hbm_gb_after_fuse"] = used_gb()
out["fuse_ns_per_depth"] = round(1e9 * out["fuse_s_incl_download"] / max(1, out["fused_depths"]), 2)
print(json.dumps(out))
e.close()

"""Counter-collection workload without torch in the profiled process (rocprofv3 --pmc crashed inside torch's own kernels on this pool):
    python tools/pmc_workload.py gen 100      # unprofiled: renders the bench scene on the GPU with torch, writes /tmp/pmc_scene.npy + .json
    rocprofv3 --pmc ... -- python tools/pmc_workload.py run [geo_iters]   # numpy + ctypes only: upload, one full photometric schedule (+ geo rounds)
The `run` leg prints the same sweep statistics bench.py reports (launches, algorithmic bytes per launch), so FETCH_SIZE per launch can be set against them."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mode = sys.argv[1]
if mode == "gen":
    import torch
    from openmvs_amd import synth
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    sc = synth.make_scene_torch(V, 1920, 1080, n_src=8, device="cuda", gt_views=0)
    np.save("/tmp/pmc_scene.npy", sc["gray"].cpu().numpy())
    json.dump({k: np.asarray(sc[k]).tolist() for k in ("K", "R", "C", "neighbors", "dmin", "dmax")}, open("/tmp/pmc_scene.json", "w"))
    print("scene written", V)
else:
    from openmvs_amd.patchmatch import PatchMatchHIP, default_params
    geo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    gray = np.load("/tmp/pmc_scene.npy", mmap_mode="r")
    m = json.load(open("/tmp/pmc_scene.json"))
    V, H, W = gray.shape
    e = PatchMatchHIP(0); e.Init(True); e.scene_create(V, W, H, 2)
    for i in range(V):
        e.scene_set_view(i, np.ascontiguousarray(gray[i]), np.array(m["K"][i]), np.array(m["R"][i]), np.array(m["C"][i]), float(m["dmin"][i]), float(m["dmax"][i]), np.array(m["neighbors"][i], np.int32))
    p = default_params(seed=1, nEstimationGeometricIters=geo)
    ids = list(range(V))
    e.stats_reset(True); t = time.time()
    e.scene_estimate(ids, -1, p)
    for g in range(geo):
        e.scene_commit_round(); e.scene_estimate(ids, g, p)
    e.sync(); dt = time.time() - t
    st = e.stats_get()
    print(json.dumps({"views": V, "geo_iters": geo, "seconds": round(dt, 3), "sweep_launches": int(st.sweepLaunches), "avg_launch_us": round(1e3 * st.sweepMs / max(1, st.sweepLaunches), 2),
                      "algorithmic_bytes_per_launch": round(st.sweepBytes / max(1, st.sweepLaunches), 1)}))
    e.close()

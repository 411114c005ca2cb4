"""Round-5 probe (tools/r04/probe_lanes.py carried over): full schedule (photometric pass + 2 geometric rounds) of a V-view 1920x1080 scene under several engine switches, one fresh engine each.
    python tools/r05/probe_groups.py V "NAME:K=V,K=V" ...
Prints seconds per step and Mpix/s per configuration, and whether view 0's depth map equals the first configuration's (probe builds: expected to differ)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from openmvs_amd import synth
from openmvs_amd.patchmatch import PatchMatchHIP, default_params

V = int(sys.argv[1]); W, H = 1920, 1080
configs = []
for a in sys.argv[2:]:
    name, _, kv = a.partition(":")
    configs.append((name, dict(x.split("=") for x in kv.split(",") if x)))
dev = torch.device("cuda", 0)
sc = synth.make_scene_torch(V, W, H, n_src=8, device=dev, gt_views=1)
gray = sc["gray"]; torch.cuda.synchronize()
p = default_params(seed=1, nEstimationGeometricIters=2)
ref = None
steps = int(os.environ.get("PROBE_STEPS", "2"))
for name, env in configs:
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = PatchMatchHIP(0); e.Init(True); e.scene_create(V, W, H, 2)
        same = os.environ.get("PROBE_SAME_NB")      # timing probe (results meaningless): every view's 8 sources are 8 copies of its first neighbour -> an eighth of the image footprint
        for i in range(V):
            nb = [sc["neighbors"][i][0]] * len(sc["neighbors"][i]) if same else sc["neighbors"][i]
            e.scene_set_view(i, None, sc["K"][i], sc["R"][i], sc["C"][i], float(sc["dmin"][i]), float(sc["dmax"][i]), nb)
        e.scene_copy(0, 0, V, gray.data_ptr(), True); e.sync()
        allv = list(range(V))
        best = 1e9
        for rep in range(1 + steps):
            for v in allv: e.scene_reset_view(v)
            e.sync(); t = time.perf_counter()
            e.scene_estimate(allv, -1, p, sync=False)
            for g in range(2):
                e.scene_commit_round(); e.scene_estimate(allv, g, p, sync=False)
            tq = time.perf_counter() - t                      # the host has enqueued everything (no sync inside the calls): host-bound if this is the whole step
            e.sync(); dt = time.perf_counter() - t
            if rep and dt < best: best = dt; enq = tq
        init_ms = None
        if os.environ.get("PROBE_STATS"):                      # one more step with the engine's event timing: milliseconds of the init kernels (ScoreDepthMapTmp) per step
            e.stats_reset(True)
            for v in allv: e.scene_reset_view(v)
            e.scene_estimate(allv, -1, p, sync=False)
            for g in range(2):
                e.scene_commit_round(); e.scene_estimate(allv, g, p, sync=False)
            e.sync(); st = e.stats_get(); init_ms = st.initMs
            print("   init kernels %.1f ms per step (%d launches), sweeps wall %.1f ms" % (st.initMs, st.initLaunches, st.sweepWallMs), flush=True)
        if os.environ.get("PROBE_SPLIT"):                      # one more step with a sync after every call: seconds of the photometric pass and of each geometric round
            for v in allv: e.scene_reset_view(v)
            e.sync(); parts = []
            t = time.perf_counter(); e.scene_estimate(allv, -1, p, sync=False); e.sync(); parts.append(time.perf_counter() - t)
            for g in range(2):
                e.scene_commit_round(); t = time.perf_counter(); e.scene_estimate(allv, g, p, sync=False); e.sync(); parts.append(time.perf_counter() - t)
            print("   photometric %.3f s, geometric rounds %.3f / %.3f s" % tuple(parts), flush=True)
        d = np.stack([e.scene_get_maps(0)[0], e.scene_get_maps(V - 1)[0]])      # the first and the last view of the batch
        if ref is None: ref = d
        print("%-28s %-60s %.3f s/step (host enqueue %.3f s)  %.2f Mpix/s  same-as-first %s" % (name, env, best, enq, V * W * H / best / 1e6, bool(np.array_equal(d, ref))), flush=True)
        e.close()
    finally:
        for k, v in saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

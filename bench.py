#!/usr/bin/env python
"""bench.py -- depth-map throughput of the MI355X-native PatchMatch engine.

Metric (BASELINE.json): Mpix/s of depth-map output at 1920x1080, N = 8 source views.
Workload (BASELINE config 3): a FIXED synthetic scene of `--views` (100) views of 1920x1080; at N GPUs every rank owns a contiguous block
of 100/N reference views ("scaling": "strong"; `--weak` restores round 1's 100 views per GPU).  One step = the full reference schedule for
every view of the rank's block: photometric pass (3-level pyramid x 3 sweeps, SceneDensify.cpp:616-805) + 2 geometric-consistency rounds,
with the depth maps of the previous round exchanged between ranks at the two round boundaries (one RCCL all-gather each; the only
collectives besides the initial image broadcast).  `--with-filter` appends the cross-view filter of config 5 to the step (all-gather of
depth + confidence, FilterDepthMap on the rank's own views, SceneDensify.cpp:2136-2222).  Inputs are resident in HBM before the clock starts.

    python bench.py --gpus N --steps K --warmup W        (N > 1: under torch.distributed.run)

Prints ONE JSON line on rank 0.  At N = 1 the line also carries (outside the timed region):
  config2  -- BASELINE config 2 through the one-call boundary (pmhip_estimate_depth_map, host buffers in / out): seconds per depth map;
  parity   -- the 9-view 1920x1080 scene against the committed digests of the sequential CPU oracle (tests/golden/), all 27 maps,
              plus a small live oracle-vs-engine comparison;
  sgm      -- BASELINE config 4: SemiGlobalMatcher::Match at 2048x1536, D = 64 and 128, ms per Match and GB/s on the 43 B/cost model;
  cpu_baseline -- the restated CPU path timed on this box's host cores.
The CPU oracle is used only for `cpu_baseline` and the live parity leg -- never inside a timed GPU region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TFLOPS = 157.3  # FP32 vector peak (packed), same guide
FLOP_PER_PIXEL = 0.33e6   # algorithmic FP32 work of the full schedule per output pixel at N = 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=100, help="views of the scene (strong scaling: shared by all GPUs)")
    ap.add_argument("--views-per-gpu", type=int, default=0, help="implies --weak with this many views per GPU")
    ap.add_argument("--weak", action="store_true", help="scene grows with the GPUs: --views per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sources", type=int, default=8)
    ap.add_argument("--geo-iters", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="reference views estimated concurrently (0 = whole block)")
    ap.add_argument("--with-filter", action="store_true", help="append the cross-view depth-map filter (config 5's exchange) to every step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the config2 / parity / sgm blocks")
    ap.add_argument("--images", choices=("auto", "broadcast", "needed"), default="auto", help="N > 1: one broadcast of the image set, or rank 0 sends every rank only the views it holds (block + the foreign views it reads); auto = needed when every rank holds less than half of the scene")
    ap.add_argument("--no-shard-rates", action="store_true", help="skip the shard-size legs (the blocks a rank owns at 2 / 4 / 8 GPUs, timed on this GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the baseline sample")
    ap.add_argument("--groups", type=int, default=0, help="PMHipTuning::viewGroups (0 = the engine's default)")
    ap.add_argument("--lanes", type=int, default=0, help="PMHipTuning::sweepLanes (0 = the engine's default)")
    ap.add_argument("--wide-pixels", type=int, default=0, help="PMHipTuning::widePixels (0 = the engine's default)")
    ap.add_argument("--wide-max-views", type=int, default=0, help="PMHipTuning::wideMaxViews (0 = the engine's default)")
    ap.add_argument("--wide-hyps", type=int, default=0, help="PMHipTuning::wideHyps (0 = the engine's default)")
    ap.add_argument("--wide8-pixels", type=int, default=0, help="PMHipTuning::wide8Pixels (0 = the engine's default)")
    ap.add_argument("--tiles", type=int, default=0, help="OPT-IN estimator mode, not the reference's sweep: tiled sweeps with tiles of this many pixels (pmhip_set_sweep_tiles); the line says so")
    ap.add_argument("--no-tiled-leg", action="store_true", help="skip the tiled-sweeps leg (the opt-in mode's rates at the full batch and at the shard sizes)")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (a.gpus, a.gpus))
    import torch.distributed as dist
    # OPENMVS_AMD_DIST_BACKEND=gloo (+ more ranks than GPUs): the N-rank path on a box with fewer GPUs -- ranks share devices, collectives go through the host.  A functional
    # check of the sharded schedule on real hardware (tools/r04/two_ranks_one_gpu.sh); the driver's scaling runs use the default, RCCL with one rank per GPU.
    backend = os.environ.get("OPENMVS_AMD_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # OPENMVS_AMD_FORCE_COLLECTIVES=1: run the collective plumbing (RCCL process group, broadcasts, all-gathers, barrier, all-reduce) with one rank too, so
    # that a 1-GPU box can dry-run the exact calls the 8-GPU launch makes (two ranks cannot share a device under RCCL)
    dist_on = world > 1 or os.environ.get("OPENMVS_AMD_FORCE_COLLECTIVES") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from openmvs_amd import synth
    from openmvs_amd.distributed import ShardedDensifier, shard_range
    from openmvs_amd.patchmatch import PatchMatchHIP, default_params

    W, H, N = a.width, a.height, a.sources
    weak = a.weak or a.views_per_gpu > 0
    V = (a.views_per_gpu or a.views) * world if weak else a.views
    mine = list(shard_range(V, world, rank))
    # ---- inputs: rank 0 renders the scene, one broadcast of the image set -------------------
    if rank == 0:
        sc = synth.make_scene_torch(V, W, H, n_src=N, device=dev, gt_views=1)
        gray = sc["gray"]
        meta = [sc["K"], sc["R"], sc["C"], sc["neighbors"], sc["dmin"], sc["dmax"], sc["diameter"]]
        gt0 = sc["gt_depth"][0].cpu().numpy()
    else:
        gray = None; meta = None; gt0 = None
    if dist_on:
        box = [meta]
        dist.broadcast_object_list(box, 0)
        meta = box[0]
    K, R, Cc, nbr, dmin, dmax, diameter = meta

    # ---- this rank's part of the scene: its block of reference views and the foreign views they read, in a compact scene of local slots (own block first) ----
    from openmvs_amd.distributed import needed_views
    nbr_lists = [[int(x) for x in nbr[i]] for i in range(V)]
    mine_chk, foreign = needed_views(nbr_lists, V, world, rank)
    assert mine_chk == mine
    held = mine + foreign                                   # global view ids in slot order
    slot = {g: i for i, g in enumerate(held)}
    # images: ONE broadcast of the whole set over xGMI (north_star; the default), every rank then keeps its block + closure; or, `--images needed`, rank 0 sends every rank
    # just the views it holds, point to point -- what a scene too large to pass through every GPU (or a closure much smaller than the scene, BASELINE config 5) wants
    images_how = "one broadcast of the image set"
    images = a.images
    if images == "auto":                                    # every rank derives the same answer from the neighbour lists
        most = max(len(set(needed_views(nbr_lists, V, world, r)[0]) | set(needed_views(nbr_lists, V, world, r)[1])) for r in range(world))
        images = "needed" if 2 * most < V else "broadcast"
    if dist_on and images == "needed" and world > 1:
        images_how = "point to point from rank 0: each rank receives the %d views it holds" % len(held)
        staged = backend != "nccl"                          # (gloo has no point-to-point for device tensors: through the host then)
        if rank == 0:
            reqs = []
            for r in range(1, world):
                m_r, f_r = needed_views(nbr_lists, V, world, r)
                if m_r + f_r:
                    buf = gray[m_r + f_r].contiguous()
                    reqs.append((dist.isend(buf.cpu() if staged else buf, r), buf))
            for w_, _ in reqs:
                w_.wait()
            local_gray = gray[held].contiguous()
        else:
            local_gray = torch.empty((len(held), H, W), dtype=torch.float32, device="cpu" if staged else dev)
            if held:
                dist.recv(local_gray, 0)
            local_gray = local_gray.to(dev)
    else:
        if rank != 0:
            gray = torch.empty((V, H, W), dtype=torch.float32, device=dev)
        if dist_on:
            dist.broadcast(gray, 0)                         # the single broadcast of the image set over xGMI
        local_gray = gray[held].contiguous() if (foreign or not mine or len(mine) < V) else gray
    torch.cuda.synchronize()
    del gray
    eng = PatchMatchHIP(local)
    eng.Init(True)
    if a.groups or a.lanes or a.wide_pixels or a.wide_max_views or a.wide_hyps or a.wide8_pixels:
        eng.tuning(viewGroups=a.groups, sweepLanes=a.lanes, wideMaxViews=a.wide_max_views, widePixels=a.wide_pixels, wideHyps=a.wide_hyps, wide8Pixels=a.wide8_pixels)
    if a.tiles:
        eng.set_sweep_tiles(a.tiles, a.tiles)                # OPT-IN estimator mode (the line's config says so): not the reference's sweep order
    eng.scene_create(max(2, len(held)), W, H, 2)
    for i, g in enumerate(held):
        eng.scene_set_view(i, None, K[g], R[g], Cc[g], float(dmin[g]), float(dmax[g]), [slot[n] for n in nbr_lists[g]] if i < len(mine) else [])
        eng.scene_set_view_id(i, g)                         # random numbers by the view's index in the whole scene: the maps do not depend on the split
    if held:
        eng.scene_copy(0, 0, len(held), local_gray.data_ptr(), True)
    eng.sync()
    del local_gray
    torch.cuda.empty_cache()
    p = default_params(seed=1, nEstimationGeometricIters=a.geo_iters)
    B = a.batch if a.batch > 0 else max(1, len(mine))

    class EngineEstimator:
        """Adapter between the sharding driver (openmvs_amd/distributed.py, also exercised under gloo in tests/test_distributed.py) and the HBM-resident scene
        interface of the HIP engine.  View ids are the scene's; the engine sees local slots."""

        def __init__(self):
            self.buf = torch.empty((max(1, len(mine)), H, W), dtype=torch.float32, device=dev)

        def reset(self, ids):
            for v in ids:
                eng.scene_reset_view(slot[v])

        def estimate(self, ids, geo):
            s = [slot[v] for v in ids]
            for i in range(0, len(s), B):
                eng.scene_estimate(s[i:i + B], geo, p, sync=False)

        def local_maps(self, ids, what):
            buf = self.buf[:len(ids)]
            if len(ids):
                assert list(ids) == mine
                eng.scene_copy({"depth": 1, "conf": 3}[what], 0, len(ids), buf.data_ptr(), False)
            eng.sync()
            return buf

        def local_depths(self, ids):
            return self.local_maps(ids, "depth")

        def set_snapshot_views(self, own_ids, own, foreign_ids, foreign_maps):
            # previous-round depth maps become visible to the next round (the reference writes depthNNNN.dmap and re-reads the neighbours' files,
            # SceneDensify.cpp:378-393,1943-1950): this rank's own maps by a device copy, the foreign ones from the exchange
            eng.scene_commit_round()
            torch.cuda.synchronize()
            for k, g in enumerate(foreign_ids):
                eng.scene_copy(4, slot[g], 1, foreign_maps[k].data_ptr(), True)
            eng.sync()                                   # the received tensor may be released by the caller

        def set_maps_views(self, what, foreign_ids, foreign_maps):
            torch.cuda.synchronize()
            for k, g in enumerate(foreign_ids):
                eng.scene_copy({"depth": 1, "conf": 3}[what], slot[g], 1, foreign_maps[k].data_ptr(), True)
            eng.sync()

        def filter(self, ids):
            if len(ids):
                eng.scene_filter([slot[v] for v in ids], True, 2, 1, 0.01, commit=True)

    drv = ShardedDensifier(EngineEstimator(), V, world, rank, geo_iters=a.geo_iters, neighbors=nbr_lists)
    assert drv.mine == mine

    def step():
        drv.run()
        if a.with_filter:
            drv.filter()

    def fence():
        eng.sync(); torch.cuda.synchronize()
        if dist_on:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    fence()
    eng.stats_reset(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    own_dt = dt
    st = eng.stats_get()
    eng.stats_reset(False)
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    mpix = V * W * H * a.steps / dt / 1e6
    digests = None
    if os.environ.get("OPENMVS_AMD_BENCH_DIGESTS") == "1":   # sha-1 of every view's final depth map, gathered: equal between a 1-rank and an N-rank run of the same scene
        import hashlib
        own = {int(g): hashlib.sha1(eng.scene_get_maps(slot[g])[0].tobytes()).hexdigest()[:12] for g in mine}
        if dist_on:
            box = [None] * world
            dist.all_gather_object(box, own)
            own = {k: v for d in box for k, v in d.items()}
        digests = {str(k): own[k] for k in sorted(own)}
    mine_info = {"rank": rank, "views_per_gpu": len(mine), "foreign_views_held": len(foreign), "kernel": sweep_kernel_name(len(mine) if not a.batch else min(a.batch, len(mine)), N),
                 # exchange = sending / receiving / installing maps at the round boundaries; wait = blocked on this rank's own asynchronous estimate before its maps can be read
                 "exchange_ms_per_step": round(1e3 * drv.exchange_seconds / max(1, a.steps + a.warmup), 2),
                 "wait_for_own_estimate_ms_per_step": round(1e3 * drv.wait_seconds / max(1, a.steps + a.warmup), 2), "seconds": round(own_dt, 3),
                 "scene_mb": round(eng.scene_bytes() / 1e6, 1), "scene_mb_per_view_held": round(eng.scene_bytes() / 1e6 / max(1, len(held)), 1)}
    ranks_info = [mine_info]
    if dist_on:
        box = [None] * world
        dist.all_gather_object(box, mine_info)
        ranks_info = box

    out = None
    if rank == 0:
        d0, n0, c0 = eng.scene_get_maps(0)
        m = d0 > 0
        rel = np.abs(d0[m] - gt0[m]) / gt0[m]
        sweep_s = st.sweepMs / 1e3          # summed over the streams the launches ran on (== sum of kernel durations)
        wall_s = st.sweepWallMs / 1e3       # wall time of the passes: hand-offs, init passes, sweeps, finalize of all view groups (which overlap)
        achieved = st.sweepBytes / 1e9 / max(sweep_s, 1e-12)
        device = st.sweepBytes / 1e9 / max(wall_s, 1e-12)
        per_launch = st.sweepBytes / max(1, st.sweepLaunches)
        valu_tf = FLOP_PER_PIXEL * (len(mine) * W * H * a.steps / dt) / 1e12   # this rank's pixels per second x algorithmic flop per pixel
        kern = sweep_kernel_name(B if B else len(mine), N)
        same_cfg = world == 1 and V == 100 and (W, H, N) == (1920, 1080, 8) and B == V and a.geo_iters == 2    # the configuration the counters were taken on
        issue = valu_issue_fields(wall_s, a.steps, st.sweepPixels) if same_cfg else {}
        if same_cfg:
            issue.update(gather_issue_fields(wall_s, a.steps))
            if "gather_issue" in issue:     # (the contract's `bound` stays "hbm"; what the kernels actually run into is named beside it)
                issue["bound_measured"] = "texture-address units in front of the vector L1 (scattered wave-loads): gather_issue.frac %.2f; VALU issue beside it: valu_issue.valu_busy_frac %.2f; HBM is not the bound (frac %.4f of its peak on algorithmic bytes)" % (
                    issue["gather_issue"]["frac"], issue.get("valu_issue", {}).get("valu_busy_frac", float("nan")), achieved / HBM_PEAK_GBS)
        tf = traffic_fields(per_launch) if same_cfg else {"traffic": None, "traffic_note": "counters exist for the 100-view 1920x1080 one-GPU configuration only"}
        if "fabric_bytes_per_step" in tf.get("traffic_measurement", {}):   # the fabric-side rate of THIS run: counter bytes of a step / this run's wall time of the passes per step
            tf["traffic_measurement"]["fabric_rate_gbs_this_run"] = round(tf["traffic_measurement"]["fabric_bytes_per_step"] * a.steps / 1e9 / max(wall_s, 1e-12), 1)
            tf["traffic_measurement"]["fabric_rate_frac_of_hbm_peak"] = round(tf["traffic_measurement"]["fabric_rate_gbs_this_run"] / HBM_PEAK_GBS, 4)   # (Infinity-Cache hits are in the counter: an upper bound on HBM traffic)
        out = {
            "metric": "Mpix/s depth-map output at 1920x1080 N-view", "value": round(mpix, 3), "unit": "Mpix/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 2),
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d-view %dx%d synthetic scene, %d source views per reference view, PatchMatch photometric pass "
                                   "(3-level pyramid x 3 sweeps) + %d geometric rounds%s, all depth maps"
                                   % (V, W, H, N, a.geo_iters, " + cross-view filter" if a.with_filter else ""),
                       **({"estimator_mode": "OPT-IN tiled sweeps, %dx%d tiles (pmhip_set_sweep_tiles): NOT the reference's sweep order; see tiled_sweeps / tolerance" % (a.tiles, a.tiles)} if a.tiles else {}),
                       "views_total": V, "views_per_gpu": len(mine), "batch": B, "parallelism": "reference views sharded over %d GPU(s)" % world,
                       "exchange": "neighbour-only point-to-point (a rank holds its block and the %d foreign views it reads)" % len(foreign), "images": images_how, "ranks": ranks_info,
                       **({"backend": backend} if dist_on else {}), **({"depth_digests": digests} if digests else {})},
            "roofline": {"bound": "hbm", "kernel": kern, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), **tf, **issue,
                         "launches": int(st.sweepLaunches), "avg_launch_us": round(1e3 * st.sweepMs / max(1, st.sweepLaunches), 2),
                         "algorithmic_bytes_per_launch": round(per_launch, 1),
                         "concurrent_streams": round(sweep_s / max(wall_s, 1e-12), 2), "device_achieved": round(device, 2),
                         "device_frac": round(device / HBM_PEAK_GBS, 6), "passes_share_of_step": round(wall_s / dt, 4),
                         "valu_achieved_tflops": round(valu_tf, 3), "valu_peak_tflops": VALU_PEAK_TFLOPS, "valu_frac": round(valu_tf / VALU_PEAK_TFLOPS, 5),
                         "note": "achieved = algorithmic bytes per launch (SURVEY 8(d) B_sweep / launches) / average launch duration from HIP events on the "
                                 "launching streams (rank 0); the view groups run on separate streams, so the device moves device_achieved; "
                                 "valu_* = SURVEY 8(d)'s 0.33 MFLOP per output pixel x this GPU's pixel rate against the FP32 vector peak"},
            "accuracy": {"valid_frac_view0": round(float(m.mean()), 4), "median_rel_err_vs_ground_truth": float(np.median(rel))},
        }
    # ---- extra legs (rank 0, 1 GPU only; outside the timed region) ---------------------------
    if rank == 0 and world == 1 and not weak and not a.no_shard_rates and V >= 16:
        out.update(shard_rate_legs(eng, a, V, W, H, p, mpix))
    if rank == 0 and world == 1 and not weak and not a.no_tiled_leg and not a.tiles and V >= 16:
        out.update(tiled_legs(eng, a, V, W, H, p, mpix))
    if rank == 0 and world == 1:
        eng.scene_create(2, 16, 16, 0)   # release the benchmark scene's HBM before the other legs
        if not a.no_extras:
            out.update(golden_and_config2(eng))
            out["sgm"] = sgm_leg(local)
        if not a.no_cpu_baseline:
            out.update(cpu_legs(a, eng))
    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if dist_on:
        dist.destroy_process_group()


def shard_rate_legs(eng, a, V, W, H, p, rate_full):
    """The one-GPU side of the scaling model, under the driver's clock: the block of reference views ONE rank owns when the same scene is split over 2 / 4 / 8 GPUs
    (ceil(V / N) contiguous views; their foreign source views are resident here as they are on that rank), full schedule, 1 warm-up + 2 timed steps each.
    scaling_model[N] = N x rate(ceil(V / N)) / rate(V): what N GPUs would deliver if the exchange at the round boundaries were free (it is two neighbour-only
    point-to-point rounds per step; its cost is `exchange_ms_per_step` of an N-rank run)."""
    rates, model = {}, {}
    for n_gpus in (2, 4, 8):
        n = -(-V // n_gpus)
        lo = (V - n) // 2                                  # a block from the middle of the scene: neighbours on both sides, like most ranks' blocks
        ids = list(range(lo, lo + n))
        best = None
        for rep in range(3):
            for v in ids:
                eng.scene_reset_view(v)
            eng.sync(); t = time.perf_counter()
            eng.scene_estimate(ids, -1, p, sync=False)
            for g in range(a.geo_iters):
                eng.scene_commit_round(); eng.scene_estimate(ids, g, p, sync=False)
            eng.sync(); dt = time.perf_counter() - t
            if rep and (best is None or dt < best):
                best = dt
        rates[str(n)] = round(n * W * H / best / 1e6, 3)
        model[str(n_gpus)] = round(n_gpus * rates[str(n)] / rate_full, 3)
    return {"shard_rates": {"unit": "Mpix/s on one GPU for a block of this many reference views of the same scene (best of 2 timed steps)", **rates},
            "scaling_model": {"note": "N x rate(ceil(views / N)) / rate(views), exchange excluded; measured on ONE GPU, not a multi-GPU run", **model}}


def tiled_legs(eng, a, V, W, H, p, rate_full, tile=64):
    """The OPT-IN tiled sweeps (pmhip_set_sweep_tiles: sweeps run inside tile x tile tiles, a neighbour across a tile border is read as the previous sweep left it -- another
    estimator than the reference's sequential sweep, deterministic, bit-identical to its own oracle restatement; how far its maps are from the reference's: `tolerance`): the
    same workload and the shard-size blocks of `shard_rates`, 1 warm-up + 2 timed steps each.  Never part of `value`."""
    eng.set_sweep_tiles(tile, tile)
    rates, model = {}, {}
    try:
        for n_gpus in (1, 2, 4, 8):
            n = -(-V // n_gpus)
            lo = (V - n) // 2
            ids = list(range(lo, lo + n))
            best = None
            for rep in range(3):
                for v in ids:
                    eng.scene_reset_view(v)
                eng.sync(); t = time.perf_counter()
                eng.scene_estimate(ids, -1, p, sync=False)
                for g in range(a.geo_iters):
                    eng.scene_commit_round(); eng.scene_estimate(ids, g, p, sync=False)
                eng.sync(); dt = time.perf_counter() - t
                if rep and (best is None or dt < best):
                    best = dt
            rates[str(n)] = round(n * W * H / best / 1e6, 3)
            if n_gpus > 1:
                model[str(n_gpus)] = round(n_gpus * rates[str(n)] / rates[str(V)], 3)
    finally:
        eng.set_sweep_tiles(0, 0)
    return {"tiled_sweeps": {"what": "OPT-IN, not the reference's sweep order: %dx%d tiles, neighbours across tile borders from the previous sweep; default off; `value` never uses it" % (tile, tile),
                             "rates": {"unit": "Mpix/s on one GPU for a block of this many reference views, full schedule (best of 2 timed steps)", **rates},
                             "vs_exact_sweep_at_%d_views" % V: round(rates[str(V)] / rate_full, 3),
                             "scaling_model": {"note": "N x rate(ceil(views / N)) / rate(views) of the tiled mode, exchange excluded; measured on ONE GPU", **model}}}


def sweep_kernel_name(n_batch, n_src):
    """The sweep kernel the engine picks for a batch of n_batch reference views (pm_engine.hip: PMHIP_DEFAULT_WIDE, PMHIP_DEFAULT_WIDE_PIXELS, PMHIP_LANES4_FROM).  Larger batches
    sweep their long diagonals (launches above PMHIP_DEFAULT_WIDE_PIXELS pixels: most of the time) with pm_sweep2_kernel and the short ones with the two-wide speculative kernel."""
    if n_batch <= 32 and n_src <= 8:
        return "pm_sweep_wide_kernel" if n_batch <= 2 else "pm_sweep_widen_kernel<2>"
    return "pm_sweep2_kernel"


def _counters():
    """profiles/traffic.json (tools/r05/make_traffic.py): counters taken on the benchmark's own workload -- 100 views, two view groups, full schedule -- through the stand-alone
    C++ program, because rocprofv3 --pmc cannot run next to this process's timing (separate passes, no trace domains; MI355X_MICROARCH.md).  None when the file is absent or
    was measured on other kernel sources than this tree's."""
    try:
        import hashlib
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        h = hashlib.sha256()
        for f in ("pm_kernels.hip", "pm_band.hip", "pm_wide_n.hip", "pm_math.h"):
            h.update(open(os.path.join(ROOT, "openmvs_amd", "csrc", f), "rb").read())
        return t if h.hexdigest()[:16] == t.get("kernel_digest") and "sweeps" in t else None
    except Exception:
        return None


def _counters_round4():
    """The committed counter file of round 4 (24-view, one-stream C++ workload, round 4's kernels), when profiles/traffic.json still is that one: reported as what it is."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return t if "bytes_per_launch_fetch_doubled" in t and "sweeps" not in t else None
    except Exception:
        return None


def traffic_fields(algorithmic_bytes_per_launch):
    """`traffic`: fabric-side bytes (FETCH_SIZE x 2 + WRITE_SIZE) per sweep launch, measured OFFLINE on this benchmark's own configuration (same scene size, batch, groups,
    kernels: the launches are the same ones); null without counters of this tree's kernels."""
    t = _counters()
    if t is None:
        o = _counters_round4()
        if o is None:
            return {"traffic": None, "traffic_note": "no counter passes of this tree's kernels (profiles/traffic.json absent or measured on other sources)"}
        ratio = o["bytes_per_launch_fetch_doubled"] / max(1.0, o["algorithmic_bytes_per_launch"])
        return {"traffic": round(ratio * algorithmic_bytes_per_launch), "traffic_unit": "bytes per sweep launch (FETCH_SIZE x 2 + WRITE_SIZE per algorithmic byte of the counter passes x this run's algorithmic bytes per launch)",
                "traffic_measurement": {"over_algorithmic": round(ratio, 2), "measured": "offline, ROUND 4: 24-view one-stream C++ workload on round 4's pm_sweep2_kernel<4,2> (kernel digest %s); this tree's sweep "
                                        "kernels differ from those in the guarded redo path and the reference-patch indexing only.  The round-5 attempt to take the counters on the benchmark's own "
                                        "100-view workload did not finish: every rocprofv3 --pmc pass ran into its 400 s limit (43 126 dispatches at ~9 ms each under counter collection; "
                                        "profiles/r05_call5_pmc_fetch_timeout.err)" % o.get("kernel_digest"), "source": o.get("source")}}
    sw = t["sweeps"]
    return {"traffic": sw["fabric_bytes_per_launch"], "traffic_unit": "bytes per sweep launch (FETCH_SIZE x 2 + WRITE_SIZE; offline counter passes on this configuration)",
            "traffic_measurement": {"over_algorithmic": sw["over_algorithmic"], "fetch_bytes_per_launch_raw": sw["fetch_bytes_per_launch_raw"], "write_bytes_per_launch": sw["write_bytes_per_launch"],
                                    "fabric_bytes_per_step": sw["fabric_bytes_per_step"], "correction": sw["correction"], "dispatches": sw["dispatches"],
                                    "per_kernel": {k: {q: v[q] for q in ("dispatches", "fabric_bytes_per_launch") if q in v} for k, v in t.get("families", {}).items()},
                                    "measured": "offline", "kernel_digest": t["kernel_digest"], "source": t["source"], **({"notes": t["notes"]} if "notes" in t else {})}}


def valu_issue_fields(sweep_wall_s, steps, sweep_pixels=0, simds=1024, clock_hz=2.4e9):
    """What the sweep kernels actually run into: the share of the SIMDs' cycles during THIS run's passes in which a VALU instruction of a sweep kernel executes
    = SQ_ACTIVE_INST_VALU summed over every sweep launch of one step (offline counters on this configuration, all kernel families) / (1024 SIMDs x 2.4 GHz x this run's wall
    time of the passes per step).  Empty without counters of this tree's kernels."""
    t = _counters()
    if t is None or "valu" not in t:
        o = _counters_round4()
        sq = (t or {}).get("sq_round4") or (o or {}).get("sq", {})
        if "frac_active_valu" not in sq or not sweep_pixels:
            return {}
        busy = 4.0 * sq["frac_active_valu"] * sq["wave_quadcycles_per_wave"]          # cycles per wave-visit of pm_sweep2_kernel<4,2>
        visits = sweep_pixels / 16.0
        return {"valu_issue": {"valu_busy_frac": round(busy * visits / (simds * clock_hz * max(sweep_wall_s, 1e-12)), 4), "valu_insts_per_wave_visit": sq["valu_insts_per_wave"],
                               "note": "EXTRAPOLATION from round 4's counters (24-view one-stream workload, round 4's pm_sweep2_kernel<4,2>): VALU-busy cycles of one wave-visit x this run's "
                                       "pixel visits / 16 pixels per wave / (1024 SIMDs x 2.4 GHz x wall time of the passes); every visit is priced as <4,2>'s although the short launches run the "
                                       "two-wide kernel.  Round 5 took FETCH_SIZE on the benchmark's own workload (roofline.traffic); its SQ passes hung "
                                       "(rocprofv3 --pmc at start-up, profiles/r05_call6_pmc/run.log)"}}
    v = t["valu"]
    return {"valu_issue": {"valu_active_cycles_per_step": v["valu_active_cycles_per_step"], "wave_visits_per_step": v["wave_visits_per_step"], "valu_insts_per_wave_visit": v["valu_insts_per_wave_visit"],
                           "valu_busy_frac": round(v["valu_active_cycles_per_step"] * steps / (simds * clock_hz * max(sweep_wall_s, 1e-12)), 4),
                           "per_kernel": {k: {q: f[q] for q in ("frac_active_valu", "frac_wait_inst_any", "valu_insts_per_wave", "waves_per_launch") if q in f} for k, f in t.get("families", {}).items()},
                           "note": "share of all SIMD cycles of the passes in which a VALU instruction of a sweep kernel executes (SQ counters of every sweep launch of one step of this "
                                   "configuration, offline; wall time of this run; 2.4 GHz): the bound the kernels approach is the issue of the reference's un-fusable fp32 arithmetic, not HBM"}}


def gather_issue_fields(sweep_wall_s, steps, cus=256, clock_hz=2.4e9):
    """The unit the sweeps run into: a CU's texture-address unit takes one scattered wave-load per 64.5 cycles whatever its width (tools/probes/l1_gather.hip), and a
    tap of the estimator is one such wave-load.  achieved = vector-memory wave-loads of one step (SQ_INSTS_VMEM_RD of every sweep launch, offline counters on this configuration)
    per second of THIS run's passes; peak = 256 units x 2.4 GHz / 64.5.  Empty without counters of this tree's kernels."""
    t = _counters()
    if t is None or "gather" not in t or not t["gather"].get("vmem_rd_wave_loads_per_step"):
        return {}
    g = t["gather"]
    peak = cus * clock_hz / g["cycles_per_scattered_wave_load"]
    ach = g["vmem_rd_wave_loads_per_step"] * steps / max(sweep_wall_s, 1e-12)
    return {"gather_issue": {"bound": "vector-L1 texture-address units: scattered wave-loads", "achieved": round(ach / 1e9, 3), "peak": round(peak / 1e9, 3), "unit": "G wave-loads/s",
                             "frac": round(ach / peak, 4), "wave_loads_per_wave_visit": g["vmem_rd_wave_loads_per_wave_visit"], "ta_busy_frac_counters": g.get("ta_busy_frac_of_the_slice"),
                             "note": "every vector-memory read of the sweep kernels priced as a scattered wave-load (the tap rows' 16-byte gathers are ~93 % of them; the rest "
                                     "coalesce and cost less, so frac is an upper estimate of the unit's load); ta_busy_frac_counters = TA_TA_BUSY over the slice's cycles, all 256 units"}}


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU boxes expose 256 logical CPUs but a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(round(int(q) / int(per)))))
    except Exception:
        pass
    return max(1, n)


def golden_and_config2(eng):
    """BASELINE config 2 (1 reference x 8 sources, 1920x1080): (1) parity of all 27 maps of the 9-view scene with the committed digests of the
    sequential oracle; (2) the reference view through the one-call boundary, timed per pass, checked against the same digests."""
    from openmvs_amd import synth
    from openmvs_amd.patchmatch import default_params
    from tests import golden_check as gc
    g = gc.load("pm_config2_1920x1080.json")
    c = g["case"]
    sc = synth.make_scene(c["n_views"], c["width"], c["height"], n_src=c["n_src"], device="cuda", gray_only=True, exact=True)
    same_inputs = gc.sha(sc.gray) == g["inputs"]["gray"] and all(gc.sha(getattr(sc, k)) == g["inputs"][k] for k in ("K", "R", "C"))
    eng.Init(True)
    eng.scene_load(sc, 2)
    p = default_params(seed=c["seed"], nEstimationGeometricIters=c["geo_iters"])
    allv = list(range(c["n_views"]))
    rounds, mismatches = [], []
    for r in range(1 + c["geo_iters"]):
        if r:
            eng.scene_commit_round()
        eng.scene_estimate(allv, r - 1, p)
        rounds.append([eng.scene_get_maps(v) for v in allv])
        for v in allv:
            try:
                gc.check_maps(rounds[r][v], g["rounds"][r][str(v)], "round %d view %d" % (r, v))
            except AssertionError as ex:
                mismatches.append(str(ex)[:300])
    # the same 27 maps through what the TIMED region runs for a 100-view batch, which the engine would not pick for 9 views by itself (set through pmhip_set_tuning):
    #  (1) pm_sweep2_kernel<4 lanes per pixel, 2 views per lane> alone; (2) the timed MIX: that kernel on the long diagonals and pm_sweep_widen_kernel<2> on the short ones
    #  inside one sweep, in two view groups -- the per-launch threshold scaled so that the same diagonals (by length) switch kernels as at 50 views per group:
    #  PMHIP_DEFAULT_WIDE_PIXELS 20000 / 50 = 400 pixels of diagonal, x 4.5 views per group here
    timed_mismatches, mix_mismatches = [], []
    from openmvs_amd.patchmatch import PatchMatchHIP
    for name, tun, sink in (("timed kernel", dict(wideMaxViews=-1, sweepLanes=4), timed_mismatches),
                            ("timed mix", dict(wideMaxViews=-1, sweepLanes=4, widePixels=1800, viewGroups=2), mix_mismatches)):
        e2 = PatchMatchHIP(0); e2.tuning(**tun); e2.Init(True); e2.scene_load(sc, 2)
        for r in range(1 + c["geo_iters"]):
            if r:
                e2.scene_commit_round()
            e2.scene_estimate(allv, r - 1, p)
            for v in allv:
                try:
                    gc.check_maps(e2.scene_get_maps(v), g["rounds"][r][str(v)], "%s, round %d view %d" % (name, r, v))
                except AssertionError as ex:
                    sink.append(str(ex)[:300])
        mix_tuning = e2.tuning()
        e2.close()
    timed_mismatches = timed_mismatches + mix_mismatches
    ref = c["ref"]
    ids = [ref] + list(sc.neighbors[ref])
    times, cur = [], None
    for rep in range(2):                 # first repetition warms the one-call path (allocations), second is reported
        times = []
        for r in range(1 + c["geo_iters"]):
            t = time.perf_counter()
            if r == 0:
                cur = eng.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[ref], sc.dmax[ref], params=p)
            else:
                cur = eng.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[ref], sc.dmax[ref], depth=cur[0], normal=cur[1],
                                           src_depths={v: rounds[r - 1][v][0] for v in ids[1:]}, nGeometricIter=r - 1, params=p)
            times.append(time.perf_counter() - t)
            if rep == 1:
                try:
                    gc.check_maps(cur, g["rounds"][r][str(ref)], "one-call boundary, round %d" % r)
                except AssertionError as ex:
                    mismatches.append(str(ex)[:300])
    px = c["width"] * c["height"]
    smp = np.asarray(g["rounds"][-1][str(ref)]["depth_sample"], np.float32)
    st = g["rounds"][-1][str(ref)]["depth_sample_step"]
    mine = rounds[-1][ref][0][::st, ::st].ravel()
    ok = (smp > 0) & (mine > 0)
    rmse = float(np.sqrt(np.mean((smp[ok].astype(np.float64) - mine[ok]) ** 2))) if ok.any() else float("nan")
    return {"config2": {"workload": "1 reference x 8 sources, 1920x1080, one depth map through pmhip_estimate_depth_map (host buffers in and out, PCIe included): "
                                    "photometric pass + %d geometric rounds" % c["geo_iters"],
                        "seconds_per_pass": [round(t, 4) for t in times], "seconds_per_depth_map": round(sum(times), 4),
                        "mpix_per_s": round(px / sum(times) / 1e6, 3)},
            "parity": {"case": "9-view 1920x1080 exact synthetic scene, every view 1 x 8, photometric + %d geometric rounds: all %d maps (depth, normal, confidence) "
                               "against the SHA-256 digests of the sequential CPU oracle (tests/golden/pm_config2_1920x1080.json), scene interface and one-call boundary"
                               % (c["geo_iters"], 3 * len(allv) * (1 + c["geo_iters"])),
                       "inputs_reproduced": bool(same_inputs), "bit_identical": bool(same_inputs and not mismatches and not timed_mismatches), "mismatches": (mismatches + timed_mismatches)[:4],
                       "kernel": "pm_sweep2_kernel", "kernel_note": "all 27 maps four ways: the engine's own choice for 9 views (pm_sweep_widen_kernel<2>); pm_sweep2_kernel<4,2> alone; the TIMED MIX "
                                 "(pm_sweep2_kernel<4,2> on the long diagonals, pm_sweep_widen_kernel<2> on the short ones of the same sweep, two view groups: tuning %s); "
                                 "the one-call boundary (pm_sweep_wide_kernel)" % json.dumps(mix_tuning),
                       "bit_identical_timed_kernel": bool(same_inputs and not timed_mismatches), "bit_identical_timed_mix": bool(same_inputs and not mix_mismatches),
                       "depth_rmse_over_diameter": rmse / sc.diameter, "tolerance": 1e-4,
                       "rmse_note": "over the golden file's strided depth sample of the reference view (exactly 0 when bit_identical)"}}


def sgm_leg(device):
    """BASELINE config 4: SemiGlobalMatcher::Match (cost volume + 8-path aggregation + WTA, SemiGlobalMatcher.cpp:863-1302) at 2048x1536.
    One reference against 4 sources in both directions = 8 Match calls per disparity range; inputs resident, HIP-event phase times from the engine."""
    from openmvs_amd import sgm
    from tests import sgm_cases as scs
    w, h = 2048, 1536
    lb, lg, rg = scs.stereo_pair(w, h, 21, seed=9)
    m = sgm.SemiGlobalMatcherHIP(device)
    res = {}
    for D in (64, 128):
        px, n, mx = scs.ranges(w, h, "uniform", 0, D)
        m.set_problem(lb, lg, rg, px, n, mx)
        m.Match()
        m.stats_reset(True)
        reps = 8
        t = time.perf_counter()
        for _ in range(reps):
            m.Match(sync=False)
        m.sync()
        dt = (time.perf_counter() - t) / reps
        s = m.stats_get()
        gb = 43.0 * n / 1e9
        res["D%d" % D] = {"ms_per_match": round(dt * 1e3, 3), "cost_ms": round(s.costMs / reps, 3), "aggregation_ms": round(s.aggrMs / reps, 3),
                          "wta_ms": round(s.wtaMs / reps, 3), "num_costs": int(n), "achieved_gbs": round(gb / dt, 1), "frac_of_hbm_peak": round(gb / dt / HBM_PEAK_GBS, 4),
                          "aggregation_gbs": round(40.0 * n / 1e9 / (s.aggrMs / reps / 1e3), 1),
                          # what the shipped kernels move per cost entry for uniform ranges (atomic-free aggregation): 1 cost write; 8 x (1 cost read + 1 delta
                          # write) in the path kernel; 8 delta + 1 cost read and 2 sum write in the sum / winner pass = 28 B
                          "moved_gbs": round(28.0 * n / 1e9 / dt, 1)}
    m.close() if hasattr(m, "close") else None
    return {"workload": "SemiGlobalMatcher::Match, 2048x1536, 8 calls (1 reference x 4 sources, both directions) per range; 43 B per cost entry "
                        "(1 cost write + 8 x (1 + 2 + 2) path traffic + 2 WTA read, SURVEY 8(d)): achieved_gbs / aggregation_gbs are the reference algorithm's bytes over "
                        "our time; moved_gbs is what the atomic-free kernels actually move (28 B per entry); wta_ms includes forming the u16 sums", "peak_gbs": HBM_PEAK_GBS, **res}


def native_oracle():
    """A second build of the oracle for TIMING only: -O3 -march=native, still -ffp-contract=off so that it computes
    the same values; built on the box it runs on (the tuned binary must not travel), into the system temp directory."""
    import ctypes as C
    import subprocess
    import tempfile
    from oracle import pyoracle as po
    so = os.path.join(tempfile.gettempdir(), "libpm_oracle_native_%d.so" % os.getuid())
    src = [os.path.join(ROOT, "oracle", f) for f in ("pm_oracle.cpp", "sgm_oracle.cpp", "filter_oracle.cpp", "fuse_oracle.cpp", "sgm_post_oracle.cpp")]
    try:
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
            subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-shared", "-o", so] + src,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = C.CDLL(so)
        lib.orc_estimate_depth_map.restype = C.c_int
        return lib, "-O3 -march=native -ffp-contract=off"
    except Exception:
        return po.lib(), "-O2 (native build failed)"


def cpu_legs(a, eng):
    """cpu_baseline: the REFERENCE'S OWN CODE on this box's host cores -- oracle/_ref/libref_driver_libm.so = DepthMapsData::EstimateDepthMap, ScaleDepthData, the pass
    bodies and the DepthEstimator cut verbatim from /root/reference (oracle/ref/build_ref.py, g++ -O3 -march=x86-64-v3, libm, std::mt19937), with the reference's own
    threading (scene.nMaxThreads = the usable cores: one estimator per thread on the shared pixel counter, SceneDensify.cpp:631-750) -- on >= 4 reference views x 8
    sources at the benchmark's own 1920x1080, full schedule (3-level pyramid + 2 geometric rounds).  OpenCV's resize is the oracle's restatement (un-vendored
    dependency).  `port` beside it: the restated oracle (oracle/pm_oracle.cpp) with the same threading on a smaller sample.  parity_live: sequential oracle vs the
    engine on a small case, bit for bit.  tolerance: what north_star's 1e-4 x diameter means against a reference BINARY (libm, mt19937, racy threads)."""
    from openmvs_amd import synth
    from openmvs_amd.patchmatch import default_params
    from oracle import pyoracle as po
    cores = usable_cores()
    seed = 1

    def engine_rounds(sc, tiles=0):
        eng.Init(True)
        eng.scene_load(sc, 2)
        eng.set_sweep_tiles(tiles, tiles)
        p = default_params(seed=seed, nEstimationGeometricIters=a.geo_iters)
        allv = list(range(sc.n_views))
        rounds = []
        eng.scene_estimate(allv, -1, p); rounds.append([eng.scene_get_maps(v) for v in allv])
        for g in range(a.geo_iters):
            eng.scene_commit_round(); eng.scene_estimate(allv, g, p); rounds.append([eng.scene_get_maps(v) for v in allv])
        eng.set_sweep_tiles(0, 0)
        return rounds

    def cpu_schedule(sc, rounds, refs, threads, estimate, chain):
        """photometric pass + geometric rounds of every view of `refs` through `estimate`; chain: feed a round with the CPU's own previous maps (parity) instead of the
        engine's (timing: decouples the views).  Returns ({ref: final maps}, seconds)."""
        t_cpu = 0.0; outs = {}
        allv = list(range(sc.n_views))
        for ref in refs:
            ids = [ref] + list(sc.neighbors[ref])
            cur = None
            for r in range(1 + a.geo_iters):
                if r == 0:
                    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
                    t = time.perf_counter(); cur = estimate(views, len(ids), ref, -1, None, None); t_cpu += time.perf_counter() - t
                else:
                    prev = {v: rounds[r - 1][v][0] for v in allv}
                    d_in, n_in = (cur[0], cur[1]) if chain else (rounds[r - 1][ref][0], rounds[r - 1][ref][1])
                    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps=prev)
                    t = time.perf_counter(); cur = estimate(views, len(ids), ref, r - 1, d_in, n_in); t_cpu += time.perf_counter() - t
            outs[ref] = cur
        return outs, t_cpu

    out = {}
    # ---- the reference's own code, threaded, at 1920x1080 ---------------------------------------------------------------------------------------------------
    base = None
    try:
        from oracle import pyref as pr
        have_ref = pr.driver_available("libm")
    except Exception:
        have_ref = False
    if have_ref:
        W, H = a.width, a.height
        sc = synth.make_scene(9, W, H, n_src=8, device="cuda", gray_only=False)
        rounds = engine_rounds(sc)
        refs = [4, 0, 2, 8]

        def est_ref(views, n, ref, geo, d, nrm):
            opt = po.default_opt(seed=seed, viewID=ref, nThreads=cores, rngMode=2, nEstimationGeometricIters=a.geo_iters)
            return pr.ref_estimate_depth_map(views, n, float(sc.dmin[ref]), float(sc.dmax[ref]), opt, geo_iter=geo, depth=d, normal=nrm, kind="libm")
        # (chain = True: a view's geometric rounds start from the reference code's OWN maps of the round before -- the full schedule of that view as the reference runs it; the
        # neighbours' previous-round maps it reads are the engine's.  Same work as the decoupled form, so the timing is unchanged, and the result is what `tolerance` compares.)
        outs, t_ref = cpu_schedule(sc, rounds, refs, cores, est_ref, True)
        base = {"value": round(len(refs) * W * H / t_ref / 1e6, 5), "unit": "Mpix/s", "cores": cores, "kind": "reference",
                "sample": "%d reference views x 8 sources at %dx%d, photometric pass (3-level pyramid x 3 sweeps) + %d geometric rounds each, through the reference's own "
                          "DepthMapsData::EstimateDepthMap / DepthEstimator code (oracle/_ref: verbatim line ranges of SceneDensify.cpp and DepthMap.cpp, g++ -O3 -march=x86-64-v3, "
                          "libm, std::mt19937) with scene.nMaxThreads = %d estimator threads on the shared pixel counter; cv::resize = the oracle's restatement; %.1f s CPU wall"
                          % (len(refs), W, H, a.geo_iters, cores, t_ref)}
        # ---- what the 1e-4 x diameter tolerance means against a reference binary (information; the gate is bit-identity with the oracle) --------------------
        def cmp(x, y):
            m = (x > 0) & (y > 0)
            ad = np.abs(x[m].astype(np.float64) - y[m]) / sc.diameter
            if not m.any():
                return {"depth_rmse_over_diameter": float("nan")}
            p95 = float(np.percentile(ad, 95))
            core = ad[ad < 10.0 * p95] if p95 > 0 else ad          # outlier-robust figure: pixels valid in both maps whose difference is below 10 x the 95th percentile
            return {"depth_rmse_over_diameter": float(np.sqrt(np.mean(ad ** 2))), "median_abs_over_diameter": float(np.median(ad)),
                    "p95_abs_over_diameter": p95, "frac_within_1e-4": float((ad <= 1e-4).mean()),
                    "robust_rmse_over_diameter": float(np.sqrt(np.mean(core ** 2))) if core.size else float("nan"), "robust_excluded": int(ad.size - core.size),
                    "valid_in_only_one": int(((x > 0) != (y > 0)).sum()), "valid_in_both": int(m.sum())}
        v0 = refs[0]
        ids0 = [v0] + list(sc.neighbors[v0])
        views0, keep0 = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids0)
        t = time.perf_counter()
        ref_a = est_ref(views0, len(ids0), v0, -1, None, None)          # two runs of the reference's photometric pass: its racy thread schedule makes them differ
        ref_b = est_ref(views0, len(ids0), v0, -1, None, None)
        t_tol = time.perf_counter() - t
        hip = rounds[0][v0]

        def vs_gt_of(x, v):
            m = x > 0
            g = sc.gt_depth[v]
            return {"depth_rmse_over_diameter": float(np.sqrt(np.mean((x[m].astype(np.float64) - g[m]) ** 2))) / sc.diameter, "valid_frac": float(m.mean())}

        def vs_gt(x):
            return vs_gt_of(x, v0)
        # the FULL schedule (photometric pass + the geometric rounds, which damp outliers): the final maps of three views (centre view 4, border views 0 and 2), HIP vs the
        # reference's code, and the reference's code against a second full run of itself on every one of them (its racy threads and mt19937 make two runs differ)
        t = time.perf_counter()
        again, _ = cpu_schedule(sc, rounds, refs[:3], cores, est_ref, True)
        t_tol += time.perf_counter() - t
        full = {"views": refs[:3],
                "hip_vs_reference_code": {str(v): cmp(rounds[-1][v][0], outs[v][0]) for v in refs[:3]},
                "reference_code_run_a_vs_run_b": {str(v): cmp(outs[v][0], again[v][0]) for v in refs[:3]},
                "hip_vs_ground_truth": {str(v): vs_gt_of(rounds[-1][v][0], v) for v in refs[:3]},
                "reference_code_vs_ground_truth": {str(v): vs_gt_of(outs[v][0], v) for v in refs[:3]}}
        # the OPT-IN tiled sweeps (another estimator: tiled_sweeps) on the same scene, against the same reference runs
        rounds_t = engine_rounds(sc, tiles=64)
        full["hip_tiled64_vs_reference_code"] = {str(v): cmp(rounds_t[-1][v][0], outs[v][0]) for v in refs[:3]}
        full["hip_tiled64_vs_hip"] = {str(v): cmp(rounds_t[-1][v][0], rounds[-1][v][0]) for v in refs[:3]}
        full["hip_tiled64_vs_ground_truth"] = {str(v): vs_gt_of(rounds_t[-1][v][0], v) for v in refs[:3]}
        out["tolerance"] = {"case": "view %d of the 9-view %dx%d scene, photometric pass (end-of-pass threshold x 1.333), depth maps" % (v0, W, H),
                            "hip_vs_reference_code": cmp(hip[0], ref_a[0]), "reference_code_run_a_vs_run_b": cmp(ref_a[0], ref_b[0]),
                            "hip_vs_ground_truth": vs_gt(hip[0]), "reference_code_vs_ground_truth": vs_gt(ref_a[0]),
                            "full_schedule": full,
                            "north_star_tolerance": 1e-4, "seconds": round(t_tol, 1),
                            "note": "reference code = oracle/_ref (libm, std::mt19937, %d racy threads); HIP = this engine (pm_math.h, Philox).  The estimator is chaotic pixel by pixel: the "
                                    "reference does not reproduce ITSELF within 1e-4 x diameter from run to run, so the tolerance is met by construction against the sequential oracle "
                                    "(bit-identical) and reported here against the reference's own run-to-run spread" % cores}
    # ---- the restated port with the same threading, smaller sample (kept beside the reference figure) -------------------------------------------------------
    nat, flags = native_oracle()
    n_ref = 4
    target_px = min(a.cpu_seconds, 8.0) * 0.008e6 * cores / n_ref      # per-thread rate of the oracle for the full schedule at N = 8 is ~0.008 Mpix/s on this class of host
    scale = min(1.0, (target_px / (a.width * a.height)) ** 0.5)
    sw = max(64, int(a.width * scale) // 16 * 16); sh = max(48, int(a.height * scale) // 16 * 16)
    scp = synth.make_scene(9, sw, sh, n_src=8, device="cuda", gray_only=True)
    rounds_p = engine_rounds(scp)

    def est_port(lib, threads):
        def f(views, n, ref, geo, d, nrm):
            saved = po._LIB; po._LIB = lib
            try:
                opt = po.default_opt(seed=seed, viewID=ref, nThreads=threads, nEstimationGeometricIters=a.geo_iters)
                return po.estimate_depth_map(views, n, float(cur_sc.dmin[ref]), float(cur_sc.dmax[ref]), opt, geo_iter=geo, depth=d, normal=nrm)
            finally:
                po._LIB = saved
        return f
    cur_sc = scp
    refs_p = [4, 0, 2, 8][:n_ref]
    _, t_port = cpu_schedule(scp, rounds_p, refs_p, cores, est_port(nat, cores), False)
    port = {"value": round(len(refs_p) * sw * sh / t_port / 1e6, 5), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": "%d reference views x 8 sources at %dx%d, full schedule, oracle/pm_oracle.cpp built %s with the reference's threading model, %.1f s CPU wall" % (len(refs_p), sw, sh, flags, t_port)}
    if base is None:
        base = port
    else:
        base["port"] = port
    out["cpu_baseline"] = base
    # ---- live parity leg: small, sequential (deterministic) oracle; the engine must match it bit for bit -----------------------------------------------------
    sc2 = synth.make_scene(9, 256, 144, n_src=8, device="cuda", gray_only=True)
    rounds2 = engine_rounds(sc2)
    cur_sc = sc2
    cpu2, _ = cpu_schedule(sc2, rounds2, [4], 1, est_port(po.lib(), 1), True)
    cpu2 = cpu2[4]; gpu2 = rounds2[-1][4]
    m = (cpu2[0] > 0) & (gpu2[0] > 0)
    rmse = float(np.sqrt(np.mean((cpu2[0][m].astype(np.float64) - gpu2[0][m]) ** 2))) if m.any() else float("nan")
    out["parity_live"] = {"case": "256x144, 8 sources, full schedule, sequential oracle vs HIP engine in this run", "depth_rmse_over_diameter": rmse / sc2.diameter,
                          "tolerance": 1e-4, "pixels_valid_in_only_one": int(((cpu2[0] > 0) != (gpu2[0] > 0)).sum()),
                          "bit_identical": bool(np.array_equal(cpu2[0], gpu2[0]) and np.array_equal(cpu2[1], gpu2[1]) and np.array_equal(cpu2[2], gpu2[2]))}
    return out


if __name__ == "__main__":
    main()

"""Multi-GPU driver: reference views sharded over ranks, one process per GPU.

The path shards by reference view (each depth map depends only on read-only images and on the
*previous round's* depth maps of its neighbours, SceneDensify.cpp:378-393), so there is no
per-iteration collective: one broadcast of the image set at start-up and one exchange of depth
maps at each round boundary (what the reference does through depthNNNN.dmap files,
SceneDensify.cpp:1943-1950).  torch.distributed is plumbing: backend "nccl" is RCCL over xGMI on
the GPU box, "gloo" in the CPU tests.

Two exchange modes:
  * all-gather (neighbors=None): every rank ends up with the maps of ALL views -- simple, and right while the
    whole scene fits every GPU;
  * neighbour-only (neighbors given): a rank receives only the maps of the views its block reads -- the source
    views of its reference views that live on other ranks -- by point-to-point messages, and need not hold
    anything else of the scene (`needed_views`): BASELINE config 5's 300 x 4K scene is ~46 GB per rank this
    way instead of ~140 GB.

`ShardedDensifier` is backend-agnostic: it drives any estimator object with the methods listed in its
docstring (the HIP engine adapter in bench.py, or a CPU stand-in in tests/test_distributed.py).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_views: int, world: int, rank: int) -> range:
    """Contiguous block of reference views owned by `rank` (blocks differ by at most one view)."""
    base, rem = divmod(n_views, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def owner_of(view: int, n_views: int, world: int) -> int:
    base, rem = divmod(n_views, world)
    cut = rem * (base + 1)
    return view // (base + 1) if view < cut else rem + (view - cut) // max(base, 1)


def needed_views(neighbors, n_views: int, world: int, rank: int):
    """(mine, foreign): this rank's block, and the views of other ranks that its reference views read as sources (sorted).  mine + foreign is all a rank has to hold."""
    mine = list(shard_range(n_views, world, rank))
    own = set(mine)
    foreign = sorted({int(n) for v in mine for n in neighbors[v]} - own)
    return mine, foreign


def _collectives_on(world: int) -> bool:
    return world > 1 or (os.environ.get("OPENMVS_AMD_FORCE_COLLECTIVES") == "1" and dist.is_initialized())


def all_gather_views(mine: torch.Tensor, n_views: int, world: int, rank: int) -> torch.Tensor:
    """mine: [len(shard), H, W] depth maps of this rank's block -> [n_views, H, W] on every rank."""
    if not _collectives_on(world):
        return mine
    sizes = [len(shard_range(n_views, world, r)) for r in range(world)]
    if len(set(sizes)) == 1 and dist.get_backend() == "nccl":
        out = torch.empty((n_views,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(out, mine.contiguous())
        return out
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    pad[:mine.shape[0]] = mine
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], 0)


def exchange_neighbour_views(mine_maps, neighbors, n_views: int, world: int, rank: int, view_shapes=None):
    """Point-to-point exchange of exactly the maps each rank reads: returns (view ids, [len(ids), H, W] tensor) of the FOREIGN views this rank needs, in ascending id.
    mine_maps: [len(block), H, W] maps of this rank's block, in block order.  Every rank derives the same send / receive lists from `neighbors`, so no metadata travels.
    Views of different sizes: mine_maps is a LIST of [h_v, w_v] tensors and view_shapes[v] = (h_v, w_v) for every view of the scene (known everywhere, like the cameras);
    one message per view then, and the foreign maps come back as a list."""
    mine, foreign = needed_views(neighbors, n_views, world, rank)
    if isinstance(mine_maps, (list, tuple)):
        return foreign, _exchange_view_lists(list(mine_maps), mine, foreign, neighbors, n_views, world, rank, view_shapes)
    if not _collectives_on(world):
        return foreign, mine_maps[:0]
    lo = mine[0] if mine else 0
    # gloo has no point-to-point for device tensors: stage through the host then (the functional check of the N-rank path on a box with fewer GPUs; RCCL sends device memory)
    staged = mine_maps.device.type != "cpu" and dist.get_backend() == "gloo"
    src = mine_maps.cpu() if staged else mine_maps
    recv = torch.empty((len(foreign),) + tuple(mine_maps.shape[1:]), dtype=mine_maps.dtype, device=src.device)
    ops, keep = [], []
    for peer in range(world):
        if peer == rank:
            continue
        _, their_foreign = needed_views(neighbors, n_views, world, peer)
        to_send = [v for v in their_foreign if owner_of(v, n_views, world) == rank]          # ascending: the receiver's order
        if to_send:
            buf = src[[v - lo for v in to_send]].contiguous(); keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, peer))
        idx = [k for k, v in enumerate(foreign) if owner_of(v, n_views, world) == peer]
        if idx:
            # (the views a peer owns are a contiguous id range, so they are a contiguous run of the sorted foreign list)
            ops.append(dist.P2POp(dist.irecv, recv[idx[0]:idx[-1] + 1], peer))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return foreign, (recv.to(mine_maps.device) if staged else recv)


def _exchange_view_lists(mine_maps, mine, foreign, neighbors, n_views, world, rank, view_shapes):
    if not _collectives_on(world) or not (mine_maps or foreign):
        return []
    like = mine_maps[0] if mine_maps else None
    staged = like is not None and like.device.type != "cpu" and dist.get_backend() == "gloo"
    dev = torch.device("cpu") if (staged or like is None) else like.device
    dtype = like.dtype if like is not None else torch.float32
    lo = mine[0] if mine else 0
    recv = [torch.empty(tuple(view_shapes[v]), dtype=dtype, device=dev) for v in foreign]
    ops, keep = [], []
    for peer in range(world):
        if peer == rank:
            continue
        _, their_foreign = needed_views(neighbors, n_views, world, peer)
        for v in their_foreign:                                                     # ascending: messages between a pair of ranks are matched in order
            if owner_of(v, n_views, world) == rank:
                buf = (mine_maps[v - lo].cpu() if staged else mine_maps[v - lo]).contiguous(); keep.append(buf)
                ops.append(dist.P2POp(dist.isend, buf, peer))
        for k, v in enumerate(foreign):
            if owner_of(v, n_views, world) == peer:
                ops.append(dist.P2POp(dist.irecv, recv[k], peer))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return [r.to(like.device) for r in recv] if staged else recv


def gather_views_to_root(mine_maps, n_views: int, world: int, rank: int, root: int = 0, view_shapes=None):
    """Every rank's block of maps (a [len(block), H, W] tensor or a list of [h_v, w_v] tensors) -> a list of all n_views maps on `root`, None on the other ranks."""
    maps = list(mine_maps) if isinstance(mine_maps, (list, tuple)) else [mine_maps[k] for k in range(mine_maps.shape[0])]
    if not _collectives_on(world):
        return maps
    like = maps[0] if maps else None
    staged = like is not None and like.device.type != "cpu" and dist.get_backend() == "gloo"
    if rank != root:
        ops, keep = [], []
        for m in maps:
            buf = (m.cpu() if staged else m).contiguous(); keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, root))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return None
    if view_shapes is None:
        if like is None:
            raise ValueError("gather_views_to_root: the root holds no view and no view_shapes were given")
        view_shapes = [tuple(like.shape)] * n_views
    dev = torch.device("cpu") if (staged or like is None) else like.device
    dtype = like.dtype if like is not None else torch.float32
    out, ops = [None] * n_views, []
    for peer in range(world):
        for k, v in enumerate(shard_range(n_views, world, peer)):
            if peer == rank:
                out[v] = maps[k]
            else:
                out[v] = torch.empty(tuple(view_shapes[v]), dtype=dtype, device=dev)
                ops.append(dist.P2POp(dist.irecv, out[v], peer))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return [m.to(like.device) if (staged and m.device != like.device) else m for m in out]


class ShardedDensifier:
    """Photometric pass + `geo_iters` geometric rounds over this rank's block of views.

    estimator must provide:
      reset(view_ids)                       -- maps of these views back to "unset"
      estimate(view_ids, geo_iter)          -- one EstimateDepthMap per view (geo_iter -1 = photometric)
      local_depths(view_ids) -> Tensor      -- [len(ids), H, W] current depth maps of these views
      set_snapshot(all_depths: Tensor)      -- previous-round depth maps of ALL views, for the next round
    and for `filter()` (BASELINE config 5's exchange before the cross-view filter):
      local_maps(view_ids, "depth"|"conf") -> Tensor   -- [len(ids), H, W] current maps of these views
      set_maps("depth"|"conf", all: Tensor)            -- install the maps of ALL views (the neighbours' unfiltered maps the filter reads)
      filter(view_ids)                                 -- DepthMapsData::FilterDepthMap for these views, results installed
    With `neighbors` (neighbour-only exchange) instead of the two "ALL views" methods:
      set_snapshot_views(own_ids, own: Tensor, foreign_ids, foreign: Tensor)   -- previous-round depth maps of this rank's block and of the foreign views it reads
      set_maps_views(what, foreign_ids, foreign: Tensor)                        -- the foreign neighbours' unfiltered maps for the filter (its own are in place)
    """

    def __init__(self, estimator, n_views: int, world: int = 1, rank: int = 0, geo_iters: int = 2, neighbors=None, view_shapes=None):
        self.est, self.n_views, self.world, self.rank, self.geo_iters = estimator, n_views, world, rank, geo_iters
        self.mine = list(shard_range(n_views, world, rank))
        self.neighbors = neighbors
        self.view_shapes = view_shapes       # (h, w) of every view's maps when they differ (neighbour-only exchange; the estimator then hands lists of tensors)
        self.exchange_seconds = 0.0          # communication + installing the received maps (the round boundary proper)
        self.wait_seconds = 0.0              # waiting for this rank's own asynchronous estimate before its maps can be read (kernel time, not communication)

    def exchange(self):
        import time
        t0 = time.perf_counter()
        own = self.est.local_depths(self.mine)          # blocks until the round's estimate has finished
        t = time.perf_counter()
        self.wait_seconds += t - t0
        if self.neighbors is None:
            self.est.set_snapshot(all_gather_views(own, self.n_views, self.world, self.rank))
        else:
            ids, maps = exchange_neighbour_views(own, self.neighbors, self.n_views, self.world, self.rank, self.view_shapes)
            self.est.set_snapshot_views(self.mine, own, ids, maps)
        self.exchange_seconds += time.perf_counter() - t

    def run(self):
        self.est.reset(self.mine)
        self.est.estimate(self.mine, -1)
        for g in range(self.geo_iters):
            self.exchange()
            self.est.estimate(self.mine, g)

    def filter(self):
        """Scene::DenseReconstructionFilter (SceneDensify.cpp:2136-2222) sharded by view: every rank filters its own depth maps against the UNFILTERED
        depth and confidence maps of their neighbours (the reference writes *.filtered.dmap files and renames them only when all are done), so one
        exchange of depth and one of confidence precede the filter; the filtered maps stay with their owner (collect them on the fusing rank with `gather("depth", root=0)`)."""
        for what in ("depth", "conf"):
            own = self.est.local_maps(self.mine, what)
            if self.neighbors is None:
                self.est.set_maps(what, all_gather_views(own, self.n_views, self.world, self.rank))
            else:
                ids, maps = exchange_neighbour_views(own, self.neighbors, self.n_views, self.world, self.rank, self.view_shapes)
                self.est.set_maps_views(what, ids, maps)
        self.est.filter(self.mine)

    def gather(self, what, root=None):
        """Maps of all views for FuseDepthMaps, which is sequential over the scene and therefore runs on one rank (SceneDensify.cpp:1372-1650).
        root=None, uniform sizes: [n_views, H, W] on every rank (one all-gather).  root=r: only rank r receives -- a list of n_views maps there, None elsewhere --
        by one point-to-point message per view, which is also the only form for views of different sizes (`view_shapes`) and the one that keeps a
        neighbour-only rank from holding the whole scene."""
        own = self.est.local_maps(self.mine, what)
        if root is None:
            if isinstance(own, (list, tuple)):
                raise ValueError("gather(%r): views of different sizes have no [n_views, H, W] tensor -- pass root= to collect them view by view on the fusing rank" % what)
            return all_gather_views(own, self.n_views, self.world, self.rank)
        return gather_views_to_root(own, self.n_views, self.world, self.rank, root, self.view_shapes)


class EngineRank:
    """One rank's HIP engine behind `ShardedDensifier` (neighbour-only exchange): a COMPACT scene of local slots -- this rank's block of reference views first, then the
    foreign views its block reads -- so that a rank holds nothing else of the scene (BASELINE config 5).  Every slot draws its random numbers under the view's index in the
    whole scene (`scene_set_view_id`), which makes the depth maps independent of the split.  Maps travel as tensors on `device`, handed to the engine by pointer
    (`scene_copy`): device memory under RCCL, host memory when the engine is the CPU emulator of the tests.  bench.py's adapter is these same calls, inlined (and timed).

    engine: a `PatchMatchHIP`; params: `PMHipParams`; neighbors: per view of the WHOLE scene, the global ids of its source views (the images themselves);
    gray_of(g): the gray image of global view g (asked only for the views this rank holds); K, R, C, dmin, dmax: indexable by global id.

    Views of different sizes: sizes[g] = (w, h) of every slot id (a view whose size is not the scene's carries its own: pmhip_scene_set_view_sized); the maps of such a block
    travel as lists of [h_v, w_v] tensors (`view_shapes`).  Resampled neighbour copies (ViewData::ScaleImage, densify.load_scene): alias_of = {copy id >= n_views: image id} and
    estimate_neighbors = the lists the estimation reads (a copy in place of its image); a rank holds the copies its block reads as source-only slots behind the foreign
    views, hands each one the depth map of the image it stands for at every round boundary (with that image's camera, the neighbour's saved .dmap of SceneDensify.cpp:378-393)
    and gives its reference views their image neighbours back before the cross-view filter -- what densify.compute_depth_maps does on one engine."""

    def __init__(self, engine, params, n_views, world, rank, neighbors, gray_of, K, R, C, dmin, dmax, width, height, n_levels=2, device="cpu", batch=0,
                 filter_args=(True, 2, 1, 0.01), init_depth=None, init_normal=None, masks=None, mask_option=False, sizes=None, alias_of=None, estimate_neighbors=None):
        self.eng, self.p, self.W, self.H, self.batch, self.filter_args = engine, params, int(width), int(height), int(batch), tuple(filter_args)
        self.init_depth, self.init_normal = init_depth or {}, init_normal or {}       # seed maps by global view id (InitViews, SceneDensify.cpp:418-460), installed by reset()
        self.device = torch.device(device)
        nbs = [[int(x) for x in neighbors[v]] for v in range(n_views)]
        self.alias_of = {int(a): int(j) for a, j in (alias_of or {}).items()}
        est = [[int(x) for x in (estimate_neighbors[v] if estimate_neighbors is not None else nbs[v])] for v in range(n_views)]
        self.mine, self.foreign = needed_views(nbs, n_views, world, rank)
        self.copies = sorted({a for v in self.mine for a in est[v] if a in self.alias_of})     # the resampled copies this block reads
        self.held = self.mine + self.foreign + self.copies      # global ids in slot order
        self.slot = {g: i for i, g in enumerate(self.held)}
        self.K, self.R, self.C, self.dmin, self.dmax, self.nbs, self.est = K, R, C, dmin, dmax, nbs, est
        size_of = (lambda g: (int(sizes[g][0]), int(sizes[g][1]))) if sizes is not None else (lambda g: (self.W, self.H))
        self.shape = {g: size_of(g)[::-1] for g in self.held}                                  # (h, w) of a held view's maps
        self.mixed = any(size_of(g) != (self.W, self.H) for g in range(n_views))               # some image has its own size: every rank's maps travel view by view (one protocol for all ranks)
        engine.Init(True)
        engine.scene_create(max(2, len(self.held)), self.W, self.H, n_levels)
        for i, g in enumerate(self.held):
            own_size = self.shape[g] != (self.H, self.W)
            (engine.scene_set_view_sized if own_size else engine.scene_set_view)(i, gray_of(g), K[g], R[g], C[g], float(dmin[g]), float(dmax[g]),
                                                                                [self.slot[n] for n in est[g]] if i < len(self.mine) else [])
            engine.scene_set_view_id(i, g)
        # ignore masks of a scene loaded with --ignore-mask-label (densify.load_scene), by global view id: installed for the slots this rank holds; the option alone already selects
        # the nearest-neighbour level hand-off (SceneDensify.cpp:661) -- as PatchMatchHIP.scene_load does on one engine
        for g, m in (masks or {}).items():
            if int(g) in self.slot:
                engine.scene_set_mask(self.slot[int(g)], m)
        if mask_option:
            engine.scene_set_mask_mode(1)
        engine.sync()
        self.buf = torch.empty((max(1, len(self.mine)), self.H, self.W), dtype=torch.float32, device=self.device)

    def _fence(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize()                            # a received tensor is complete before the engine's own stream copies out of it

    def reset(self, ids):
        for v in ids:
            self.eng.scene_reset_view(self.slot[v])
            if v in self.init_depth:
                self.eng.scene_set_maps(self.slot[v], self.init_depth[v], self.init_normal.get(v))

    def post_filters(self, ids, n_optimize=3, n_speckle_size=100, n_ipol_gap_size=7, f_depth_diff_threshold=0.01):
        """The per-map filters of the last round (nOptimize bits REMOVE_SPECKLES = 1, FILL_GAPS = 2; SceneDensify.cpp:2069-2093) on this rank's own maps."""
        s = [self.slot[v] for v in ids]
        if s and n_optimize & 1:
            self.eng.scene_remove_small_segments(s, n_speckle_size, f_depth_diff_threshold)
        if s and n_optimize & 2:
            self.eng.scene_gap_interpolation(s, n_ipol_gap_size, f_depth_diff_threshold)

    def estimate(self, ids, geo):
        s = [self.slot[v] for v in ids]
        step = self.batch if self.batch > 0 else max(1, len(s))
        for i in range(0, len(s), step):
            self.eng.scene_estimate(s[i:i + step], geo, self.p, sync=False)

    def local_maps(self, ids, what):
        if list(ids) != self.mine:
            raise ValueError("a rank hands out the maps of its own block")
        if self.mixed:                                          # views of different sizes: one tensor per view
            kind = {"depth": 1, "normal": 2, "conf": 3}[what]
            out = []
            for i, g in enumerate(self.mine):
                t = torch.empty(self.shape[g] + ((3,) if what == "normal" else ()), dtype=torch.float32, device=self.device)
                self.eng.scene_copy(kind, i, 1, t.data_ptr(), False)
                out.append(t)
            self.eng.sync()
            return out
        if what == "normal":                                    # (only the fusing rank ever asks: its own buffer)
            buf = torch.empty((len(self.mine), self.H, self.W, 3), dtype=torch.float32, device=self.device)
        else:
            buf = self.buf[:len(self.mine)]
        if len(self.mine):
            self.eng.scene_copy({"depth": 1, "normal": 2, "conf": 3}[what], 0, len(self.mine), buf.data_ptr(), False)
        self.eng.sync()                                         # (also the wait for this rank's asynchronous estimate)
        return buf

    def local_depths(self, ids):
        return self.local_maps(ids, "depth")

    def set_snapshot_views(self, own_ids, own, foreign_ids, foreign_maps):
        # previous-round depth maps become visible to the next round (the reference writes depthNNNN.dmap and re-reads the neighbours' files, SceneDensify.cpp:378-393,
        # 1943-1950): this rank's own maps by a device copy inside the engine, the foreign ones from the exchange
        self.eng.scene_commit_round()
        self._fence()
        for k, g in enumerate(foreign_ids):
            self.eng.scene_copy(4, self.slot[g], 1, foreign_maps[k].data_ptr(), True)
        for a in self.copies:                                   # a resampled copy reads the depth map of the image it stands for, at that image's size and camera
            j = self.alias_of[a]
            d = own[list(own_ids).index(j)] if j in self.slot and self.slot[j] < len(self.mine) else foreign_maps[list(foreign_ids).index(j)]
            self.eng.scene_set_source_depth(self.slot[a], d.cpu().numpy(), self.K[j], self.R[j], self.C[j])
        self.eng.sync()                                         # the received tensor may be released by the caller

    def set_maps_views(self, what, foreign_ids, foreign_maps):
        self._fence()
        for k, g in enumerate(foreign_ids):
            self.eng.scene_copy({"depth": 1, "conf": 3}[what], self.slot[g], 1, foreign_maps[k].data_ptr(), True)
        self.eng.sync()

    def filter(self, ids):
        for v in ids:                                           # FilterDepthMap reads the images themselves (arrDepthData[ID], SceneDensify.cpp:1049-1299), not their resampled copies
            if self.est[v] != self.nbs[v]:
                self.eng.scene_set_view(self.slot[v], None, self.K[v], self.R[v], self.C[v], float(self.dmin[v]), float(self.dmax[v]), [self.slot[n] for n in self.nbs[v]])
        if len(ids):
            b_adjust, n_min, n_min_adjust, f_depth = self.filter_args
            self.eng.scene_filter([self.slot[v] for v in ids], b_adjust, n_min, n_min_adjust, f_depth, commit=True)

    def maps_of(self, g):
        """(depth, normal, conf) of global view g of this rank's block, on the host."""
        return self.eng.scene_get_maps(self.slot[g])


def dense_reconstruction(engine, scene, opt, world: int = 1, rank: int = 0, seed: int = 0, root: int = 0, device="cpu", fuse_engine=None):
    """The PatchMatch path of `Scene::DenseReconstruction` (libs/MVS/SceneDensify.cpp:1655-1750) over `world` ranks, one engine each: every rank estimates its block of
    reference views (photometric pass, geometric rounds with the neighbour-only exchange), runs the per-map filters on its own maps and the cross-view filter against its
    neighbours' unfiltered maps; the fusing rank `root` then collects depth, normal and confidence maps view by view and fuses them (FuseDepthMaps is sequential over the
    scene, :1372-1650).  Same maps and the same cloud as `densify.dense_reconstruction` on one engine, whatever the split.

    scene: a `densify.SceneViews` (every rank runs `densify.load_scene` on the same archive; a rank's engine is handed only the images it holds); its views may differ in size
    and read resampled copies of their neighbours (`alias_of`, ViewData::ScaleImage); opt: an `optdense.OptDense`; fuse_engine: the engine `root` fuses on (default:
    `engine`, whose compact scene is replaced by the whole one).  Returns the cloud on `root`, None elsewhere."""
    from . import densify
    alias_of = dict(getattr(scene, "alias_of", None) or {})
    n = len(scene.gray) - len(alias_of)                                         # the images; the resampled copies follow them as source-only slots
    if list(scene.ids) != list(range(n)):
        raise NotImplementedError("the sharded driver takes scenes all of whose views passed view selection")
    nbs = [[int(x) for x in scene.neighbors[v]] for v in range(n)]
    sizes = [tuple(wh) for wh in scene.sizes] if getattr(scene, "sizes", None) else None
    mixed = sizes is not None and len(set(sizes[:n])) > 1
    G = int(opt.nEstimationGeometricIters)
    est = EngineRank(engine, opt.params(seed), n, world, rank, nbs, lambda g: scene.gray[g], scene.K, scene.R, scene.C, scene.dmin, scene.dmax, scene.width, scene.height,
                     n_levels=int(opt.nSubResolutionLevels), device=device,
                     filter_args=(bool(opt.bFilterAdjust), int(opt.nMinViewsFilter), int(opt.nMinViewsFilterAdjust), float(opt.fDepthDiffThreshold)),
                     init_depth=scene.init_depth, init_normal=scene.init_normal, masks=getattr(scene, "masks", None), mask_option=bool(getattr(scene, "mask_option", False)),
                     sizes=sizes, alias_of=alias_of, estimate_neighbors=getattr(scene, "estimate_neighbors", None) if alias_of else None)
    shapes = [tuple(sizes[v])[::-1] for v in range(n)] if mixed else None         # (h, w) of every image's maps: known on every rank, like the cameras
    drv = ShardedDensifier(est, n, world, rank, geo_iters=G, neighbors=nbs, view_shapes=shapes)
    drv.run()
    est.post_filters(drv.mine, int(opt.nOptimize), int(opt.nSpeckleSize), int(opt.nIpolGapSize), float(opt.fDepthDiffThreshold))
    if int(opt.nOptimize) & densify.ADJUST_FILTER:
        drv.filter()

    def own(w):
        m = est.local_maps(est.mine, w)
        return [t.clone() for t in m] if isinstance(m, (list, tuple)) else m.clone()
    maps = {w: gather_views_to_root(own(w), n, world, rank, root, [s + (3,) for s in shapes] if (shapes and w == "normal") else shapes) for w in ("depth", "normal", "conf")}
    if rank != root:
        return None
    fe = fuse_engine or engine
    fe.scene_load(scene, n_levels=int(opt.nSubResolutionLevels))
    for v in range(n):
        if alias_of and not np.array_equal(scene.estimate_neighbors[v], scene.neighbors[v]):     # fusion reads the images themselves, as the filter did
            fe.scene_set_view(v, None, scene.K[v], scene.R[v], scene.C[v], float(scene.dmin[v]), float(scene.dmax[v]), scene.neighbors[v])
        fe.scene_set_maps(v, maps["depth"][v].cpu().numpy(), maps["normal"][v].cpu().numpy())
        fe.scene_set_conf(v, maps["conf"][v].cpu().numpy())
    return densify.fuse_depth_maps(fe, scene, opt, bgr=scene.bgr)

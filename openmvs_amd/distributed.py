"""Multi-GPU driver: reference views sharded over ranks, one process per GPU.

The path shards by reference view (each depth map depends only on read-only images and on the
*previous round's* depth maps of its neighbours, SceneDensify.cpp:378-393), so there is no
per-iteration collective: one broadcast of the image set at start-up and one all-gather of the
depth maps at each round boundary (what the reference does through depthNNNN.dmap files,
SceneDensify.cpp:1943-1950).  torch.distributed is plumbing: backend "nccl" is RCCL over xGMI on
the GPU box, "gloo" in the CPU tests.

`ShardedDensifier` is backend-agnostic: it drives any estimator object with the four methods used
below (the HIP engine adapter in bench.py, or a CPU stand-in in tests/test_distributed.py).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(n_views: int, world: int, rank: int) -> range:
    """Contiguous block of reference views owned by `rank` (blocks differ by at most one view)."""
    base, rem = divmod(n_views, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def all_gather_views(mine: torch.Tensor, n_views: int, world: int, rank: int) -> torch.Tensor:
    """mine: [len(shard), H, W] depth maps of this rank's block -> [n_views, H, W] on every rank."""
    if world == 1 and not (os.environ.get("OPENMVS_AMD_FORCE_COLLECTIVES") == "1" and dist.is_initialized()):
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
    """

    def __init__(self, estimator, n_views: int, world: int = 1, rank: int = 0, geo_iters: int = 2):
        self.est, self.n_views, self.world, self.rank, self.geo_iters = estimator, n_views, world, rank, geo_iters
        self.mine = list(shard_range(n_views, world, rank))

    def exchange(self):
        self.est.set_snapshot(all_gather_views(self.est.local_depths(self.mine), self.n_views, self.world, self.rank))

    def run(self):
        self.est.reset(self.mine)
        self.est.estimate(self.mine, -1)
        for g in range(self.geo_iters):
            self.exchange()
            self.est.estimate(self.mine, g)

    def filter(self):
        """Scene::DenseReconstructionFilter (SceneDensify.cpp:2136-2222) sharded by view: every rank filters its own depth maps against the UNFILTERED
        depth and confidence maps of their neighbours (the reference writes *.filtered.dmap files and renames them only when all are done), so one
        all-gather of depth and one of confidence precede the filter; the filtered maps stay with their owner (gather them with `gather("depth")`)."""
        for what in ("depth", "conf"):
            self.est.set_maps(what, all_gather_views(self.est.local_maps(self.mine, what), self.n_views, self.world, self.rank))
        self.est.filter(self.mine)

    def gather(self, what):
        """[n_views, H, W] maps of all views on every rank (e.g. the filtered maps before FuseDepthMaps, which is sequential over the scene and
        therefore runs on one rank, SceneDensify.cpp:1372-1650)."""
        return all_gather_views(self.est.local_maps(self.mine, what), self.n_views, self.world, self.rank)

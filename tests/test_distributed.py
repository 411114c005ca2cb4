"""N > 1 path on CPU: 2 gloo ranks shard the reference views, exchange depth maps at round
boundaries, and must reproduce the single-process result exactly.  The estimator stand-in is the
CPU oracle (tests may use it); the GPU box runs the same driver with the HIP engine (bench.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openmvs_amd import synth
from openmvs_amd.distributed import ShardedDensifier, all_gather_views, shard_range

SEED = 9


class OracleEstimator:
    def __init__(self, sc):
        from oracle import pyoracle as po
        self.po, self.sc = po, sc
        h, w = sc.height, sc.width
        self.depth = np.zeros((sc.n_views, h, w), np.float32); self.normal = np.zeros((sc.n_views, h, w, 3), np.float32)
        self.conf = np.zeros((sc.n_views, h, w), np.float32); self.snap = None

    def reset(self, ids):
        for v in ids:
            self.depth[v] = 0; self.normal[v] = 0; self.conf[v] = 0

    own = ()

    def estimate(self, ids, geo):
        po, sc = self.po, self.sc
        self.own = list(ids)
        for v in ids:
            vid = [v] + list(sc.neighbors[v])
            src = None if geo < 0 else {i: self.snap[i] for i in vid[1:]}
            views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, vid, depth_maps=src)
            opt = po.default_opt(seed=SEED, viewID=v, nSubResolutionLevels=1)
            self.depth[v], self.normal[v], self.conf[v] = po.estimate_depth_map(views, len(vid), float(sc.dmin[v]), float(sc.dmax[v]), opt, geo_iter=geo,
                                                                                  depth=self.depth[v], normal=self.normal[v])

    def local_depths(self, ids):
        return torch.from_numpy(self.depth[list(ids)].copy())

    def set_snapshot(self, allv):
        self.snap = allv.numpy().copy()

    # neighbour-only exchange: only the rank's own block and the foreign views it reads arrive; everything else is poisoned, so a read of it would show
    def set_snapshot_views(self, own_ids, own, foreign_ids, foreign):
        self.snap = np.full_like(self.depth, np.nan)
        self.snap[list(own_ids)] = own.numpy()
        if len(foreign_ids):
            self.snap[list(foreign_ids)] = foreign.numpy()

    def set_maps_views(self, what, foreign_ids, foreign):
        a = {"depth": self.depth, "conf": self.conf}[what]
        keep = set(self.own) | set(int(v) for v in foreign_ids)
        for v in range(self.sc.n_views):
            if v not in keep:
                a[v] = np.nan
        if len(foreign_ids):
            a[list(foreign_ids)] = foreign.numpy()

    # config 5's exchange: the cross-view filter reads the neighbours' unfiltered depth and confidence maps
    def local_maps(self, ids, what):
        return torch.from_numpy({"depth": self.depth, "conf": self.conf}[what][list(ids)].copy())

    def set_maps(self, what, allv):
        {"depth": self.depth, "conf": self.conf}[what][:] = allv.numpy()

    def filter(self, ids):
        po, sc = self.po, self.sc
        dep, cnf = self.depth.copy(), self.conf.copy()     # every view is filtered against the unfiltered maps (SceneDensify.cpp:2183-2210)
        for v in ids:
            rc, d, c = po.filter_depth_map(dep, cnf, sc.K, sc.R, sc.C, v, list(sc.neighbors[v]), sc.dmin[v], sc.dmax[v])
            assert rc == 0
            self.depth[v], self.conf[v] = d, c


def _scene():
    return synth.make_scene(5, 48, 32, n_src=3)     # 5 views on 2 ranks: blocks of 3 and 2, the padded all-gather path of unequal shards (100 views / 8 GPUs)


def _worker(rank, world, port, out_dir, neighbour_only=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = _scene()
    est = OracleEstimator(sc)
    drv = ShardedDensifier(est, sc.n_views, world, rank, geo_iters=2, neighbors=[list(sc.neighbors[v]) for v in range(sc.n_views)] if neighbour_only else None)
    drv.run()
    final = all_gather_views(est.local_depths(drv.mine), sc.n_views, world, rank)
    # the estimator finalises confidences like EndDepthMapTmp only inside the engine; here give the filter something to weigh
    for v in drv.mine:
        est.conf[v] = np.where(est.depth[v] > 0, np.float32(1) - np.minimum(est.conf[v], np.float32(0.9)), np.float32(0)).astype(np.float32)
    drv.filter()
    fd, fc = drv.gather("depth"), drv.gather("conf")
    rooted = drv.gather("depth", root=0)          # one message per view to the fusing rank only
    assert (rooted is None) == (rank != 0)
    if rank == 0:
        assert len(rooted) == sc.n_views and all(torch.equal(rooted[v], fd[v]) for v in range(sc.n_views))
        np.save(os.path.join(out_dir, "sharded.npy"), final.numpy())
        np.save(os.path.join(out_dir, "filtered_depth.npy"), fd.numpy()); np.save(os.path.join(out_dir, "filtered_conf.npy"), fc.numpy())
    dist.destroy_process_group()


def _single_process_reference():
    sc = _scene()
    est = OracleEstimator(sc)
    drv = ShardedDensifier(est, sc.n_views, 1, 0, geo_iters=2)
    drv.run()
    final = est.depth.copy()
    for v in drv.mine:
        est.conf[v] = np.where(est.depth[v] > 0, np.float32(1) - np.minimum(est.conf[v], np.float32(0.9)), np.float32(0)).astype(np.float32)
    drv.filter()
    return final, est.depth.copy(), est.conf.copy()


def test_neighbour_only_exchange_matches_single_process(tmp_path):
    """The point-to-point exchange of exactly the maps each rank reads (distributed.exchange_neighbour_views), on 2 ranks (blocks of 3 and 2 views) and on 3 ranks
    (2, 2 and 1: uneven, one rank owning a single view): the same bits as one process, with every map a rank does not need poisoned."""
    from openmvs_amd.distributed import needed_views, owner_of
    sc = _scene()
    nbs = [list(sc.neighbors[v]) for v in range(sc.n_views)]
    for world in (2, 3):
        for r in range(world):
            mine, foreign = needed_views(nbs, sc.n_views, world, r)
            assert mine == list(shard_range(sc.n_views, world, r)) and not set(mine) & set(foreign)
            assert all(owner_of(v, sc.n_views, world) != r for v in foreign)
    final, fd, fc = _single_process_reference()
    for world in (2, 3):
        out = tmp_path / ("w%d" % world); out.mkdir()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        mp.spawn(_worker, args=(world, port, str(out), True), nprocs=world, join=True)
        assert np.array_equal(np.load(out / "sharded.npy"), final), "world %d" % world
        assert np.array_equal(np.load(out / "filtered_depth.npy"), fd) and np.array_equal(np.load(out / "filtered_conf.npy"), fc), "world %d filter" % world


def test_shard_ranges_cover_everything():
    for n, w in ((100, 8), (7, 2), (5, 8), (16, 4)):
        got = [v for r in range(w) for v in shard_range(n, w, r)]
        assert got == list(range(n))


def test_two_ranks_match_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sharded = np.load(tmp_path / "sharded.npy")
    sc = _scene()
    est = OracleEstimator(sc)
    drv = ShardedDensifier(est, sc.n_views, 1, 0, geo_iters=2)
    drv.run()
    assert np.array_equal(sharded, est.depth)
    assert (sharded > 0).mean() > 0.3
    # the sharded cross-view filter (depth + confidence all-gather, then FilterDepthMap per owner) equals the single-process filter
    for v in drv.mine:
        est.conf[v] = np.where(est.depth[v] > 0, np.float32(1) - np.minimum(est.conf[v], np.float32(0.9)), np.float32(0)).astype(np.float32)
    drv.filter()
    assert np.array_equal(np.load(tmp_path / "filtered_depth.npy"), est.depth) and np.array_equal(np.load(tmp_path / "filtered_conf.npy"), est.conf)
    assert (est.depth != sharded).any() and (est.depth > 0).mean() > 0.2


# ---- views of different sizes: per-view messages (exchange_neighbour_views with lists) --------------------------------------------------------------
class MixedSizeEstimator:
    """Oracle stand-in for a scene whose views differ in size: every map a [h_v, w_v] array; what a rank was not sent stays None (a read of it fails)."""

    def __init__(self):
        from oracle import pyoracle as po
        self.po = po
        base = synth.make_scene(5, 48, 32, n_src=3); small = synth.make_scene(5, 36, 24, n_src=3); big = synth.make_scene(5, 60, 40, n_src=3)
        src = {0: base, 1: small, 2: base, 3: big, 4: small}
        self.base = base
        self.gray = {v: src[v].gray[v] for v in range(5)}; self.K = {v: src[v].K[v] for v in range(5)}
        self.shapes = [self.gray[v].shape for v in range(5)]
        self.depth = [np.zeros(s, np.float32) for s in self.shapes]; self.normal = [np.zeros(s + (3,), np.float32) for s in self.shapes]
        self.conf = [np.zeros(s, np.float32) for s in self.shapes]; self.snap = [None] * 5

    def reset(self, ids):
        for v in ids:
            self.depth[v][:] = 0; self.normal[v][:] = 0; self.conf[v][:] = 0

    def estimate(self, ids, geo):
        po, b = self.po, self.base
        for v in ids:
            vid = [v] + [int(i) for i in b.neighbors[v]]
            src = None if geo < 0 else {i: self.snap[i] for i in vid[1:]}
            cams = None if geo < 0 else {i: (self.K[i], b.R[i], b.C[i]) for i in vid[1:]}
            views, keep = po.make_views(self.gray, self.K, b.R, b.C, vid, depth_maps=src, depth_cams=cams)
            self.depth[v], self.normal[v], self.conf[v] = po.estimate_depth_map(views, len(vid), float(b.dmin[v]), float(b.dmax[v]), po.default_opt(seed=SEED, viewID=v, nSubResolutionLevels=1),
                                                                                  geo_iter=geo, depth=self.depth[v], normal=self.normal[v])

    def local_depths(self, ids):
        return [torch.from_numpy(self.depth[v].copy()) for v in ids]

    def local_maps(self, ids, what):
        return [torch.from_numpy({"depth": self.depth, "conf": self.conf}[what][v].copy()) for v in ids]

    def set_snapshot_views(self, own_ids, own, foreign_ids, foreign):
        self.snap = [None] * 5
        for v, m in zip(own_ids, own):
            self.snap[v] = m.numpy()
        for v, m in zip(foreign_ids, foreign):
            assert tuple(m.shape) == self.shapes[v]
            self.snap[v] = m.numpy()


def _mixed_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    est = MixedSizeEstimator()
    nbs = [[int(i) for i in est.base.neighbors[v]] for v in range(5)]
    drv = ShardedDensifier(est, 5, world, rank, geo_iters=1, neighbors=nbs, view_shapes=est.shapes)
    drv.run()
    for v in drv.mine:
        np.save(os.path.join(out_dir, "depth%d.npy" % v), est.depth[v])
    # the fusing rank collects every view's map view by view (ShardedDensifier.gather(root=)); there is no [n, H, W] tensor of such a scene
    try:
        drv.gather("depth")
        raise AssertionError("gather without a root must refuse views of different sizes")
    except ValueError:
        pass
    root = world - 1
    got = drv.gather("depth", root=root)
    if rank == root:
        assert [tuple(m.shape) for m in got] == [tuple(s) for s in est.shapes]
        for v in range(5):
            np.save(os.path.join(out_dir, "gathered%d.npy" % v), got[v].numpy())
    else:
        assert got is None
    dist.destroy_process_group()


def test_neighbour_only_exchange_of_views_of_different_sizes(tmp_path):
    """Views whose maps differ in size travel one message per view: 3 gloo ranks equal one process (photometric pass + one geometric round reading the neighbours'
    maps at the neighbours' sizes)."""
    est = MixedSizeEstimator()
    nbs = [[int(i) for i in est.base.neighbors[v]] for v in range(5)]
    ShardedDensifier(est, 5, 1, 0, geo_iters=1, neighbors=nbs, view_shapes=est.shapes).run()
    assert len({d.shape for d in est.depth}) == 3
    for world in (3,):                                                    # blocks of 2, 2 and 1 views (two ranks pass too)
        out = tmp_path / ("w%d" % world); out.mkdir()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        mp.spawn(_mixed_worker, args=(world, port, str(out)), nprocs=world, join=True)
        for v in range(5):
            assert np.array_equal(np.load(out / ("depth%d.npy" % v)), est.depth[v]), "world %d view %d" % (world, v)
            assert np.array_equal(np.load(out / ("gathered%d.npy" % v)), est.depth[v]), "world %d view %d gathered on the last rank" % (world, v)
    assert (est.depth[3] > 0).mean() > 0.2


# ---- the real engine behind the driver: distributed.EngineRank on the CPU emulator, 2 and 3 gloo ranks --------------------------------------------------------
def _engine_scene():
    return synth.make_scene(6, 48, 36, n_src=3)        # (one sub-resolution level below: the emulator is ~1000x slower than the device)


def _engine_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    from openmvs_amd import patchmatch
    from openmvs_amd.distributed import EngineRank, gather_views_to_root
    from tests import emu
    with emu.emulated(patchmatch, "PMHIP_LIB", "libpmhip_emu.so"):
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        sc = _engine_scene()
        nbs = [[int(x) for x in sc.neighbors[v]] for v in range(sc.n_views)]
        asked = []
        eng = patchmatch.PatchMatchHIP(0)
        p = patchmatch.default_params(seed=SEED, nEstimationGeometricIters=2, nSubResolutionLevels=1)
        est = EngineRank(eng, p, sc.n_views, world, rank, nbs, lambda g: (asked.append(g), sc.gray[g])[1], sc.K, sc.R, sc.C, sc.dmin, sc.dmax, sc.width, sc.height, n_levels=1)
        assert sorted(asked) == sorted(est.held) and len(est.held) <= sc.n_views            # a rank reads only the images it holds
        drv = ShardedDensifier(est, sc.n_views, world, rank, geo_iters=2, neighbors=nbs)
        drv.run()
        final = [m.clone() for m in est.local_depths(drv.mine)]
        drv.filter()
        fd = gather_views_to_root(est.local_maps(drv.mine, "depth").clone(), sc.n_views, world, rank, 0)
        fc = gather_views_to_root(est.local_maps(drv.mine, "conf").clone(), sc.n_views, world, rank, 0)
        est_final = gather_views_to_root(torch.stack(final) if final else torch.zeros((0, sc.height, sc.width)), sc.n_views, world, rank, 0)
        if rank == 0:
            np.save(os.path.join(out_dir, "final.npy"), torch.stack(est_final).numpy())
            np.save(os.path.join(out_dir, "filtered_depth.npy"), torch.stack(fd).numpy()); np.save(os.path.join(out_dir, "filtered_conf.npy"), torch.stack(fc).numpy())
        eng.close()
        if world > 1:
            dist.destroy_process_group()


def test_engine_ranks_match_single_process(tmp_path):
    """distributed.EngineRank -- the compact per-rank scene, slots numbered locally, random numbers by the view's index in the whole scene, maps handed over by pointer --
    with the product's kernels and host engine (under the wave64 emulator) behind ShardedDensifier: 2 and 3 gloo ranks give the bits of one process, through the
    photometric pass, two geometric rounds with the neighbour-only exchange, and the cross-view filter."""
    one = tmp_path / "w1"; one.mkdir()
    _engine_worker(0, 1, 0, str(one))
    want = [np.load(one / n) for n in ("final.npy", "filtered_depth.npy", "filtered_conf.npy")]
    assert (want[0] > 0).mean() > 0.3 and (want[1] != want[0]).any()
    for world in (2, 3):
        out = tmp_path / ("w%d" % world); out.mkdir()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        mp.spawn(_engine_worker, args=(world, port, str(out)), nprocs=world, join=True)
        for n, w in zip(("final.npy", "filtered_depth.npy", "filtered_conf.npy"), want):
            assert np.array_equal(np.load(out / n), w), "world %d %s" % (world, n)


# ---- the whole PatchMatch path of Scene::DenseReconstruction over ranks (distributed.dense_reconstruction) ------------------------------------------------------
_REAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "scene", "scene.mvs")


def _recon_opt():
    from openmvs_amd import optdense
    opt = optdense.defaults()
    opt.nResolutionLevel = 3; opt.nMinResolution = 40; opt.nNumViews = 8; opt.nEstimateNormals = 2; opt.nSpeckleSize = 20; opt.nEstimationGeometricIters = 1
    return opt


def _recon_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    from openmvs_amd import densify, patchmatch
    from openmvs_amd import distributed as D
    from tests import emu
    with emu.emulated(patchmatch, "PMHIP_LIB", "libpmhip_emu.so"):
        dist.init_process_group("gloo", rank=rank, world_size=world)
        opt = _recon_opt()
        sv = densify.load_scene(_REAL, opt=opt)
        eng = patchmatch.PatchMatchHIP(0)
        cloud = D.dense_reconstruction(eng, sv, opt, world, rank, seed=3)
        assert (cloud is None) == (rank != 0)
        if rank == 0:
            np.savez(os.path.join(out_dir, "cloud.npz"), **{k: v for k, v in cloud.items() if isinstance(v, np.ndarray)})
        eng.close()
        dist.destroy_process_group()


def test_dense_reconstruction_over_ranks_matches_one_engine(tmp_path):
    """distributed.dense_reconstruction on the reference's pipeline-test scene (80x60), 2 and 3 gloo ranks of the emulated engine: the fused cloud -- points, views, weights,
    normals, colours -- equals densify.dense_reconstruction on one engine."""
    from openmvs_amd import densify, patchmatch
    from tests import emu
    with emu.emulated(patchmatch, "PMHIP_LIB", "libpmhip_emu.so"):
        eng = patchmatch.PatchMatchHIP(0)
        sv, want = densify.dense_reconstruction(eng, _REAL, None, _recon_opt(), seed=3)
        eng.close()
    assert want["nPoints"] > 100
    for world in (2, 3):
        out = tmp_path / ("w%d" % world); out.mkdir()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        mp.spawn(_recon_worker, args=(world, port, str(out)), nprocs=world, join=True)
        got = np.load(out / "cloud.npz")
        for k in ("points", "viewStart", "views", "weights", "projs", "colors", "normals"):
            assert np.array_equal(got[k], want[k]), (world, k)


# ---- the same over a scene whose images differ in size and read resampled copies of their neighbours (ViewData::ScaleImage) --------------------------------
def _mixed_loader(p):
    """The pipeline-test scene with its third image cut to 3/4 of the size of the others: the footprints then differ by the resolution ratio and the reference's rule
    (SceneDensify.cpp:306-345) makes the other images read it enlarged and it read them reduced."""
    from PIL import Image
    with Image.open(p) as im:
        rgb = np.asarray(im.convert("RGB"))
    names = sorted(os.listdir(os.path.dirname(p)))
    if os.path.basename(p) == [n for n in names if n.lower().endswith((".jpg", ".png"))][2]:
        h, w = rgb.shape[:2]
        return np.ascontiguousarray(rgb[:h * 3 // 4, :w * 3 // 4])
    return rgb


def _sized_recon_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    from openmvs_amd import densify, patchmatch
    from openmvs_amd import distributed as D
    from tests import emu
    with emu.emulated(patchmatch, "PMHIP_LIB", "libpmhip_emu.so"):
        dist.init_process_group("gloo", rank=rank, world_size=world)
        opt = _recon_opt()
        sv = densify.load_scene(_REAL, opt=opt, image_loader=_mixed_loader)
        eng = patchmatch.PatchMatchHIP(0)
        cloud = D.dense_reconstruction(eng, sv, opt, world, rank, seed=3)
        assert (cloud is None) == (rank != 0)
        if rank == 0:
            np.savez(os.path.join(out_dir, "cloud.npz"), **{k: v for k, v in cloud.items() if isinstance(v, np.ndarray)})
        eng.close()
        dist.destroy_process_group()


def test_dense_reconstruction_over_ranks_with_sizes_and_copies(tmp_path):
    """distributed.dense_reconstruction on a scene with images of two sizes, whose views read resampled copies of their neighbours: 2 gloo ranks of the emulated engine --
    maps exchanged view by view, a rank's copies fed the depth map of the image they stand for at every round boundary, image neighbours back for filter and fusion --
    give the cloud of densify.dense_reconstruction on one engine."""
    from openmvs_amd import densify, patchmatch
    from tests import emu
    with emu.emulated(patchmatch, "PMHIP_LIB", "libpmhip_emu.so"):
        eng = patchmatch.PatchMatchHIP(0)
        sv, want = densify.dense_reconstruction(eng, _REAL, None, _recon_opt(), seed=3, image_loader=_mixed_loader)
        eng.close()
    assert sv.alias_of and len(set(sv.sizes[:4])) == 2 and want["nPoints"] > 50
    out = tmp_path / "w2"; out.mkdir()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(_sized_recon_worker, args=(2, port, str(out)), nprocs=2, join=True)
    got = np.load(out / "cloud.npz")
    for k in ("points", "viewStart", "views", "weights", "projs", "colors", "normals"):
        assert np.array_equal(got[k], want[k]), k

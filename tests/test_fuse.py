"""CPU tests of FuseDepthMaps: the sequential oracle's properties, and the device algorithm (deterministic reservations,
openmvs_amd/csrc/pm_fuse.h) run through a host emulation of the GPU scheduler against that oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import fuse_cases as fc

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul():
    src = os.path.join(HERE, "cpp", "fuse_emul.cpp")
    out = os.path.join(HERE, "cpp", "build", "libfuse_emul.so")
    hdr = os.path.join(HERE, "..", "openmvs_amd", "csrc", "pm_fuse.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(out):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", out, src])
    lib = C.CDLL(out)
    lib.emu_fuse_depth_maps.restype = C.c_int
    return lib


def _fuse(sc, maps, fn=None, **kw):
    d, n, c = maps
    return po.fuse_depth_maps(d, n, c, list(sc.bgr), sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], fn=fn, **kw)


def test_oracle_fuse_properties(small_scene):
    sc = small_scene
    maps = fc.make_maps(sc, seed=1)
    r = _fuse(sc, maps)
    nv = np.diff(r["viewStart"])
    assert r["nPoints"] > 2000 and nv.min() >= 2 and nv.max() <= 1 + sc.neighbors.shape[1]
    assert r["nDepths"] <= sum(int((d > 0).sum()) for d in maps[0])
    # every pixel belongs to at most one point, and views inside a point are strictly ascending
    key = r["views"].astype(np.int64) * (1 << 32) + r["projs"][:, 1].astype(np.int64) * 65536 + r["projs"][:, 0]
    assert len(np.unique(key)) == len(key)
    for s, e in zip(r["viewStart"][:200], r["viewStart"][1:201]):
        assert np.all(np.diff(r["views"][s:e].astype(np.int64)) > 0)
    # fused points lie on the ground-truth surface: re-project into the first view of each point
    v0 = r["views"][r["viewStart"][:-1]]; xy = r["projs"][r["viewStart"][:-1]]
    X = r["points"].astype(np.float64)
    z = np.einsum("ij,ij->i", X - sc.C[v0], sc.R[v0][:, 2])
    gt = sc.gt_depth[v0, xy[:, 1], xy[:, 0]]
    assert np.median(np.abs(z - gt) / gt) < 2e-3
    assert np.allclose(np.linalg.norm(r["normals"], axis=1), 1, atol=1e-5)
    # nMinViewsFuse = 3 keeps a subset; merging more views never invents points
    r3 = _fuse(sc, maps, nMinViewsFuse=3)
    assert 0 < r3["nPoints"] < r["nPoints"] and np.diff(r3["viewStart"]).min() >= 3
    # no colours / normals requested -> absent
    r0 = _fuse(sc, maps, bEstimateColor=False, bEstimateNormal=False)
    assert r0["colors"] is None and r0["normals"] is None and np.array_equal(r0["points"], r["points"])


def test_oracle_fuse_known_answer():
    """Two fronto-parallel views of the plane z = 4, one pixel each: the point is the weighted mean of the two back-projections."""
    K = np.array([[100.0, 0, 2], [0, 100.0, 2], [0, 0, 1]]); R = np.eye(3)
    Cs = [np.zeros(3), np.array([0.04, 0, 0])]            # 1 pixel of disparity at z = 4
    d = [np.zeros((5, 5), np.float32) for _ in range(2)]
    d[0][2, 3] = 4.0; d[1][2, 2] = 4.0
    n = [np.tile(np.float32([0, 0, -1]), (5, 5, 1)) for _ in range(2)]
    c = [np.full((5, 5), 0.5, np.float32), np.full((5, 5), 0.9, np.float32)]
    bgr = [np.full((5, 5, 3), 100, np.uint8), np.full((5, 5, 3), 200, np.uint8)]
    r = po.fuse_depth_maps(d, n, c, bgr, [K, K], [R, R], Cs, [[1], [0]])
    assert r["nPoints"] == 1 and r["nDepths"] == 2 and list(r["views"]) == [0, 1]
    assert [tuple(p) for p in r["projs"]] == [(3, 2), (2, 2)]
    w0 = np.float32(1) / (np.float32(0.5) * np.float32(4) * np.float32(4)); w1 = np.float32(1) / (np.float32(0.1) * np.float32(16))
    assert np.allclose(r["weights"], [w0, w1], rtol=1e-6)
    X0 = np.array([0.04, 0, 4.0]); X1 = np.array([0.04, 0, 4.0])
    assert np.allclose(r["points"][0], (X0 * w0 + X1 * w1) / (w0 + w1), atol=1e-6)
    assert np.allclose(r["colors"][0], np.rint((100 * w0 + 200 * w1) / (w0 + w1)), atol=1)
    assert np.allclose(r["normals"][0], [0, 0, -1], atol=1e-6)
    # an occluder in view 1 (closer than the projected point) is only zeroed if the point survives; here it blocks the claim instead
    d2 = [d[0].copy(), d[1].copy()]; d2[1][2, 2] = 6.0          # view-0 point at z=4 is in front of view 1's z=6 estimate -> invalidates it
    r2 = po.fuse_depth_maps(d2, n, c, bgr, [K, K], [R, R], Cs, [[1], [0]])
    assert r2["nPoints"] == 0 and r2["nDepths"] == 2                # single-view seeds are rolled back, nothing is zeroed


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_device_algorithm_emulation_matches_the_sequential_oracle(emul, small_scene, nine_scene, mode, monkeypatch):
    monkeypatch.setenv("EMU_FUSE_ORDER", mode)          # 0: ascending threads, 1: descending, 2: random per phase
    for sc, seed, kw in ((small_scene, 1, {}), (small_scene, 2, dict(nMinViewsFuse=3)), (nine_scene, 3, {}),
                         (nine_scene, 4, dict(fDepthDiffThreshold=0.03, fNormalDiffThreshold=60.0, bEstimateColor=False))):
        maps = fc.make_maps(sc, seed=seed)
        ref = _fuse(sc, maps, **kw)
        got = _fuse(sc, maps, fn=emul.emu_fuse_depth_maps, **kw)
        fc.same_cloud(got, ref, f"seed {seed} mode {mode}")
    rounds, seeds = C.c_uint64(), C.c_uint64()
    emul.emu_fuse_stats(C.byref(rounds), C.byref(seeds))
    assert rounds.value < 40 * 9          # a handful of rounds per image, not one per seed


def test_emulation_with_missing_depth_maps_and_custom_order(emul, small_scene):
    sc = small_scene
    d, n, c = fc.make_maps(sc, seed=5)
    d[3] = None
    order = [4, 0, 2, 1]
    a = po.fuse_depth_maps(d, n, c, list(sc.bgr), sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], order=order)
    b = po.fuse_depth_maps(d, n, c, list(sc.bgr), sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], order=order, fn=emul.emu_fuse_depth_maps)
    fc.same_cloud(b, a, "missing map")
    assert 3 not in set(a["views"]) and a["nPoints"] > 0


def test_merge_depth_maps_mode(emul, small_scene):
    """nMinViewsFuse < 2 = DepthMapsData::MergeDepthMaps: every valid depth is a single-view point, images in index order."""
    sc = small_scene
    maps = fc.make_maps(sc, seed=6)
    order = list(range(sc.n_views))
    a = _fuse(sc, maps, nMinViewsFuse=1, order=order)
    assert a["nPoints"] == a["nDepths"] == sum(int((d > 0).sum()) for d in maps[0])
    assert np.all(np.diff(a["viewStart"]) == 1) and np.all(np.diff(a["views"].astype(np.int64)) >= 0) and not a["weights"].any()
    v, xy = a["views"], a["projs"]
    assert np.array_equal(a["colors"], sc.bgr[v, xy[:, 1], xy[:, 0]])
    z = np.einsum("ij,ij->i", a["points"].astype(np.float64) - sc.C[v], sc.R[v][:, 2])
    assert np.allclose(z, np.stack(maps[0])[v, xy[:, 1], xy[:, 0]], rtol=1e-5)
    b = _fuse(sc, maps, nMinViewsFuse=1, order=order, fn=emul.emu_fuse_depth_maps)
    fc.same_cloud(b, a, "merge")


def _golden_inputs():
    g = np.load(os.path.join(HERE, "golden", "pm_golden_96x64.npz")); f = np.load(os.path.join(HERE, "golden", "filter_golden_96x64.npz"))
    z = np.load(os.path.join(HERE, "golden", "fuse_golden_96x64.npz"))
    nv = int(g["n_views"])
    return g, z, list(f["filt_depth"]), list(f["gap_normal"]), list(f["filt_conf"]), list(z["bgr"]), [list(g["neighbors"][v]) for v in range(nv)]


def test_oracle_reproduces_the_committed_golden_cloud():
    """Regression pin of the fusion oracle (tests/golden/make_fuse_golden.py wrote the fixture from it)."""
    g, z, d, n, c, bgr, nbrs = _golden_inputs()
    for tag, kw in (("fuse2", dict(nMinViewsFuse=2)), ("fuse3", dict(nMinViewsFuse=3, fNormalDiffThreshold=40.0))):
        r = po.fuse_depth_maps(d, n, c, bgr, g["K"], g["R"], g["C"], nbrs, **kw)
        assert r["nDepths"] == int(z[tag + "_nDepths"]) and r["nPoints"] == len(z[tag + "_points"]) > 500
        for k in ("points", "viewStart", "views", "weights", "projs", "colors", "normals"):
            assert np.array_equal(r[k], z[tag + "_" + k]), (tag, k)

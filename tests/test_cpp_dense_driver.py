"""include/DenseDepthMapsHIP.hpp: the C++ host driver of the HBM-resident scene interface (Scene::ComputeDepthMaps + DenseReconstruction shape,
SceneDensify.cpp:1754-1982, :1655-1750) compiles against nothing but this repo's headers and, on a GPU, reproduces the chain of oracle stages --
estimation rounds, speckle / gap / cross-view filters, .dmap files, fusion -- without Python in the loop."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def _build(tmp):
    from openmvs_amd import build
    lib = build.build_lib("libpmhip.so")
    dm = build.build_host_lib("libdmapio.so")
    exe = os.path.join(tmp, "dense_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "dense_driver.cpp"),
                           "-o", exe, lib, dm, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def test_dense_driver_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path))
    assert subprocess.run([exe]).returncode == 2          # usage error path, no GPU touched


@pytest.mark.gpu
def test_dense_driver_matches_the_oracle_pipeline(tmp_path, small_scene):
    from openmvs_amd import dmap
    from oracle import pyoracle as po
    from tests import fuse_cases as fcs
    sc = small_scene
    seed, speckle = 31, 30
    exe = _build(str(tmp_path))
    inp = tmp_path / "scene.bin"; out = tmp_path / "out.bin"; ddir = tmp_path / "dmaps"; ddir.mkdir()
    n, w, h, ns = sc.n_views, sc.width, sc.height, sc.neighbors.shape[1]
    with open(inp, "wb") as f:
        f.write(np.array([n, w, h, ns], np.int32).tobytes())
        for i in range(n):
            f.write(np.ascontiguousarray(sc.gray[i], np.float32).tobytes()); f.write(np.ascontiguousarray(sc.bgr[i], np.uint8).tobytes())
            f.write(np.concatenate([sc.K[i].ravel(), sc.R[i].ravel(), sc.C[i].ravel()]).astype(np.float64).tobytes())
            f.write(np.array([sc.dmin[i], sc.dmax[i]], np.float32).tobytes()); f.write(np.ascontiguousarray(sc.neighbors[i], np.int32).tobytes())
    subprocess.check_call([exe, str(inp), str(out), str(ddir), str(seed), str(speckle)])
    raw = np.fromfile(out, np.uint8)
    P = w * h
    maps = np.frombuffer(raw[:n * P * 5 * 4].tobytes(), np.float32).reshape(n, 5 * P)
    got = {v: (maps[v][:P].reshape(h, w), maps[v][P:4 * P].reshape(h, w, 3), maps[v][4 * P:].reshape(h, w)) for v in range(n)}
    # oracle chain (the one of test_whole_dense_schedule_matches_the_oracle_pipeline)
    allv = list(range(n))

    def orc(v, geo_iter=-1, depth=None, normal=None, src=None):
        ids = [v] + list(sc.neighbors[v])
        views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps=src)
        return po.estimate_depth_map(views, len(ids), float(sc.dmin[v]), float(sc.dmax[v]), po.default_opt(seed=seed, viewID=v), geo_iter=geo_iter, depth=depth, normal=normal)
    cur = {v: orc(v) for v in allv}
    for geo in range(2):
        prev = {v: cur[v][0] for v in allv}
        cur = {v: orc(v, geo, cur[v][0], cur[v][1], prev) for v in allv}
    cur = {v: po.gap_interpolation(*po.remove_small_segments(*cur[v], nSpeckleSize=speckle)) for v in allv}
    dep = np.stack([cur[v][0] for v in allv]); cnf = np.stack([cur[v][2] for v in allv])
    final = {}
    for v in allv:
        rc, fd, fc = po.filter_depth_map(dep, cnf, sc.K, sc.R, sc.C, v, list(sc.neighbors[v]), sc.dmin[v], sc.dmax[v])
        assert rc == 0
        final[v] = (fd, cur[v][1], fc)
        for a, b, what in zip(got[v], final[v], ("depth", "normal", "conf")):
            assert np.array_equal(a, b), "view %d %s: %d values differ" % (v, what, int((a != b).sum()))
        f = dmap.load(str(ddir / dmap.depth_file_name(v)))
        assert np.array_equal(f["depth_map"], fd) and np.array_equal(f["confidence_map"], fc) and f["neighbor_view_ids"] == [int(i) for i in sc.neighbors[v]]
    # fused cloud: same points in the same order as the sequential oracle fusion over the final maps
    order = po.fuse_order([len(sc.neighbors[v]) for v in allv])
    ref = po.fuse_depth_maps([final[v][0] for v in allv], [final[v][1] for v in allv], [final[v][2] for v in allv], [sc.bgr[v] for v in allv],
                             sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], order=order)
    o = n * P * 5 * 4
    nP, nV = (int(x) for x in np.frombuffer(raw[o:o + 16].tobytes(), np.uint64)); o += 16
    pts = np.frombuffer(raw[o:o + nP * 12].tobytes(), np.float32).reshape(nP, 3); o += nP * 12
    vs = np.frombuffer(raw[o:o + (nP + 1) * 4].tobytes(), np.uint32); o += (nP + 1) * 4
    views = np.frombuffer(raw[o:o + nV * 4].tobytes(), np.uint32)
    assert nP == ref["nPoints"] and nP > 1000
    assert np.array_equal(pts, ref["points"]) and np.array_equal(vs, ref["viewStart"]) and np.array_equal(views, ref["views"])

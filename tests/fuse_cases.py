"""Inputs for the FuseDepthMaps tests: depth / normal / confidence maps derived from a synthetic scene's ground truth, perturbed so
that every branch of the fusion runs: agreeing depths (claims), slightly-off depths, far outliers in front of and behind the surface
(occlusion -> invalidation), holes, tilted normals (normal test fails), low and high confidences."""
import numpy as np


def normals_from_depth(depth, K):
    """Camera-space unit normals facing the camera from a depth map (finite differences of the back-projected points)."""
    h, w = depth.shape
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    X = np.stack([(xs - K[0, 2]) * depth / K[0, 0], (ys - K[1, 2]) * depth / K[1, 1], depth.astype(np.float64)], -1)
    dx = np.gradient(X, axis=1); dy = np.gradient(X, axis=0)
    n = np.cross(dx, dy)
    n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)
    n[(n * X).sum(-1) > 0] *= -1
    return n.astype(np.float32)


def make_maps(sc, seed=0, hole=0.15, outlier=0.06, tilt=0.05, noise=2e-3):
    r = np.random.RandomState(seed)
    deps, nrms, cnfs = [], [], []
    for v in range(sc.n_views):
        gt = sc.gt_depth[v]
        d = gt * (1 + noise * r.randn(*gt.shape)).astype(np.float32)
        o = r.rand(*gt.shape) < outlier
        d[o] = gt[o] * r.choice([0.7, 0.9, 1.1, 1.4], size=int(o.sum())).astype(np.float32)
        n = normals_from_depth(gt, sc.K[v])
        t = r.rand(*gt.shape) < tilt
        nt = n[t] + 0.9 * r.randn(int(t.sum()), 3).astype(np.float32)
        n[t] = nt / np.linalg.norm(nt, axis=1, keepdims=True)
        d[r.rand(*gt.shape) < hole] = 0
        d[~np.isfinite(d) | (gt <= 0)] = 0
        deps.append(d.astype(np.float32)); nrms.append(n.astype(np.float32))
        cnfs.append((0.02 + 0.97 * r.rand(*gt.shape)).astype(np.float32))
    return deps, nrms, cnfs


def same_cloud(a, b, what=""):
    assert a["nPoints"] == b["nPoints"] and a["nDepths"] == b["nDepths"], f"{what}: {a['nPoints']}/{a['nDepths']} vs {b['nPoints']}/{b['nDepths']}"
    for k in ("viewStart", "views", "projs"):
        assert np.array_equal(a[k], b[k]), f"{what}: {k} differs"
    for k in ("points", "weights", "normals"):
        if a[k] is None or b[k] is None:
            assert a[k] is None and b[k] is None, f"{what}: {k} presence"
            continue
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), f"{what}: {k} differs in {(a[k].view(np.uint32) != b[k].view(np.uint32)).sum()} values"
    if a["colors"] is None or b["colors"] is None:
        assert a["colors"] is None and b["colors"] is None
    else:
        assert np.array_equal(a["colors"], b["colors"]), f"{what}: colors differ"


def make_mixed(seed=0, W=160, H=120, n_views=5, n_src=4):
    """The same cameras and surface rendered at three sizes (1x, 0.8x, 1.25x), every view taking its maps, image and K from one of them: a scene whose depth maps have
    different sizes, as DepthMapsData::InitViews leaves them when the images do (SceneDensify.cpp:306-459).  -> deps, nrms, cnfs, bgrs, K, R, C, neighbour lists."""
    from openmvs_amd import synth
    scs = [synth.make_scene(n_views, W, H, n_src=n_src), synth.make_scene(n_views, W * 4 // 5, H * 4 // 5, n_src=n_src), synth.make_scene(n_views, W * 5 // 4, H * 5 // 4, n_src=n_src)]
    maps = [make_maps(s, seed + k) for k, s in enumerate(scs)]
    pick = [(v * 2 + seed) % 3 if v else 0 for v in range(n_views)]
    deps = [maps[pick[v]][0][v] for v in range(n_views)]; nrms = [maps[pick[v]][1][v] for v in range(n_views)]; cnfs = [maps[pick[v]][2][v] for v in range(n_views)]
    bgrs = [scs[pick[v]].bgr[v] for v in range(n_views)]; K = [scs[pick[v]].K[v] for v in range(n_views)]
    base = scs[0]
    assert len({d.shape for d in deps}) == 3
    return deps, nrms, cnfs, bgrs, K, base.R, base.C, [list(base.neighbors[v]) for v in range(n_views)]

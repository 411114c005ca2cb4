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

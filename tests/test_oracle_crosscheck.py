"""An independent second restatement (numpy, written from the reference source on its own) of the heart of the estimator -- FillPixelPatch,
ComputeHomographyMatrix, ScorePixelImage's photometric term and the MINMEAN aggregation (libs/MVS/DepthMap.cpp:422-462,470-520,567-610;
DepthMap.h:403-423 in /root/reference) -- compared with the C++ oracle's orc_score_pixel.  Two readings of the same text must agree; the only
tolerated difference is the last bits of exp (numpy's vs the Cephes kernel of pm_math.h)."""
import numpy as np
import pytest

from oracle import pyoracle as po

f32 = np.float32
HW, STEP = 4, 2
TH_ROBUST = f32(f32(f32(0.9) * f32(4)) / f32(3))      # thRobust = fNCCThresholdKeep * 4.f / 3.f, DepthMap.cpp:406 (default 1.2)


def _weights(img, x, y):
    """FillPixelPatch (weighted variant) + GetWeight."""
    center = img[y, x]
    sigmaColor = f32(-1) / (f32(2) * f32(0.1) * f32(0.1)); sigmaSpatial = f32(-1) / (f32(2) * f32((HW - 1) ** 2))
    w, tw = [], []
    acc = f32(0); sumW = f32(0)
    for i in range(-HW, HW + 1, STEP):
        for j in range(-HW, HW + 1, STEP):
            I = img[y + i, x + j]
            dc = I - center
            wt = f32(np.exp(f32(dc * dc) * sigmaColor + f32(j * j + i * i) * sigmaSpatial))
            acc = f32(acc + f32(I * wt)); sumW = f32(sumW + wt)
            w.append(wt); tw.append(I)
    tm = f32(acc / sumW)
    normSq0 = f32(0)
    for n in range(len(w)):
        t = f32(tw[n] - tm)
        tw[n] = f32(w[n] * t)
        normSq0 = f32(normSq0 + f32(tw[n] * t))
    return w, tw, sumW, normSq0


def _sample(img, px, py):
    lx, ly = int(px), int(py)
    x = f32(px - f32(lx)); x1 = f32(f32(1) - x); y = f32(py - f32(ly)); y1 = f32(f32(1) - y)
    return f32(f32(f32(img[ly, lx] * x1) + f32(img[ly, lx + 1] * x)) * y1 + f32(f32(img[ly + 1, lx] * x1) + f32(img[ly + 1, lx + 1] * x)) * y)


def score_view(img0, K0, R0, C0, img1, K1, R1, C1, x, y, depth, normal, thRobust=TH_ROBUST):
    w, tw, sumW, normSq0 = _weights(img0, x, y)
    Hl = K1 @ R1 @ R0.T; Hm = K1 @ R1 @ (C0 - C1); Hr = np.linalg.inv(K0)
    X0 = np.array([(x - K0[0, 2]) / K0[0, 0], (y - K0[1, 2]) / K0[1, 1], 1.0])
    n = normal.astype(np.float64)
    H = ((Hl + np.outer(Hm, n * (1.0 / (n.dot(X0) * float(depth))))) @ Hr).astype(f32)
    X = np.array([f32(f32(H[r, 0] * f32(x - HW)) + f32(H[r, 1] * f32(y - HW))) + H[r, 2] for r in range(3)], f32)
    base = X.copy()
    H2 = (H * f32(STEP)).astype(f32)
    s = f32(0); sq = f32(0); num = f32(0); k = 0
    h1, w1 = img1.shape
    for i in range(-HW, HW + 1, STEP):
        for j in range(-HW, HW + 1, STEP):
            px, py = f32(X[0] / X[2]), f32(X[1] / X[2])
            if not (px >= 1 and py >= 1 and px <= w1 - 2 and py <= h1 - 2):
                return thRobust
            v = _sample(img1, px, py)
            vw = f32(v * w[k])
            s = f32(s + vw); sq = f32(sq + f32(v * vw)); num = f32(num + f32(v * tw[k])); k += 1
            X = (X + H2[:, 0]).astype(f32)
        base = (base + H2[:, 1]).astype(f32)
        X = base.copy()
    normSq1 = f32(sq - f32(f32(s * s) / sumW))
    nrm = f32(normSq0 * normSq1)
    if nrm <= f32(1e-16):
        return thRobust
    ncc = min(max(f32(num / f32(np.sqrt(nrm))), f32(-1)), f32(1))
    return min(f32(2), f32(f32(1) - ncc))


def minmean(scores, thRobust=TH_ROBUST):
    s = sorted(scores)
    if len(s) == 1:                          # idxScore == 0
        return s[0]
    return s[0] if s[1] >= thRobust else f32(f32(s[0] + s[1]) / f32(2))


def test_score_pixel_two_readings_agree(small_scene):
    sc = small_scene
    r = np.random.RandomState(0)
    ref = 1
    ids = [ref] + list(sc.neighbors[ref])
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
    opt = po.default_opt(seed=1, viewID=ref)
    checked = offimg = 0
    for _ in range(60):
        x = int(r.randint(HW + 2, sc.width - HW - 2)); y = int(r.randint(HW + 2, sc.height - HW - 2))
        gt = float(sc.gt_depth[ref][y, x])
        depth = f32(gt * (1 + 0.02 * r.randn()) if r.rand() < 0.7 else r.uniform(sc.dmin[ref], sc.dmax[ref]))
        view_ray = np.array([(x - sc.K[ref][0, 2]) / sc.K[ref][0, 0], (y - sc.K[ref][1, 2]) / sc.K[ref][1, 1], 1.0])
        nrm = r.randn(3); nrm /= np.linalg.norm(nrm)
        if nrm.dot(view_ray) > 0:
            nrm = -nrm
        nrm = nrm.astype(f32)
        rc, got, agg = po.score_pixel(views, len(ids), opt, x, y, float(depth), nrm)
        if rc != 0:
            continue                         # low-texture pixel: the oracle refuses it (FillPixelPatch returns false)
        mine = [score_view(sc.gray[ref], sc.K[ref], sc.R[ref], sc.C[ref], sc.gray[v], sc.K[v], sc.R[v], sc.C[v], x, y, depth, nrm) for v in ids[1:]]
        assert np.allclose(got, np.array(mine, f32), rtol=0, atol=3e-5), (x, y, depth, got, mine)
        assert abs(agg - minmean(mine)) < 3e-5
        checked += 1; offimg += sum(1 for m in mine if m == TH_ROBUST)
    assert checked > 40 and offimg > 0        # both branches seen: in-image scores and the out-of-image constant thRobust


def test_thRobust_is_the_reference_value():
    """thRobust = fNCCThresholdKeep * 4/3 (DepthMap.cpp:406) with the default fNCCThresholdKeep 0.9 -> 1.2."""
    o = po.default_opt()
    assert abs(o.fNCCThresholdKeep - 0.9) < 1e-7

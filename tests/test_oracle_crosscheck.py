"""An independent second restatement (numpy, written from the reference source on its own) of the heart of the estimator -- FillPixelPatch,
ComputeHomographyMatrix, ScorePixelImage's photometric term and the MINMEAN aggregation (libs/MVS/DepthMap.cpp:422-462,470-520,567-610;
DepthMap.h:403-423 in /root/reference) -- compared with the C++ oracle's orc_score_pixel.  Two readings of the same text must agree; the only
tolerated difference is the last bits of exp (numpy's vs the Cephes kernel of pm_math.h)."""
import numpy as np

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


def _inv_k(K):
    o = np.eye(3); o[0, 0] = 1 / K[0, 0]; o[1, 1] = 1 / K[1, 1]; o[0, 2] = -K[0, 2] * o[0, 0]; o[1, 2] = -K[1, 2] * o[1, 1]
    return o


def _sample_if(img, px, py, ok):
    """TImage::sample(v, pt, functor), libs/Common/Types.inl:2296-2314: bilinear over the neighbours that pass the test, each failing one replaced."""
    lx, ly = int(px), int(py)
    x = f32(px - f32(lx)); x1 = f32(f32(1) - x); y = f32(py - f32(ly)); y1 = f32(f32(1) - y)
    v00, v10, v01, v11 = img[ly, lx], img[ly, lx + 1], img[ly + 1, lx], img[ly + 1, lx + 1]
    b00, b10, b01, b11 = ok(v00), ok(v10), ok(v01), ok(v11)
    if not (b00 or b10 or b01 or b11):
        return False, f32(0)
    a = v00 if b00 else (v10 if b10 else (v01 if b01 else v11)); b = v10 if b10 else (v00 if b00 else (v11 if b11 else v01))
    c = v01 if b01 else (v11 if b11 else (v00 if b00 else v10)); d = v11 if b11 else (v01 if b01 else (v10 if b10 else v00))
    return True, f32(f32(y1 * f32(f32(x1 * a) + f32(x * b))) + f32(y * f32(f32(x1 * c) + f32(x * d))))


def _sample(img, px, py):
    lx, ly = int(px), int(py)
    x = f32(px - f32(lx)); x1 = f32(f32(1) - x); y = f32(py - f32(ly)); y1 = f32(f32(1) - y)
    return f32(f32(f32(img[ly, lx] * x1) + f32(img[ly, lx + 1] * x)) * y1 + f32(f32(img[ly + 1, lx] * x1) + f32(img[ly + 1, lx + 1] * x)) * y)


def score_view(img0, K0, R0, C0, img1, K1, R1, C1, x, y, depth, normal, thRobust=TH_ROBUST, depth1_map=None, geo_weight=f32(0.1), prior=None):
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
    score = f32(f32(1) - ncc)
    if depth1_map is not None:           # geometric consistency with the source view's depth map, DepthMap.cpp:536-551
        KR1 = K1 @ R1
        Tl = (KR1 @ R0.T).astype(f32); Tm = (KR1 @ (C0 - C1)).astype(f32)
        Tr = (K0 @ R0 @ R1.T @ _inv_k(K1)).astype(f32); Tn = (K0 @ R0 @ (C1 - C0)).astype(f32)
        mv = lambda M, v: np.array([f32(f32(f32(M[r, 0] * v[0]) + f32(M[r, 1] * v[1])) + f32(M[r, 2] * v[2])) for r in range(3)], f32)
        p = np.array([f32(f32(X0[0]) * depth), f32(f32(X0[1]) * depth), depth], f32)
        X1 = (mv(Tl, p) + Tm).astype(f32)
        consistency = f32(4)
        if X1[2] > 0:
            x1 = (f32(X1[0] / X1[2]), f32(X1[1] / X1[2]))
            if x1[0] >= 1 and x1[1] >= 1 and x1[0] <= w1 - 2 and x1[1] <= h1 - 2:
                ok, d1 = _sample_if(depth1_map, x1[0], x1[1], lambda d: abs(f32(X1[2] - d)) / X1[2] < f32(0.03))
                if ok:
                    q = (mv(Tr, np.array([f32(x1[0] * d1), f32(x1[1] * d1), d1], f32)) + Tn).astype(f32)
                    xb = (f32(q[0] / q[2]), f32(q[1] / q[2]))
                    dx, dy = f32(f32(x) - xb[0]), f32(f32(y) - xb[1])
                    dist = f32(np.sqrt(f32(f32(dx * dx) + f32(dy * dy))))
                    consistency = min(f32(np.sqrt(f32(dist * f32(dist + f32(2))))), consistency)
        score = f32(score + f32(geo_weight * consistency))
    if prior is not None and prior[y, x] > 0:      # low-resolution depth prior on texture-less patches, :553-561
        d0 = prior[y, x]
        deltaDepth = min(f32(abs(f32(d0 - depth)) / d0), f32(0.5))
        factor = f32(np.exp(f32(normSq0 * f32(f32(-1) / f32(f32(1) * f32(0.02))))))
        score = f32(f32(f32(f32(1) - factor) * score) + f32(factor * deltaDepth))
    return min(f32(2), score)


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


def test_geometric_and_prior_terms_two_readings_agree(small_scene):
    sc = small_scene
    r = np.random.RandomState(1)
    ref = 2
    ids = [ref] + list(sc.neighbors[ref])
    noisy = {v: (sc.gt_depth[v] * (1 + 0.01 * r.randn(*sc.gt_depth[v].shape))).astype(f32) for v in range(sc.n_views)}
    for v in noisy:
        noisy[v][r.rand(*noisy[v].shape) < 0.1] = 0
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids, depth_maps=noisy)
    opt = po.default_opt(seed=1, viewID=ref)
    gw = f32(opt.fEstimationGeometricWeight)
    prior = (sc.gt_depth[ref] * f32(1.01)).astype(f32); prior[::3] = 0
    checked = 0
    for _ in range(50):
        x = int(r.randint(HW + 2, sc.width - HW - 2)); y = int(r.randint(HW + 2, sc.height - HW - 2))
        depth = f32(sc.gt_depth[ref][y, x] * (1 + 0.01 * r.randn()))
        view_ray = np.array([(x - sc.K[ref][0, 2]) / sc.K[ref][0, 0], (y - sc.K[ref][1, 2]) / sc.K[ref][1, 1], 1.0])
        nrm = -view_ray / np.linalg.norm(view_ray) + 0.2 * r.randn(3); nrm /= np.linalg.norm(nrm)
        nrm = nrm.astype(f32)
        rc, got, agg = po.score_pixel(views, len(ids), opt, x, y, float(depth), nrm, prior)
        if rc != 0:
            continue
        mine = [score_view(sc.gray[ref], sc.K[ref], sc.R[ref], sc.C[ref], sc.gray[v], sc.K[v], sc.R[v], sc.C[v], x, y, depth, nrm,
                           depth1_map=noisy[v], geo_weight=gw, prior=prior) for v in ids[1:]]
        assert np.allclose(got, np.array(mine, f32), rtol=0, atol=5e-5), (x, y, got, mine)
        checked += 1
    assert checked > 30


# ---- second readings of the small per-pixel helpers of ProcessPixel (DepthMap.cpp:915-971, DepthMap.h:447-453, Rotation.inl:696-728) ----------
def interpolate_pixel(K, x0, y0, nx, ny, depth, n, dmin, dmax):
    if x0 == nx:
        nx1 = f32((float(y0) - K[1, 2]) / K[1, 1]); denom = f32(n[2] + f32(nx1 * n[1]))
        if abs(denom) < f32(0.0001):
            return depth
        x1 = f32((float(ny) - K[1, 2]) / K[1, 1]); nom = f32(depth * f32(n[2] + f32(x1 * n[1])))
    else:
        nx1 = f32((float(x0) - K[0, 2]) / K[0, 0]); denom = f32(n[2] + f32(nx1 * n[0]))
        if abs(denom) < f32(0.0001):
            return depth
        x1 = f32((float(nx) - K[0, 2]) / K[0, 0]); nom = f32(depth * f32(n[2] + f32(x1 * n[0])))
    dn = f32(nom / denom)
    return dn if (dmin <= dn < dmax) else depth


def correct_normal(K, x0, y0, n):
    v = np.array([(x0 - K[0, 2]) / K[0, 0], (y0 - K[1, 2]) / K[1, 1], 1.0]).astype(f32)
    c = f32(n.dot(v))
    if c < 0:
        return n
    phi = min(f32(f32(f32(np.arccos(f32(c / f32(np.linalg.norm(v))))) - f32(np.deg2rad(90))) * f32(1.01)), f32(-0.001))
    axis = np.cross(n, v).astype(f32); axis = (axis * f32(f32(1) / f32(np.linalg.norm(axis)))).astype(f32)
    O = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]], f32)
    Rm = (np.eye(3, dtype=f32) + O * f32(np.sin(phi)) + (O @ O) * f32(f32(1) - f32(np.cos(phi)))).astype(f32)
    return (Rm @ n).astype(f32)


def smooth_factor(K, x0, y0, hyp_depth, hyp_n, nx, ny, ndepth, nn, bonus=f32(0.93), sdepth=f32(0.02), snormal_deg=f32(13)):
    X0 = np.array([(x0 - K[0, 2]) / K[0, 0], (y0 - K[1, 2]) / K[1, 1], 1.0]).astype(f32)
    D = f32(-hyp_depth * f32(hyp_n.dot(X0)))                                           # InitPlane
    Xn = np.array([(nx - K[0, 2]) * float(ndepth) / K[0, 0], (ny - K[1, 2]) * float(ndepth) / K[1, 1], float(ndepth)]).astype(f32)   # TransformPointI2C in double
    dist = f32(f32(hyp_n.dot(Xn)) + D)
    sd = f32(f32(-1) / f32(f32(2) * f32(sdepth * sdepth))); rn = f32(np.deg2rad(snormal_deg)); sn = f32(f32(-1) / f32(f32(2) * f32(rn * rn)))
    fd = f32(np.exp(f32(f32(dist / hyp_depth) ** 2) * sd))
    ca = min(max(f32(f32(hyp_n.dot(nn)) / f32(np.sqrt(f32(f32(hyp_n.dot(hyp_n)) * f32(nn.dot(nn)))))), f32(-1)), f32(1))
    fn = f32(np.exp(f32(f32(np.arccos(ca)) ** 2) * sn))
    bd = f32(f32(1) - bonus); bn = f32(bd * f32(0.96))
    return f32(f32(f32(1) - f32(bd * fd)) * f32(f32(1) - f32(bn * fn)))


def test_process_pixel_helpers_two_readings_agree(small_scene):
    sc = small_scene
    r = np.random.RandomState(2)
    ref = 0
    ids = [ref] + list(sc.neighbors[ref])
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
    opt = po.default_opt(seed=1, viewID=ref)
    K = sc.K[ref]; dmin, dmax = float(sc.dmin[ref]), float(sc.dmax[ref])
    rotated = 0
    for _ in range(200):
        x = int(r.randint(HW + 1, sc.width - HW - 1)); y = int(r.randint(HW + 1, sc.height - HW - 1))
        nx, ny = [(x - 1, y), (x + 1, y), (x, y - 1), (x, y + 1)][r.randint(4)]
        ndepth = f32(r.uniform(dmin, dmax))
        nn = r.randn(3); nn /= np.linalg.norm(nn); nn = nn.astype(f32)          # any unit vector: about half of them face away and get rotated
        hd = f32(r.uniform(dmin, dmax)); hn = r.randn(3); hn /= np.linalg.norm(hn); hn = hn.astype(f32)
        rc, di, cn, sf = po.pixel_helpers(views, len(ids), opt, x, y, dmin, dmax, nx, ny, float(ndepth), nn, float(hd), hn)
        assert rc == 0
        mine_d = interpolate_pixel(K, x, y, nx, ny, ndepth, nn, f32(dmin), f32(dmax))
        assert abs(di - mine_d) <= 2e-6 * max(1.0, abs(di)), (x, y, nx, ny, di, mine_d)
        mine_n = correct_normal(K, x, y, nn)
        rotated += int(not np.array_equal(mine_n, nn))
        assert np.abs(cn - mine_n).max() < 3e-6, (cn, mine_n)
        vd = np.array([(x - K[0, 2]) / K[0, 0], (y - K[1, 2]) / K[1, 1], 1.0])
        if rotated and not np.array_equal(mine_n, nn):
            assert mine_n.dot(vd) < 0                                             # after the correction the normal faces the camera
        mine_s = smooth_factor(K, x, y, hd, hn, nx, ny, ndepth, nn)
        assert abs(sf - mine_s) < 3e-6, (sf, mine_s)
    assert 40 < rotated < 160

"""Deterministic synthetic N-view scene generator.

A smooth height field textured with a multi-octave sinusoid albedo is rendered by analytic
ray / height-field intersection from pin-hole cameras on a grid above it.  The result is the
input contract of the depth-map engine: per view a gray float image in [0,1] (BGR u8 ->
gray exactly as the reference's Image::toGray, libs/Common/Types.inl:2377-2420, NormRGB_t
:1611-1615), K/R/C (x_cam = R (X - C), OpenMVS convention), a neighbour list, a depth range,
and the ground-truth depth (z in camera space) for accuracy reporting.

Runs on torch CPU (tests) or on the GPU (bench); the generator is a tool, not part of the
hot path, and its output is handed unchanged to both the HIP engine and the CPU oracle.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch

SEED = 20241016


@dataclass
class Scene:
    width: int
    height: int
    gray: np.ndarray        # [V,H,W] float32 in [0,1]
    bgr: np.ndarray         # [V,H,W,3] uint8
    K: np.ndarray           # [V,3,3] float64
    R: np.ndarray           # [V,3,3] float64
    C: np.ndarray           # [V,3]   float64
    gt_depth: np.ndarray    # [V,H,W] float32
    neighbors: np.ndarray   # [V,N] int32 (global view ids, nearest first)
    dmin: np.ndarray        # [V] float32
    dmax: np.ndarray        # [V] float32
    diameter: float         # diagonal of the AABB of the visible ground-truth surface
    meta: dict = field(default_factory=dict)

    @property
    def n_views(self) -> int:
        return self.gray.shape[0]


_PI_HI = 3.141592653589793          # fl(pi)
_PI_LO = 1.2246467991473532e-16     # pi - fl(pi)
_SIN_C = (-1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -1.0 / 39916800, 1.0 / 6227020800, -1.0 / 1307674368000, 1.0 / 355687428096000)


def exact_sin(x):
    """sin(x) from IEEE +,-,*,round only (no libm, no fused operations): the same bits on every host CPU and on the GPU, for Python floats
    and float64 torch tensors alike.  Accuracy ~1e-16 for the |x| < 1e5 the generator produces; what matters here is reproducibility, because
    the full-size golden files (tests/golden/) pin outputs for inputs that every machine has to regenerate."""
    if isinstance(x, torch.Tensor):
        k = torch.round(x * (1.0 / _PI_HI))
        odd = k - 2.0 * torch.floor(k * 0.5)
    else:
        k = float(round(x * (1.0 / _PI_HI)))
        odd = k - 2.0 * math.floor(k * 0.5)
    r = (x - k * _PI_HI) - k * _PI_LO
    r2 = r * r
    p = _SIN_C[-1]
    for c in _SIN_C[-2::-1]:
        p = p * r2 + c
    s = r + r * (r2 * p)
    return s * (1.0 - 2.0 * odd)


def exact_cos(x):
    return exact_sin(x + 0.5 * _PI_HI)


def _height(x, y, hp, sin=torch.sin):
    z = torch.zeros_like(x)
    for (f, g, p, q, a) in hp:
        z = z + a * sin(f * x + p) * sin(g * y + q)
    return z


def _albedo(x, y, ap, ch, sin=torch.sin):
    v = torch.full_like(x, 0.5)
    for (fx, fy, ph, amp, dph) in ap:
        v = v + amp * sin(fx * x + fy * y + ph + dph * ch)
    return v


def make_scene_torch(n_views: int, width: int, height: int, n_src: int = 8, seed: int = SEED,
                     device: str | torch.device = "cpu", spacing: float = 0.12, grid_cols: int | None = None,
                     want_bgr: bool = False, gt_views: int | None = None, exact: bool = False) -> dict:
    """Render `n_views` views of the seeded scene; images stay torch tensors on `device`.
    gt_views: keep ground-truth depth only for the first gt_views views (None = all).
    exact: build the scene from correctly rounded IEEE operations only (exact_sin instead of libm / the device's sin, explicit
    3-vector arithmetic instead of BLAS): identical images, cameras and ranges on any host and on the GPU (a different, equally valid scene
    than exact=False; used by the full-size golden files)."""
    _sin = exact_sin if exact else torch.sin
    _msin = exact_sin if exact else math.sin
    _mcos = exact_cos if exact else math.cos

    def _norm(v):
        return math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) if exact else np.linalg.norm(v)

    def _cross(a, b):
        return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]) if exact else np.cross(a, b)
    rng = np.random.RandomState(seed)
    dev = torch.device(device)
    S = 1.0
    A = 0.05 * S
    cam_h = 1.5 * S
    fpx = 1.2 * width
    footprint = cam_h / fpx  # world units per pixel at the mean depth
    # height field: 4 sinusoid products, wavelengths 0.5..2 S
    hp = []
    for _ in range(4):
        f = 2 * math.pi / rng.uniform(0.5, 2.0)
        g = 2 * math.pi / rng.uniform(0.5, 2.0)
        hp.append((f, g, rng.uniform(0, 2 * math.pi), rng.uniform(0, 2 * math.pi), A / 2))
    # albedo: octaves with wavelengths expressed in *pixels* so every pyramid level sees texture
    ap = []
    for lam_px in (3.3, 4.7, 6.1, 8.3, 11.0, 15.0, 21.0, 29.0, 41.0, 57.0, 83.0, 127.0, 191.0, 293.0):
        for _ in range(2):
            th = rng.uniform(0, math.pi)
            k = 2 * math.pi / (lam_px * footprint)
            ap.append((k * _mcos(th), k * _msin(th), rng.uniform(0, 2 * math.pi),
                       0.055 * min(1.0, math.sqrt(lam_px / 6.0)), rng.uniform(-0.6, 0.6)))
    # cameras on a grid, mildly tilted towards the scene centre
    cols = grid_cols or int(math.ceil(math.sqrt(n_views)))
    rows = int(math.ceil(n_views / cols))
    Ks = np.zeros((n_views, 3, 3)); Rs = np.zeros((n_views, 3, 3)); Cs = np.zeros((n_views, 3))
    for i in range(n_views):
        r, c = divmod(i, cols)
        cx = (c - (cols - 1) / 2) * spacing + rng.uniform(-0.1, 0.1) * spacing
        cy = (r - (rows - 1) / 2) * spacing + rng.uniform(-0.1, 0.1) * spacing
        C = np.array([cx, cy, cam_h + rng.uniform(-0.02, 0.02)])
        target = np.array([0.7 * cx, 0.7 * cy, 0.0])
        z = target - C; z /= _norm(z)
        x = _cross(np.array([0.0, -1.0, 0.0]), z); x /= _norm(x)
        y = _cross(z, x)
        Rs[i] = np.stack([x, y, z]); Cs[i] = C
        Ks[i] = np.array([[fpx, 0, (width - 1) / 2], [0, fpx, (height - 1) / 2], [0, 0, 1]])
    u = torch.arange(width, device=dev, dtype=torch.float64)
    v = torch.arange(height, device=dev, dtype=torch.float64)
    vv, uu = torch.meshgrid(v, u, indexing="ij")
    n_gt = n_views if gt_views is None else min(gt_views, n_views)
    gray = torch.zeros((n_views, height, width), dtype=torch.float32, device=dev)
    bgr = torch.zeros((n_views, height, width, 3), dtype=torch.uint8, device=dev) if want_bgr else None
    gt = torch.zeros((n_gt, height, width), dtype=torch.float32, device=dev)
    dmin = np.zeros(n_views, np.float32); dmax = np.zeros(n_views, np.float32)
    lo = np.full(3, np.inf); hi = np.full(3, -np.inf)
    c114 = torch.tensor(0.114, dtype=torch.float32, device=dev); c587 = torch.tensor(0.587, dtype=torch.float32, device=dev)
    c299 = torch.tensor(0.299, dtype=torch.float32, device=dev)
    inv255 = torch.tensor(1.0, dtype=torch.float32, device=dev) / 255.0
    for i in range(n_views):
        K = Ks[i]; R = torch.tensor(Rs[i], device=dev); C = torch.tensor(Cs[i], device=dev)
        rx = (uu - K[0, 2]) / K[0, 0]; ry = (vv - K[1, 2]) / K[1, 1]
        # world direction of the ray with z_cam == 1:  d = R^T (rx, ry, 1)
        dx = R[0, 0] * rx + R[1, 0] * ry + R[2, 0]
        dy = R[0, 1] * rx + R[1, 1] * ry + R[2, 1]
        dz = R[0, 2] * rx + R[1, 2] * ry + R[2, 2]
        t = (0.0 - C[2]) / dz
        for _ in range(40):
            t = (_height(C[0] + t * dx, C[1] + t * dy, hp, _sin) - C[2]) / dz
        X = C[0] + t * dx; Y = C[1] + t * dy; Z = C[2] + t * dz
        chans = []
        for ch in range(3):
            a = _albedo(X, Y, ap, float(ch - 1), _sin)
            # 8-px checker modulation (in mean-footprint units)
            chk = (torch.floor(X / (8 * footprint)) + torch.floor(Y / (8 * footprint))) % 2
            a = a * (0.92 + 0.08 * chk)
            chans.append(torch.clamp(torch.round(a * 255.0), 0, 255))
        b, g, r_ = (c_.to(torch.float32) for c_ in chans)
        # Image::toGray(BGR, normalize): cb*(B/255) + cg*(G/255) + cr*(R/255) in float, Types.inl:2377-2420
        gray[i] = c114 * (b * inv255) + c587 * (g * inv255) + c299 * (r_ * inv255)
        if bgr is not None:
            bgr[i] = torch.stack(chans, -1).to(torch.uint8)
        tf = t.to(torch.float32)
        if i < n_gt:
            gt[i] = tf
        dmin[i] = float(tf.min()) * 0.9; dmax[i] = float(tf.max()) * 1.1
        for a_, P in enumerate((X, Y, Z)):
            lo[a_] = min(lo[a_], float(P.min())); hi[a_] = max(hi[a_], float(P.max()))
    # neighbours: nearest camera centres
    n_src = min(n_src, n_views - 1)
    d2 = ((Cs[:, None, :] - Cs[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    nbr = np.argsort(d2, axis=1, kind="stable")[:, :n_src].astype(np.int32)
    return dict(width=width, height=height, gray=gray, bgr=bgr, K=Ks, R=Rs, C=Cs, gt_depth=gt, neighbors=nbr,
                dmin=dmin.astype(np.float32), dmax=dmax.astype(np.float32), diameter=float(np.linalg.norm(hi - lo)),
                meta={"seed": seed, "spacing": spacing, "cols": cols})


def make_scene(n_views: int, width: int, height: int, n_src: int = 8, seed: int = SEED,
               device: str | torch.device = "cpu", spacing: float = 0.12, grid_cols: int | None = None,
               gray_only: bool = False, exact: bool = False) -> Scene:
    """Render `n_views` views of the seeded scene at width x height; numpy arrays on the host."""
    t = make_scene_torch(n_views, width, height, n_src, seed, device, spacing, grid_cols, want_bgr=not gray_only, exact=exact)
    bgr = t["bgr"].cpu().numpy() if t["bgr"] is not None else np.zeros((0,), np.uint8)
    return Scene(width, height, t["gray"].cpu().numpy(), bgr, t["K"], t["R"], t["C"], t["gt_depth"].cpu().numpy(),
                 t["neighbors"], t["dmin"], t["dmax"], t["diameter"], t["meta"])

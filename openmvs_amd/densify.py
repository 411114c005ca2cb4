"""Host-side mirror of Scene::ComputeDepthMaps (libs/MVS/SceneDensify.cpp:1754-1982) for the HBM-resident engine:
photometric pass, geometric-consistency rounds, per-map post-filters after the last round (nOptimize bits REMOVE_SPECKLES
and FILL_GAPS, :1884-1886,1919-1920,2069-2093), cross-view filter (ADJUST_FILTER, :1955-1980), optional `.dmap` output
(:2095-2117).  Everything between the image upload and the final download stays on the device."""
from __future__ import annotations

import os

from . import dmap as _dmap

REMOVE_SPECKLES, FILL_GAPS, ADJUST_FILTER = 1, 2, 4   # OPTDENSE::DepthFlags, libs/MVS/DepthMap.h:87-92


def compute_depth_maps(engine, view_ids, params, n_optimize: int = 7, b_filter_adjust: bool = True,
                       n_speckle_size: int = 100, n_ipol_gap_size: int = 7, f_depth_diff_threshold: float = 0.01,
                       n_min_views_filter: int = 2, n_min_views_filter_adjust: int = 1, init_depth=None, init_normal=None):
    """Runs the reference's dense schedule for `view_ids` on a loaded scene (engine.scene_load / scene_set_view).
    `init_depth` / `init_normal` (dicts view id -> map) seed the photometric pass like `InitViews(..., loadDepthMaps=0)` does
    (SceneDensify.cpp:418-460); views without an entry start from random planes."""
    ids = list(view_ids)
    G = int(params.nEstimationGeometricIters)

    def post():
        if n_optimize & REMOVE_SPECKLES:
            engine.scene_remove_small_segments(ids, n_speckle_size, f_depth_diff_threshold)
        if n_optimize & FILL_GAPS:
            engine.scene_gap_interpolation(ids, n_ipol_gap_size, f_depth_diff_threshold)

    engine.Init(False)
    for v in ids:
        engine.scene_reset_view(v)
        if init_depth is not None and v in init_depth:
            engine.scene_set_maps(v, init_depth[v], None if init_normal is None else init_normal.get(v))
    engine.scene_estimate(ids, -1, params)
    if G == 0:
        post()
    for g in range(G):
        engine.scene_commit_round()
        engine.Init(True)
        engine.scene_estimate(ids, g, params)
        if g + 1 == G:
            post()
    if n_optimize & ADJUST_FILTER:
        engine.scene_filter(ids, b_filter_adjust, n_min_views_filter, n_min_views_filter_adjust, f_depth_diff_threshold, commit=True)


def save_depth_maps(engine, scene, view_ids, out_dir: str, image_names=None):
    """DepthData::Save for every view: depthNNNN.dmap next to each other (ComposeDepthFilePath, DepthMap.h:72)."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for v in view_ids:
        d, n, c = engine.scene_get_maps(v)
        ids = [int(v)] + [int(i) for i in scene.neighbors[v]]
        name = image_names[v] if image_names else "images/%05d.jpg" % v
        p = os.path.join(out_dir, _dmap.depth_file_name(int(v)))
        _dmap.save(p, name, ids, (scene.width, scene.height), scene.K[v], scene.R[v], scene.C[v], float(scene.dmin[v]), float(scene.dmax[v]), d, n, c)
        paths.append(p)
    return paths


# ---- scene front end: *.mvs + images -> engine scene -----------------------------------------------------------------------------

class SceneViews:
    """What `Scene::ComputeDepthMaps` has in hand after its preparation steps (libs/MVS/SceneDensify.cpp:1772-1870) for a scene whose
    images share one resolution: gray images, pixel cameras, neighbour lists, depth ranges and the sparse initial maps.  Same
    attribute names as `synth.Scene`, so `PatchMatchHIP.scene_load` takes either."""

    def __init__(self):
        self.width = self.height = 0
        self.gray = []; self.K = []; self.R = []; self.C = []
        self.dmin = []; self.dmax = []; self.neighbors = []; self.view_scores = []
        self.init_depth = {}; self.init_normal = {}
        self.names = []; self.ids = []          # ids: images that passed view selection, in scene order
        self.masks = {}; self.mask_option = False   # ignore masks by image index (1 = process, 0 = ignore); mask_option: OPTDENSE::nIgnoreMaskLabel >= 0

    @property
    def n_views(self):
        return len(self.gray)


def _area_tab(ssize: int, dsize: int):
    """OpenCV's computeResizeAreaTab (imgproc/src/resize.cpp) for one axis: for every destination cell the source cells it covers and their float weights, in order.
    Returns (di, si, alpha) arrays; scale = ssize / dsize in double, the partial cells at both ends weighted by their covered fraction of the cell width."""
    import math
    import numpy as np
    scale = ssize / dsize
    di, si, al = [], [], []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            di.append(dx); si.append(sx1 - 1); al.append(np.float32((sx1 - fsx1) / cell))
        for sx in range(sx1, sx2):
            di.append(dx); si.append(sx); al.append(np.float32(1.0 / cell))
        if fsx2 - sx2 > 1e-3:
            di.append(dx); si.append(sx2); al.append(np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell))
    return np.asarray(di, np.int64), np.asarray(si, np.int64), np.asarray(al, np.float32)


def _accumulate_in_order(dst_n, di, vals):
    """out[di[k]] += vals[k] in k order, in float32 (entries of one destination cell are consecutive: the j-th entries of all cells are added together, j = 0, 1, ...)."""
    import numpy as np
    out = np.zeros((dst_n,) + vals.shape[1:], np.float32)
    first = np.r_[True, di[1:] != di[:-1]]
    start = np.maximum.accumulate(np.where(first, np.arange(len(di)), 0))
    rank = np.arange(len(di)) - start
    for j in range(int(rank.max()) + 1 if len(di) else 0):
        m = rank == j
        out[di[m]] = out[di[m]] + vals[m]
    return out


def _resize_area_u8(img, w: int, h: int):
    """cv::resize(img, Size(w, h), 0, 0, INTER_AREA) of an 8-bit image that shrinks on both axes (Image::ResizeImage, libs/MVS/Image.cpp:139-155, reached by
    --max-resolution / nResolutionLevel).  Exact integer factors: OpenCV's integer path, the box mean rounded to nearest-even.  Any other factor: its general area path
    (computeResizeAreaTab + ResizeArea_Invoker, imgproc/src/resize.cpp, restated here -- OpenCV is not vendored with the reference, so this is unpinned): per source
    row the x cells weighted in float and accumulated in table order, the rows then weighted and accumulated in float per destination row, saturate_cast<uchar> = round to
    nearest-even."""
    import numpy as np
    H, W = img.shape[:2]
    if w > W or h > H:
        raise NotImplementedError("image resize %dx%d -> %dx%d: INTER_AREA is implemented for shrinking only (OpenCV enlarges with the bilinear path)" % (W, H, w, h))
    if W % w == 0 and H % h == 0 and W // w == H // h:
        f = W // w
        s = img.reshape(h, f, w, f, -1).astype(np.float32).sum(axis=(1, 3))
        out = np.rint(s * np.float32(1.0 / (f * f))).astype(np.uint8)
        return out if img.ndim == 3 else out[..., 0]
    src = img.reshape(H, W, -1).astype(np.float32)
    xdi, xsi, xal = _area_tab(W, w)
    ydi, ysi, yal = _area_tab(H, h)
    # every source row that takes part: buf[dx] = sum over its x cells, in table order
    rows = np.unique(ysi)
    buf = {}
    for sy in rows:
        buf[int(sy)] = _accumulate_in_order(w, xdi, src[sy][xsi] * xal[:, None])
    # destination rows: sum[dx] = beta_0 * buf_0, then += beta_k * buf_k in table order
    vals = np.stack([yal[k] * buf[int(ysi[k])] for k in range(len(ydi))])
    out = _accumulate_in_order(h, ydi, vals)
    out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out if img.ndim == 3 else out[..., 0]


def load_scene(mvs_path: str, opt=None, image_loader=None, view_neighbors_file=None, ignore_mask_label=None, mask_path=None, mask_loader=None):
    """Reads an MVSI scene and its images and runs view selection + depth initialisation for every valid image.

    `opt`: a `views.DenseOptions` or the whole option table (`optdense.OptDense`, e.g. from `optdense.load(<--dense-config-file>)`).
    `view_neighbors_file`: `DensifyPointCloud --view-neighbors-file` (Scene::LoadViewNeighbors, applied right after the scene is loaded,
    apps/DensifyPointCloud/DensifyPointCloud.cpp:342-343): the listed neighbours replace view selection for those images.
    `ignore_mask_label` (`--ignore-mask-label`; default: the option table's nIgnoreMaskLabel, else off) >= 0: every image's segmentation mask is looked up like
    `DepthEstimator::ImportIgnoreMask` does (the scene's mask name, else <image>.mask.png; `mask_path` = `--mask-path`), read by `mask_loader(path) -> (h,w) labels`
    (default PIL), brought to the working resolution with INTER_NEAREST, and lands in `SceneViews.masks[i]` (1 = process, 0 = ignore) for `scene_set_mask`; an image
    whose mask cannot be read has none, as there (a warning in the reference), but `SceneViews.mask_option` tells the engine that the option is on (SceneDensify.cpp:661).
    `image_loader(path) -> (h,w,3) uint8 RGB` defaults to PIL.  Returns a `SceneViews`."""
    import numpy as np
    from . import mvsi, views
    opt = opt or views.DenseOptions()
    if ignore_mask_label is None:
        ignore_mask_label = int(getattr(opt, "nIgnoreMaskLabel", -1))
    if hasattr(opt, "dense_options"):
        opt = opt.dense_options()
    sc = mvsi.load(mvs_path)
    if view_neighbors_file:
        mvsi.load_view_neighbors(sc, view_neighbors_file)
    base = os.path.dirname(os.path.abspath(mvs_path))
    if image_loader is None:
        def image_loader(p):
            from PIL import Image
            with Image.open(p) as im:
                return np.asarray(im.convert("RGB"))
    sv = SceneViews()
    rgbs, sizes = [], []
    for im in sc.images:
        p = im.name if os.path.isabs(im.name) else os.path.join(base, im.name)
        rgb = image_loader(p)
        H, W = rgb.shape[:2]
        res, _ = views.compute_max_resolution(W, H, opt.nResolutionLevel, opt.nMinResolution, opt.nMaxResolution)
        w, h = views.resized_size(W, H, res)
        if (w, h) != (W, H):
            rgb = _resize_area_u8(rgb, w, h)
        rgbs.append(rgb); sizes.append((w, h))
    if len(set(sizes)) != 1:
        raise NotImplementedError("the batch scene interface needs one image resolution; got %s" % sorted(set(sizes)))
    sv.width, sv.height = sizes[0]
    if ignore_mask_label >= 0:
        sv.mask_option = True
        if mask_loader is None:
            def mask_loader(p):
                from PIL import Image
                with Image.open(p) as im:
                    return np.asarray(im)
        for i, im in enumerate(sc.images):
            name = views.mask_file_name(im.name, im.mask_name, mask_path)
            p = name if os.path.isabs(name) else os.path.join(base, name)
            if mask_path and not os.path.exists(p):
                raise FileNotFoundError("Mask image %s not found" % p)          # DensifyPointCloud.cpp:315-318
            try:
                labels = mask_loader(p)
            except (OSError, ValueError):
                continue                                                         # "warning: can not load the segmentation mask": the image is estimated unmasked
            sv.masks[i] = views.import_ignore_mask(labels, sizes[i], ignore_mask_label)
    cams = views.Cameras(sc, sizes)
    for i, im in enumerate(sc.images):
        sv.gray.append(views.to_gray(rgbs[i])); sv.names.append(im.name)
        sv.K.append(cams.K[i]); sv.R.append(cams.R[i]); sv.C.append(cams.C[i])
        sel = views.select_views(sc, cams, i, opt) if im.is_valid() else None
        if sel is None:
            sv.neighbors.append(np.zeros(0, np.int32)); sv.view_scores.append(None); sv.dmin.append(0.1); sv.dmax.append(100.0)
            continue
        nb, points, _ = sel
        if np.any(np.abs(nb["scale"] - 1) >= 0.15):          # DepthData::ViewData::NeedScaleImage, libs/MVS/DepthMap.h:194-197
            raise NotImplementedError("image %d: a neighbour needs rescaling (scale %s)" % (i, nb["scale"]))
        d, n, dmin, dmax = views.init_depth_map(sc, cams, i, points, opt)
        sv.ids.append(i); sv.neighbors.append(nb["ID"].astype(np.int32)); sv.view_scores.append(nb)
        sv.dmin.append(dmin); sv.dmax.append(dmax); sv.init_depth[i] = d; sv.init_normal[i] = n
    return sv

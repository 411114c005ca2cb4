"""Host-side mirror of Scene::ComputeDepthMaps (libs/MVS/SceneDensify.cpp:1754-1982) for the HBM-resident engine:
photometric pass, geometric-consistency rounds, per-map post-filters after the last round (nOptimize bits REMOVE_SPECKLES
and FILL_GAPS, :1884-1886,1919-1920,2069-2093), cross-view filter (ADJUST_FILTER, :1955-1980), optional `.dmap` output
(:2095-2117).  Everything between the image upload and the final download stays on the device."""
from __future__ import annotations

import os

import numpy as np

from . import dmap as _dmap

REMOVE_SPECKLES, FILL_GAPS, ADJUST_FILTER = 1, 2, 4   # OPTDENSE::DepthFlags, libs/MVS/DepthMap.h:87-92


def compute_depth_maps(engine, view_ids, params, n_optimize: int = 7, b_filter_adjust: bool = True,
                       n_speckle_size: int = 100, n_ipol_gap_size: int = 7, f_depth_diff_threshold: float = 0.01,
                       n_min_views_filter: int = 2, n_min_views_filter_adjust: int = 1, init_depth=None, init_normal=None, scene=None, dmap_dir=None, image_names=None):
    """Runs the reference's dense schedule for `view_ids` on a loaded scene (engine.scene_load / scene_set_view).
    `init_depth` / `init_normal` (dicts view id -> map) seed the photometric pass like `InitViews(..., loadDepthMaps=0)` does
    (SceneDensify.cpp:418-460); views without an entry start from random planes.
    `scene`: the `SceneViews` the engine was loaded from, needed when it holds resampled copies of neighbours (`alias_of`, ViewData::ScaleImage): at every round
    boundary a copy is handed the depth map of the image it stands for, at that image's size and camera (the neighbour's saved .dmap, SceneDensify.cpp:378-393), and
    before the cross-view filter the reference views get their image neighbours back (FilterDepthMap reads arrDepthData[ID], :1049-1299).
    `dmap_dir` (needs `scene`): the reference's file-based checkpoint contract (SceneDensify.cpp:2010-2029,1943-1950).  A view whose `depthNNNN.dmap` is already there
    is NOT estimated in the photometric pass: its depth, normal and confidence maps are read back and the schedule carries on from them (every view takes part in the geometric
    rounds again, as there); every map is written when it has been estimated -- after the photometric pass, after each geometric round (the reference's .geo.dmap + rename)
    and after the filters -- each file atomically (.tmp + rename, DepthData::Save).  Returns the views whose maps were resumed from files."""
    ids = list(view_ids)
    if dmap_dir is not None and scene is None:
        raise ValueError("dmap_dir needs the scene (cameras, neighbours and depth ranges go into the files)")
    G = int(params.nEstimationGeometricIters)
    alias_of = dict(getattr(scene, "alias_of", None) or {})

    def hand_depths_to_copies():
        for a, j in alias_of.items():
            engine.scene_set_source_depth(a, engine.scene_get_maps(j)[0], scene.K[j], scene.R[j], scene.C[j])

    def image_neighbours():
        for v in ids:
            if alias_of and not np.array_equal(scene.estimate_neighbors[v], scene.neighbors[v]):
                engine.scene_set_view(v, None, scene.K[v], scene.R[v], scene.C[v], float(scene.dmin[v]), float(scene.dmax[v]), scene.neighbors[v])

    def post():
        if n_optimize & REMOVE_SPECKLES:
            engine.scene_remove_small_segments(ids, n_speckle_size, f_depth_diff_threshold)
        if n_optimize & FILL_GAPS:
            engine.scene_gap_interpolation(ids, n_ipol_gap_size, f_depth_diff_threshold)

    def save(which):
        if dmap_dir is not None and which:
            save_depth_maps(engine, scene, which, dmap_dir, image_names)

    engine.Init(False)
    resumed = []
    for v in ids:
        engine.scene_reset_view(v)
        done = os.path.join(dmap_dir, _dmap.depth_file_name(int(v))) if dmap_dir is not None else None
        if done is not None and os.path.exists(done):          # depthmapComputed (SceneDensify.cpp:2010): loaded instead of estimated
            f = _dmap.load(done)
            want = tuple(scene.sizes[v])[::-1] if getattr(scene, "sizes", None) else (scene.height, scene.width)
            if f["depth_map"].shape != want or f.get("confidence_map") is None:
                raise ValueError("%s does not hold the maps of view %d (%s, expected %s with a confidence map)" % (done, v, f["depth_map"].shape, want))
            normal = f.get("normal_map")
            if normal is None:                                     # a depth map stored without normals gets them from its depths (EstimateNormalMap, SceneDensify.cpp:411-414)
                from . import views as _views
                normal = _views.estimate_normal_map(scene.K[v], f["depth_map"])
            engine.scene_set_maps(v, f["depth_map"], normal)
            engine.scene_set_conf(v, f["confidence_map"])
            resumed.append(v)
        elif init_depth is not None and v in init_depth:
            engine.scene_set_maps(v, init_depth[v], None if init_normal is None else init_normal.get(v))
    todo = [v for v in ids if v not in resumed]
    if todo:
        engine.scene_estimate(todo, -1, params)
    if G == 0:
        post()
    save(ids if (G == 0 and n_optimize & (REMOVE_SPECKLES | FILL_GAPS)) else todo)
    for g in range(G):
        engine.scene_commit_round()
        hand_depths_to_copies()
        engine.Init(True)
        engine.scene_estimate(ids, g, params)
        if g + 1 == G:
            post()
        save(ids)
    image_neighbours()
    if n_optimize & ADJUST_FILTER:
        engine.scene_filter(ids, b_filter_adjust, n_min_views_filter, n_min_views_filter_adjust, f_depth_diff_threshold, commit=True)
        save(ids)
    return resumed


def save_depth_maps(engine, scene, view_ids, out_dir: str, image_names=None):
    """DepthData::Save for every view: depthNNNN.dmap next to each other (ComposeDepthFilePath, DepthMap.h:72)."""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for v in view_ids:
        d, n, c = engine.scene_get_maps(v)
        ids = [int(v)] + [int(i) for i in scene.neighbors[v]]
        name = image_names[v] if image_names else "images/%05d.jpg" % v
        p = os.path.join(out_dir, _dmap.depth_file_name(int(v)))
        size = tuple(scene.sizes[v]) if getattr(scene, "sizes", None) else (scene.width, scene.height)
        _dmap.save(p, name, ids, size, scene.K[v], scene.R[v], scene.C[v], float(scene.dmin[v]), float(scene.dmax[v]), d, n, c)
        paths.append(p)
    return paths


# ---- scene front end: *.mvs + images -> engine scene -----------------------------------------------------------------------------

class SceneViews:
    """What `Scene::ComputeDepthMaps` has in hand after its preparation steps (libs/MVS/SceneDensify.cpp:1772-1870) for a scene:
    gray images, pixel cameras, neighbour lists, depth ranges and the sparse initial maps.  Same
    attribute names as `synth.Scene`, so `PatchMatchHIP.scene_load` takes either."""

    def __init__(self):
        self.width = self.height = 0
        self.gray = []; self.K = []; self.R = []; self.C = []
        self.dmin = []; self.dmax = []; self.neighbors = []; self.view_scores = []
        self.init_depth = {}; self.init_normal = {}
        self.names = []; self.ids = []          # ids: images that passed view selection, in scene order
        self.masks = {}; self.mask_option = False   # ignore masks by image index (1 = process, 0 = ignore); mask_option: OPTDENSE::nIgnoreMaskLabel >= 0
        self.sizes = []                             # (w, h) of every slot: images of another size than the scene's carry their own (pmhip_scene_set_view_sized)
        self.estimate_neighbors = None              # per slot: the slots the ESTIMATION reads (a resampled copy where ViewData::ScaleImage applies); None = `neighbors`
        self.alias_of = {}                          # extra source-only slot -> the image it is a resampled copy of
        self.bgr = []                               # the colour images at the working resolution (B, G, R), for the fused cloud's colours
        self.all_view_scores = {}                   # image -> its WHOLE scored neighbour list (Image::neighbors: fusion order, and what a dense archive stores)
        self.avg_depth = {}                         # image -> average depth of its sparse points (Image::avgDepth)

    @property
    def n_views(self):
        return len(self.gray)


def _area_tab(ssize: int, dsize: int, scale: float | None = None):
    """OpenCV's computeResizeAreaTab (imgproc/src/resize.cpp) for one axis: for every destination cell the source cells it covers and their float weights, in order.
    Returns (di, si, alpha) arrays; scale = ssize / dsize in double, the partial cells at both ends weighted by their covered fraction of the cell width."""
    import math
    import numpy as np
    if scale is None:
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
    --max-resolution / nResolutionLevel).  Factor 2 (the default halving, nResolutionLevel = 1): OpenCV's ResizeAreaFastVec for integer pixels, (a + b + c + d + 2) >> 2
    -- round half UP, in integers.  Other exact integer factors (the two axes may have different ones): the integer box sum times the float 1 / (fx fy), saturate_cast = round to nearest-even.  Any other factor: its general area path
    (computeResizeAreaTab + ResizeArea_Invoker, imgproc/src/resize.cpp, restated here -- OpenCV is not vendored with the reference, so this is unpinned): per source
    row the x cells weighted in float and accumulated in table order, the rows then weighted and accumulated in float per destination row, saturate_cast<uchar> = round to
    nearest-even."""
    import numpy as np
    H, W = img.shape[:2]
    if w > W or h > H:
        raise NotImplementedError("image resize %dx%d -> %dx%d: INTER_AREA is implemented for shrinking only (OpenCV enlarges with the bilinear path)" % (W, H, w, h))
    if W % w == 0 and H % h == 0:          # OpenCV's `is_area_fast`: both factors are integers (they may differ)
        fx, fy = W // w, H // h
        s = img.reshape(h, fy, w, fx, -1).astype(np.int64).sum(axis=(1, 3))
        if fx == 2 and fy == 2:
            out = ((s + 2) >> 2).astype(np.uint8)
        else:
            out = np.rint(s.astype(np.float32) * np.float32(1.0 / (fx * fy))).astype(np.uint8)
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


# ---- DepthData::ViewData::ScaleImage: a neighbour whose footprint differs by 15 % or more is resampled (libs/MVS/DepthMap.h:193-204) -----------------------------

def need_scale_image(scale) -> bool:
    """`NeedScaleImage`: ABS(scale - 1.f) >= 0.15f, in float."""
    import numpy as np
    return bool(np.abs(np.float32(scale) - np.float32(1)) >= np.float32(0.15))


def _resize_area_f32(img, w: int, h: int, scale: float):
    """cv::resize(img, Size(), 1/scale, 1/scale, INTER_AREA) of a float image that shrinks (`scale` = OpenCV's scale_x = scale_y > 1, (w, h) = the destination size it derived).
    An integer `scale`: the "area fast" path -- full f x f blocks are ((a+b)+(c+d))*0.25 for f = 2, else the row-major running sum times 1/f^2; blocks cut by the right /
    bottom border average what is there (the statement of oracle/pm_oracle.cpp's resizeArea, which the estimator's pyramid uses).  Any other factor: the general path
    (computeResizeAreaTab + ResizeArea_Invoker) in float, as `_resize_area_u8` without the final rounding.  OpenCV is not vendored with the reference: unpinned."""
    import numpy as np
    H, W = img.shape
    src = np.ascontiguousarray(img, np.float32)
    f = int(np.rint(scale))
    if abs(scale - f) < np.finfo(np.float64).eps:
        out = np.zeros((h, w), np.float32)
        fw, fh = min(w, W // f), min(h, H // f)                                # destination cells whose block is complete
        if fw and fh:
            blk = src[:fh * f, :fw * f].reshape(fh, f, fw, f)
            if f == 2:
                full = ((blk[:, 0, :, 0] + blk[:, 0, :, 1]) + (blk[:, 1, :, 0] + blk[:, 1, :, 1])) * np.float32(0.25)
            else:
                acc = np.zeros((fh, fw), np.float32)
                for j in range(f):
                    for i in range(f):
                        acc = acc + blk[:, j, :, i]
                full = acc * np.float32(1.0 / (f * f))
            out[:fh, :fw] = full
        for y in range(h):                                                     # cells cut by the border (sizes not divisible by f)
            for x in range(w):
                if y < fh and x < fw:
                    continue
                blk = src[y * f:min(y * f + f, H), x * f:min(x * f + f, W)]
                if blk.size:
                    acc = np.float32(0)
                    for v in blk.reshape(-1):
                        acc = np.float32(acc + v)
                    out[y, x] = acc / np.float32(blk.size)
        return out
    xdi, xsi, xal = _area_tab(W, w, scale)
    ydi, ysi, yal = _area_tab(H, h, scale)
    buf = {int(sy): _accumulate_in_order(w, xdi, src[sy][xsi] * xal) for sy in np.unique(ysi)}
    vals = np.stack([yal[k] * buf[int(ysi[k])] for k in range(len(ydi))])
    return _accumulate_in_order(h, ydi, vals)


def _cubic_coeffs(x):
    """interpolateCubic (imgproc/src/resize.cpp), A = -0.75, in float."""
    import numpy as np
    f = np.float32
    A = f(-0.75)
    x = x.astype(np.float32)
    c0 = ((A * (x + f(1)) - f(5) * A) * (x + f(1)) + f(8) * A) * (x + f(1)) - f(4) * A
    c1 = ((A + f(2)) * x - (A + f(3))) * x * x + f(1)
    c2 = ((A + f(2)) * (f(1) - x) - (A + f(3))) * (f(1) - x) * (f(1) - x) + f(1)
    c3 = f(1) - c0 - c1 - c2
    return c0, c1, c2, c3


def _resize_cubic_f32(img, w: int, h: int, scale: float):
    """cv::resize(img, Size(), 1/scale, 1/scale, INTER_CUBIC) of a float image (`scale` = OpenCV's scale_x = scale_y < 1 when enlarging): resizeGeneric_ with
    HResizeCubic / VResizeCubic in their scalar form -- source position (d + 0.5) * scale - 0.5 in float, four taps around its floor with interpolateCubic's weights,
    taps outside the image clamped to the border, rows first, then columns, each as ((t0*c0 + t1*c1) + t2*c2) + t3*c3 in float.  OpenCV's own SIMD column pass fuses and
    reorders these products depending on how it was built, so the last bit of this resampling is not defined by the reference either: unpinned."""
    import numpy as np
    H, W = img.shape
    src = np.ascontiguousarray(img, np.float32)

    def taps(n_dst, n_src):
        fx = ((np.arange(n_dst) + 0.5) * scale - 0.5).astype(np.float32)
        sx = np.floor(fx).astype(np.int64)
        fx = fx - sx.astype(np.float32)
        idx = np.clip(sx[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
        return idx, _cubic_coeffs(fx)

    xi, (a0, a1, a2, a3) = taps(w, W)
    rows = ((src[:, xi[:, 0]] * a0 + src[:, xi[:, 1]] * a1) + src[:, xi[:, 2]] * a2) + src[:, xi[:, 3]] * a3          # [H, w]
    yi, (b0, b1, b2, b3) = taps(h, H)
    return (((rows[yi[:, 0]] * b0[:, None] + rows[yi[:, 1]] * b1[:, None]) + rows[yi[:, 2]] * b2[:, None]) + rows[yi[:, 3]] * b3[:, None]).astype(np.float32)


def scale_image(gray, scale):
    """`DepthData::ViewData::ScaleImage` (libs/MVS/DepthMap.h:198-204): None if the scale is within 15 % of 1, else cv::resize(image, Size(), scale, scale,
    scale > 1 ? INTER_CUBIC : INTER_AREA) -- destination size saturate_cast<int>(size * scale) (round half to even)."""
    import numpy as np
    if not need_scale_image(scale):
        return None
    s = float(np.float32(scale))
    H, W = gray.shape
    w, h = int(np.rint(W * s)), int(np.rint(H * s))
    if w < 1 or h < 1:
        raise ValueError("scale %g leaves no image" % s)
    return _resize_cubic_f32(gray, w, h, 1.0 / s) if s > 1 else _resize_area_f32(gray, w, h, 1.0 / s)


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
    sv.width, sv.height = max(set(sizes), key=lambda wh: (sizes.count(wh), -sizes.index(wh)))     # the scene's size: the most frequent one; the other images carry their own
    sv.sizes = list(sizes)
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
        sv.bgr.append(np.ascontiguousarray(rgbs[i][..., ::-1]))
        sv.K.append(cams.K[i]); sv.R.append(cams.R[i]); sv.C.append(cams.C[i])
        whole = []
        sel = views.select_views(sc, cams, i, opt, all_neighbors=whole) if im.is_valid() else None
        if whole:
            sv.all_view_scores[i] = whole[0]
        if sel is None:
            sv.neighbors.append(np.zeros(0, np.int32)); sv.view_scores.append(None); sv.dmin.append(0.1); sv.dmax.append(100.0)
            continue
        nb, points, sv.avg_depth[i] = sel
        d, n, dmin, dmax = views.init_depth_map(sc, cams, i, points, opt)
        sv.ids.append(i); sv.neighbors.append(nb["ID"].astype(np.int32)); sv.view_scores.append(nb)
        sv.dmin.append(dmin); sv.dmax.append(dmax); sv.init_depth[i] = d; sv.init_normal[i] = n
    # InitViews, SceneDensify.cpp:333-352: a neighbour whose footprint differs by 15 % or more enters the estimation as a RESAMPLED image with the camera of that size
    # (ViewData::ScaleImage).  Each (image, scale) pair becomes an extra source-only slot behind the images; the reference view's estimation list points at it, while
    # `neighbors` keeps the image itself -- whose own depth map, at its own size and camera, is what the geometric rounds, the cross-view filter and the fusion read
    # (cameraDepthMap, :378-393; FilterDepthMap / FuseDepthMaps go through arrDepthData[ID]).
    n_img = len(sc.images)
    sv.estimate_neighbors = [nb.copy() for nb in sv.neighbors]
    made = {}
    for i in sv.ids:
        for k, v in enumerate(sv.view_scores[i]):
            j, scale = int(v["ID"]), np.float32(v["scale"])
            if not need_scale_image(scale):
                continue
            key = (j, float(scale))
            if key not in made:
                img = scale_image(sv.gray[j], scale)
                Kj, Rj, Cj, w, h = sc.camera(j, (img.shape[1], img.shape[0]))          # Image::GetCamera(platforms, image.size())
                made[key] = len(sv.gray)
                sv.alias_of[made[key]] = j
                sv.gray.append(img); sv.K.append(Kj); sv.R.append(Rj); sv.C.append(Cj); sv.sizes.append((w, h)); sv.names.append(sv.names[j])
                sv.dmin.append(sv.dmin[j]); sv.dmax.append(sv.dmax[j]); sv.neighbors.append(np.zeros(0, np.int32)); sv.estimate_neighbors.append(np.zeros(0, np.int32))
                sv.view_scores.append(None)
            sv.estimate_neighbors[i][k] = made[key]
    assert len(sv.gray) == n_img + len(sv.alias_of)
    return sv


# ---- the consumer of the depth maps: fusion and the dense scene archive ---------------------------------------------------------------------------

def fuse_order(scene) -> list:
    """Processing order of `DepthMapsData::FuseDepthMaps` (libs/MVS/SceneDensify.cpp:1406-1452): the images that have a depth map, best connected first -- by the size
    of the image's whole neighbour list (`Image::neighbors`), images without neighbours dropped; equal counts in index order (the reference's sort is not stable there)."""
    n_all = {i: len(scene.all_view_scores[i]) if i in getattr(scene, "all_view_scores", {}) else len(scene.neighbors[i]) for i in scene.ids}
    return sorted((i for i in scene.ids if n_all[i] > 0), key=lambda i: (-n_all[i], i))


def fuse_depth_maps(engine, scene, opt=None, bgr=None) -> dict:
    """`Scene::DenseReconstruction`'s last step (SceneDensify.cpp:1695-1712) on the resident maps: `FuseDepthMaps(pointcloud, nEstimateColors == 2, nEstimateNormals == 2)`
    with `opt` = an `optdense.OptDense` (default: the table's defaults with the application's `--estimate-normals 2`).  `bgr`: {image: (h, w, 3) uint8 BGR} for the colours.
    Returns the MVS::PointCloud fields of `PatchMatchHIP.scene_fuse`."""
    n_min, f_depth, f_normal, colors, normals = 2, 0.01, 25.0, 2, 2
    if opt is not None:
        n_min, f_depth, f_normal, colors, normals = int(opt.nMinViewsFuse), float(opt.fDepthDiffThreshold), float(opt.fNormalDiffThreshold), int(opt.nEstimateColors), int(opt.nEstimateNormals)
    want_color = colors == 2 and bgr is not None
    if want_color:
        for i in range(len(scene.gray) - len(scene.alias_of)):
            engine.scene_set_color(i, bgr[i])
    return engine.scene_fuse(fuse_order(scene), n_min, f_depth, f_normal, want_color, normals == 2)


def save_dense_scene(mvs_in: str, mvs_out: str, cloud: dict, scene=None, version: int | None = None) -> None:
    """What `DensifyPointCloud` leaves behind as `<scene>_dense.mvs` (`Scene::SaveInterface`, libs/MVS/Scene.cpp:218-300): the input archive with the fused cloud in place
    of the sparse points -- position, the images that see the point with the fusion weight as `confidence`, normal, colour (Col3: B, G, R) -- and, when `scene`
    (the `SceneViews`) is given and the archive version stores them (> 6), every image's whole scored neighbour list and average depth."""
    import numpy as np
    from . import mvsi
    sc = mvsi.load(mvs_in)
    P = int(cloud["nPoints"])
    sc.vertices = np.ascontiguousarray(cloud["points"], np.float32).reshape(P, 3)
    sc.vertex_view_start = np.asarray(cloud["viewStart"], np.int64)
    vv = np.zeros(len(cloud["views"]), mvsi.VIEW_DTYPE)
    vv["image_id"] = cloud["views"]; vv["confidence"] = cloud["weights"]
    sc.vertex_views = vv
    sc.vertices_normal = np.zeros((0, 3), np.float32) if cloud.get("normals") is None else np.ascontiguousarray(cloud["normals"], np.float32).reshape(P, 3)
    sc.vertices_color = np.zeros((0, 3), np.uint8) if cloud.get("colors") is None else np.ascontiguousarray(cloud["colors"], np.uint8).reshape(P, 3)
    if scene is not None:
        for i, nb in getattr(scene, "all_view_scores", {}).items():
            sc.images[i].view_scores = np.ascontiguousarray(nb, mvsi.VIEW_SCORE_DTYPE)
        for i, a in getattr(scene, "avg_depth", {}).items():
            sc.images[i].avg_depth = float(a)
    mvsi.save(mvs_out, sc, version=version)


def dense_reconstruction(engine, mvs_in: str, mvs_out: str | None = None, opt=None, seed: int = 0, fusion_mode: int = 0, dmap_dir: str | None = None, **load_args):
    """`Scene::DenseReconstruction(nFusionMode)` (libs/MVS/SceneDensify.cpp:1655-1750) for the PatchMatch path on one engine: prepare the views (`load_scene`), estimate all
    depth maps with the geometric rounds and the filters the option table asks for (`compute_depth_maps`; with `dmap_dir` under the reference's file contract), and -- unless
    `fusion_mode` is 1, "export depth maps only" -- fuse them and write `<scene>_dense.mvs` to `mvs_out`.  `opt`: an `optdense.OptDense` (default: the table's defaults with the
    application's own `--number-views 8`, `--estimate-normals 2`, `--number-views-fuse` left at the table's 2); `load_args` go to `load_scene` (image_loader,
    view_neighbors_file, ignore_mask_label, mask_path, mask_loader).  The SGM modes (-1, -2) are openmvs_amd.sgm_pipeline's.  Returns (SceneViews, cloud or None)."""
    from . import optdense
    if fusion_mode not in (0, 1):
        raise ValueError("fusion_mode %d: the PatchMatch path is modes 0 (estimate + fuse) and 1 (depth maps only)" % fusion_mode)
    if opt is None:
        opt = optdense.defaults(); opt.nNumViews = 8; opt.nEstimateNormals = 2
    sv = load_scene(mvs_in, opt=opt, **load_args)
    engine.scene_load(sv, n_levels=int(opt.nSubResolutionLevels))
    compute_depth_maps(engine, sv.ids, opt.params(seed), n_optimize=int(opt.nOptimize), b_filter_adjust=bool(opt.bFilterAdjust), n_speckle_size=int(opt.nSpeckleSize),
                       n_ipol_gap_size=int(opt.nIpolGapSize), f_depth_diff_threshold=float(opt.fDepthDiffThreshold), n_min_views_filter=int(opt.nMinViewsFilter),
                       n_min_views_filter_adjust=int(opt.nMinViewsFilterAdjust), init_depth=sv.init_depth, init_normal=sv.init_normal, scene=sv, dmap_dir=dmap_dir,
                       image_names=sv.names)
    if fusion_mode == 1:
        return sv, None
    cloud = fuse_depth_maps(engine, sv, opt, bgr=sv.bgr)
    if mvs_out:
        save_dense_scene(mvs_in, mvs_out, cloud, sv)
    return sv, cloud

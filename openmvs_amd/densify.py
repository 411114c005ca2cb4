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
                       n_min_views_filter: int = 2, n_min_views_filter_adjust: int = 1):
    """Runs the reference's dense schedule for `view_ids` on a loaded scene (engine.scene_load / scene_set_view)."""
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

"""-m gpu, collected last (after tests/test_zz_gpu_narrow_speculation.py): ViewData::ScaleImage through the scene front end's bookkeeping and densify.compute_depth_maps --
resampled copies of two neighbours (0.8x INTER_AREA, 1.25x INTER_CUBIC) in extra source-only slots, handed their images' previous-round depth maps at the round boundary;
every view's final map equals the oracle given exactly those inputs (body: tests/test_gpu_patchmatch.py::resampled_neighbour_copies_through_the_driver).

Its own file, sorted after the others: the host logic was written after the round's GPU budget was spent.  The engine mechanisms it drives (sized source views, installed
source depth maps) are device-verified by test_mixed_resolution_neighbours_parity, and this very case -- same sizes -- passes under the CPU emulator
(tests/test_emu_kernels.py runs it at 80x60; profiles/r05_emu_scale_image_160x120.log at the device test's 160x120); a surprise here must not keep `pytest -x` from
running the rest of the suite first.  The same holds for the second case, densify.dense_reconstruction end to end on the pipeline-test scene (emulator: 80x60 in the CPU
suite, the device case's 160x120 in profiles/r05_emu_dense_reconstruction_160x120.log)."""
import pytest

pytestmark = pytest.mark.gpu


def test_resampled_neighbour_copies_through_the_driver():
    from tests import test_gpu_patchmatch as g
    g.resampled_neighbour_copies_through_the_driver(160, 120)


def dense_reconstruction_chain(tmp_path, level=2):
    """densify.dense_reconstruction (Scene::DenseReconstruction for the PatchMatch path) on the reference's pipeline-test scene at 1 / 2^level of its resolution: option
    table -> views -> all depth maps (file contract on) -> fusion -> <scene>_dense.mvs.  The fused cloud equals the sequential oracle's FuseDepthMaps on the engine's own
    final maps with the same order and colours, and the archive reads back to it."""
    import os
    import numpy as np
    from openmvs_amd import densify, dmap, mvsi, optdense
    from openmvs_amd.patchmatch import PatchMatchHIP
    from oracle import pyoracle as po
    from tests import fuse_cases as fcs
    scene = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "scene", "scene.mvs")
    opt = optdense.defaults()
    opt.nResolutionLevel = level; opt.nMinResolution = 40; opt.nNumViews = 8; opt.nEstimateNormals = 2; opt.nSpeckleSize = 20
    out, dm = str(tmp_path / "scene_dense.mvs"), str(tmp_path / "dmaps")
    e = PatchMatchHIP(0)
    sv, cloud = densify.dense_reconstruction(e, scene, out, opt, seed=3, dmap_dir=dm)
    assert sv.ids == [0, 1, 2, 3] and not sv.alias_of and sorted(os.listdir(dm)) == ["depth%04d.dmap" % i for i in range(4)]
    final = {v: e.scene_get_maps(v) for v in sv.ids}
    lo = 0.3 if level <= 2 else 0.02                                      # (at 80x60 the speckle and cross-view filters leave a few per cent: too few pixels for this scene's texture)
    assert all(lo < (final[v][0] > 0).mean() < 0.98 for v in sv.ids)
    f2 = dmap.load(os.path.join(dm, "depth0002.dmap"))
    assert np.array_equal(f2["depth_map"], final[2][0]) and np.array_equal(f2["confidence_map"], final[2][2]) and f2["neighbor_view_ids"] == [int(i) for i in sv.neighbors[2]]
    order = densify.fuse_order(sv)
    ref = po.fuse_depth_maps([final[v][0] for v in sv.ids], [final[v][1] for v in sv.ids], [final[v][2] for v in sv.ids], sv.bgr, sv.K, sv.R, sv.C,
                             [list(x) for x in sv.neighbors], order=order)
    fcs.same_cloud(cloud, ref, "dense_reconstruction: fused cloud")
    assert cloud["nPoints"] > (0.3 if level <= 2 else 0.03) * sv.width * sv.height
    back = mvsi.load(out)
    assert np.array_equal(back.vertices, cloud["points"]) and np.array_equal(back.vertices_color, cloud["colors"]) and np.array_equal(back.vertices_normal, cloud["normals"])
    assert np.array_equal(back.vertex_views["image_id"], cloud["views"]) and np.array_equal(back.vertex_views["confidence"], cloud["weights"])
    e.close()
    # fusion mode 1: depth maps only; a second pass over the same directory reads every map back instead of estimating it in the photometric pass
    e = PatchMatchHIP(0)
    sv1, none = densify.dense_reconstruction(e, scene, None, opt, seed=3, fusion_mode=1, dmap_dir=dm)
    assert none is None and sv1.ids == sv.ids
    e.close()


def test_dense_reconstruction_chain(tmp_path):
    dense_reconstruction_chain(tmp_path, level=2)

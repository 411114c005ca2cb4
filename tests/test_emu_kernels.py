"""The product's kernels and host engine, compiled for the host against the wave64 emulator (tests/cpp/hipemu, tests/emu.py) and run through the very
same C ABI and Python mirrors as on the device, against the oracle.  The checks are the bodies of the `-m gpu` parity tests, called with engines
that sit on the emulated libraries -- same inputs, same assertions, smaller selection (the emulator is ~1000x slower than the device).

What this establishes in the CPU suite: the kernels' arithmetic AND their launch geometry, index math, cross-lane exchanges (shuffles, ballots,
DPP), LDS staging, atomics, and the host orchestration between launches are bit-exact with the oracle, and no launch writes outside its buffers
(guard bands around every device allocation).  What it cannot establish is listed in the emulator's header; the device run remains the gate."""
import os
import subprocess
import sys

import numpy as np
import pytest

from openmvs_amd import patchmatch, sgm
from tests import emu


@pytest.fixture(scope="module")
def pm_emulated():
    with emu.emulated(patchmatch, "PMHIP_LIB", "libpmhip_emu.so"):
        yield


@pytest.fixture(scope="module")
def sgm_emulated():
    with emu.emulated(sgm, "SGMHIP_LIB", "libsgmhip_emu.so"):
        yield


@pytest.fixture(scope="module")
def engine(pm_emulated):
    e = patchmatch.PatchMatchHIP(0)
    e.Init(False)
    yield e
    e.close()


@pytest.fixture(scope="module")
def matcher(sgm_emulated):
    m = sgm.SemiGlobalMatcherHIP(0)
    yield m
    m.close()


@pytest.fixture(scope="module")
def tiny_scene():
    from openmvs_amd import synth
    return synth.make_scene(5, 96, 72, n_src=4)


# ---- PatchMatch: estimator, filters, fusion ------------------------------------------------------------------------------------------------------
def test_estimator_single_view(engine, small_scene):
    from tests import test_gpu_patchmatch as g
    g.test_device_resampling_matches_oracle(engine)
    g.test_single_view_photometric_parity_N4(engine, small_scene, 2)          # 3 pyramid levels, 4 source views (G = 4 lanes per pixel)
    g.test_initial_estimate_is_honoured(engine, small_scene)
    g.test_single_call_with_ignore_mask(engine, small_scene)
    for k in (0, 2, 3, 5):                                                   # non-default OPTDENSE values (all six sets pass; four run here)
        g.test_non_default_options_parity(engine, small_scene, k)


def test_estimator_group_sizes_and_geometric_pass(engine, nine_scene):
    from tests import test_gpu_patchmatch as g
    g.test_single_view_parity_N8_and_N1(engine, nine_scene)                   # G = 8 and G = 1
    g.test_geometric_round_parity_and_golden(engine)
    g.test_many_source_views_parity(engine)                                   # G = 16 (9 .. 16 sources) and partial groups


@pytest.mark.parametrize("lanes", [4])   # (8 lanes per pixel: device only)
def test_estimator_views_per_lane(pm_emulated, nine_scene, small_scene, lanes):
    from tests import test_gpu_patchmatch as g
    g.test_views_per_lane_mappings_parity(nine_scene, small_scene, lanes)    # (4,2), (4,1), (4,4): several source views per lane


@pytest.mark.parametrize("variant", ["quad_pointer"])   # (the default addressing -- the level's quad buffer -- is what every other case of this module runs)
def test_estimator_sweep_kernel_variants(pm_emulated, nine_scene, small_scene, variant):
    from tests import test_gpu_patchmatch as g
    g.test_sweep_kernel_variants_parity(nine_scene, small_scene, variant, quick=True)


def test_estimator_tuning_through_the_abi(pm_emulated, nine_scene):
    from tests import test_gpu_patchmatch as g
    g.test_tuning_through_the_abi(nine_scene, views=[4, 6])


def test_estimator_wide_latency_mode(pm_emulated, nine_scene, small_scene):
    from tests import test_gpu_patchmatch as g
    g.test_wide_latency_mode_parity(nine_scene, small_scene, quick=True)     # one wave per pixel, eight hypotheses per round (the whole case passes too: 260 s)


@pytest.mark.parametrize("hyps", [2])   # (the four-wide instantiation runs on the device: tests/test_zz_gpu_narrow_speculation.py; both pass here too)
def test_estimator_narrower_speculation(pm_emulated, nine_scene, small_scene, hyps):
    """pm_sweep_widen_kernel: two hypotheses per round, four pixels per wave -- the engine's default from 3 to 25 reference views per batch."""
    from tests import test_gpu_patchmatch as g
    g.test_wide_latency_mode_parity(nine_scene, small_scene, quick=True, hyps=str(hyps))


def test_estimator_mixed_resolution_neighbours_wide_kernel(pm_emulated):
    from tests import test_gpu_patchmatch as g
    g.test_mixed_resolution_neighbours_parity_both_kernels(16, 80, 60)     # (quarter of the pixels of the device case: the regular-kernel run below has the full size)


def test_estimator_reference_views_of_different_sizes(pm_emulated):
    from tests import test_gpu_patchmatch as g
    g.test_reference_views_of_different_sizes(96, 72)          # three size classes in one call, geometric round, per-map filters, cross-view filter
    g.test_ignore_mask_on_a_view_with_its_own_size(96, 72)
    g.test_sized_view_api_edges(64, 48)


def test_estimator_resampled_neighbour_copies_through_the_driver(pm_emulated):
    from tests import test_gpu_patchmatch as g
    g.resampled_neighbour_copies_through_the_driver(80, 60)           # ViewData::ScaleImage: copies in extra slots, handed their images' depth maps at the round boundary


@pytest.mark.parametrize("kernel", ["sweep2", "speculative"])
def test_estimator_tiled_sweeps(pm_emulated, nine_scene, small_scene, kernel):
    from tests import test_gpu_patchmatch as g
    g.test_tiled_sweeps_equal_the_tiled_oracle(nine_scene, small_scene, kernel, tiles=((24, 16),) if kernel == "sweep2" else ((9, 40),))


def test_fusion_with_a_source_only_slot_without_colour(pm_emulated, small_scene):
    from tests import test_gpu_fuse as g
    g.test_fuse_with_a_source_only_slot_that_has_no_colour(small_scene)


def test_dense_reconstruction_chain_on_the_pipeline_test_scene(pm_emulated, tmp_path):
    from tests import test_zz_gpu_scale_image as z
    z.dense_reconstruction_chain(tmp_path, level=3)             # 80x60: option table -> views -> depth maps (file contract) -> fusion -> scene_dense.mvs


def test_estimator_mixed_resolution_neighbours(engine):
    from tests import test_gpu_patchmatch as g
    g.test_mixed_resolution_neighbours_parity(engine)                        # sources at 0.8x / 1.25x, cameraDepthMap of another size


def test_estimator_portrait_image(engine, W=30, H=76):
    """An image more than twice as tall as wide: the folded anti-diagonal-major reference image (PMTask::refS, texel (u,v) at ((u+v) mod w)*h + v) wraps its rows more than
    once there (anti-diagonals d, d + w and d + 2w share a row), and the sweeps' diagonals are long in y.  Photometric pass over two levels, through the one-call boundary.
    (Written after the round's last device run: it lives here, in the emulator suite, and joins the gpu suite once it has been seen green on a device.)"""
    import numpy as np
    from openmvs_amd import synth
    from openmvs_amd.patchmatch import default_params
    from tests import test_gpu_patchmatch as g
    sc = synth.make_scene(4, W, H, n_src=3)
    engine.Init(False)
    p = default_params(seed=6, nSubResolutionLevels=1)
    for ref in (0, 2):
        ids = [ref] + list(sc.neighbors[ref])
        d, n, c = engine.EstimateDepthMap(sc.gray, sc.K, sc.R, sc.C, ids, sc.dmin[ref], sc.dmax[ref], params=p)
        od, on, oc = g._oracle(sc, ref, 6, nSubResolutionLevels=1)
        g._same(d, od, "portrait depth v%d" % ref); g._same(n, on, "normal"); g._same(c, oc, "conf")
    assert (d > 0).mean() > 0.3


def test_estimator_odd_sizes(engine):
    from tests import test_gpu_patchmatch as g
    g.test_non_divisible_image_size_parity(engine)
    g.test_degenerate_inputs(engine)                                          # textureless, tiny, empty maps through filters and fusion


def test_scene_schedule_masks_and_filters(pm_emulated, small_scene):
    from tests import test_gpu_patchmatch as g
    g.test_scene_batch_full_schedule_matches_oracle(small_scene)
    g.test_ignore_mask_parity(small_scene)
    g.test_post_filter_option_sweep(small_scene)                              # the three post-filters, 17 settings (their default-setting tests
                                                                              # test_filter_depth_map_parity / _remove_small_segments_ / _gap_interpolation_ pass too)


def test_fusion(pm_emulated, small_scene, nine_scene):
    from tests import test_gpu_fuse as g
    g.test_device_fuse_is_the_sequential_fuse(small_scene, nine_scene, 0)
    g.test_device_fuse_is_the_sequential_fuse(small_scene, nine_scene, 3)
    g.test_device_fuse_reproduces_the_golden_cloud()
    g.test_device_merge_mode(small_scene)
    g.test_device_fuse_custom_order_and_errors(small_scene)
    g.test_fuse_option_sweep(small_scene)
    g.test_device_fuse_of_depth_maps_of_different_sizes(96, 72)


# ---- SGM: cost volume, 8-path aggregation, winner-take-all, the tSGM steps ------------------------------------------------------------------------
def test_sgm_match(matcher):
    from tests import test_gpu_sgm as g
    g.test_p2s_match(matcher)
    g.test_match_parity(matcher, 96, 64, "uniform", 0, 16)
    g.test_match_parity(matcher, 97, 65, "ragged", -5, 40)                    # odd valid width, ragged ranges
    for args in ((96, 64, -8, 56), (150, 90, 0, 33), (90, 160, -3, 100), (214, 77, 0, 127), (71, 12, 0, 1), (12, 140, -1, 1)):
        g.test_uniform_range_path_kernel(matcher, *args)                      # register-resident path kernel: full / odd / two entries per lane
    g.test_uniform_premise_is_checked(matcher)
    g.test_uniform_ranges_with_penalties_above_a_byte(matcher)
    g.test_pixel_table_with_entries_out_of_pixel_order(matcher)
    g.test_tiny_and_degenerate(matcher)
    g.test_sub_group_widths(matcher, 8); g.test_sub_group_widths(matcher, 32)
    g.test_range_limits(matcher, False); g.test_range_limits(matcher, True)   # 256 disparities, 257 rejected, one-row / one-column grids
    g.test_match_parity_across_long_invalid_runs(matcher)                     # whole table chunks of invalid pixels
    for args in ((96, 64, "uniform", 0, 16), (97, 65, "ragged", -5, 40), (120, 90, "ragged", 0, 200), (230, 100, "holes", -2, 12)):
        g.test_sub_group_kernels_match_parity(matcher, *args)                 # 16 lanes per pixel / pair / line (the resident tSGM loop's choice)


def test_sgm_steps(matcher):
    from tests import test_gpu_sgm_post as g
    g.test_map_steps_match_the_oracle(matcher, 131, 77, 1)
    g.test_map_steps_match_the_oracle(matcher, 9, 5, 3)
    for w, h, seed in ((64, 40, 0), (97, 53, 1)):
        g.test_range_map_matches_the_oracle(matcher, w, h, seed)
        g.test_disparity_depth_conversions_match_the_oracle(matcher, w, h, seed)
        g.test_projection_and_pair_fusion_match_the_oracle(matcher, w, h, seed)
    g.test_filter_speckles_matches_the_oracle(matcher)
    g.test_resident_fuse_equals_the_stepwise_fuse(matcher, 64, 40, 0)
    g.test_resident_fuse_equals_the_stepwise_fuse(matcher, 97, 53, 1)


def test_tsgm_loop_on_the_emulated_device(matcher, pm_emulated):
    """The whole coarse-to-fine loop with every step on the (emulated) device against the same loop on the oracle, 256x192, 3 levels.  This is the
    case that exposed the path kernel's reliance on wave lock-step (WAVE_LOCKSTEP_POINT in sgm_kernels.hip)."""
    from tests import test_gpu_sgm_post as g
    g.test_tsgm_loop_on_the_device_equals_the_loop_on_the_oracle(matcher)


def test_resident_tsgm_loop(matcher):
    """sgmhip_tsgm_match: pyramids, range tables (device scan), both matches, checks, speckles, masks and refinement without leaving the device."""
    from tests import test_gpu_sgm_post as g
    g.test_resident_tsgm_loop_equals_the_stepwise_loop(matcher, 96, 64, 6, 32)
    g.test_resident_tsgm_loop_equals_the_stepwise_loop(matcher, 200, 120, 7, 30)


def test_zz_no_lane_ever_read_a_lane_that_was_not_there(pm_emulated, sgm_emulated):
    """Runs last in this module: over everything above, no shuffle / DPP / readlane took its value from a lane that was not executing the same
    operation (on the hardware such a read returns an unspecified value -- the kernels must not depend on one)."""
    for module in (patchmatch, sgm):
        launches, fibers, exchanges, inactive = emu.counters(module)
        assert launches > 100 and fibers > 100000 and exchanges > 1000
        assert inactive == 0, "%s: %d cross-lane reads of non-participating lanes" % (module.__name__, inactive)


@pytest.mark.parametrize("order", ["reverse"])
def test_results_do_not_depend_on_the_execution_order(order):
    """The same checks with the lanes of every workgroup and the workgroups of every grid executed in another order (HIPEMU_ORDER): kernels free of
    races -- in particular the reservation rounds of the fusion, the union-find of the segment / speckle filters, the atomic splats -- give the
    same bits.  (The whole module passes under "reverse" and "stride"; a quick selection runs here under "reverse".)"""
    if os.environ.get("HIPEMU_ORDER"):
        pytest.skip("already inside a permuted run")
    env = dict(os.environ, HIPEMU_ORDER=order)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "test_fusion or test_sgm_steps or test_sgm_match or test_estimator_odd_sizes"],
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]

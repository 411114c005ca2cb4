"""-m gpu, collected last (after tests/test_zz_gpu_narrow_speculation.py): ViewData::ScaleImage through the scene front end's bookkeeping and densify.compute_depth_maps --
resampled copies of two neighbours (0.8x INTER_AREA, 1.25x INTER_CUBIC) in extra source-only slots, handed their images' previous-round depth maps at the round boundary;
every view's final map equals the oracle given exactly those inputs (body: tests/test_gpu_patchmatch.py::resampled_neighbour_copies_through_the_driver).

Its own file, sorted after the others: the host logic was written after the round's GPU budget was spent.  The engine mechanisms it drives (sized source views, installed
source depth maps) are device-verified by test_mixed_resolution_neighbours_parity, and this very case -- same sizes -- passes under the CPU emulator
(tests/test_emu_kernels.py runs it at 80x60; profiles/r05_emu_scale_image_160x120.log at the device test's 160x120); a surprise here must not keep `pytest -x` from
running the rest of the suite first."""
import pytest

pytestmark = pytest.mark.gpu


def test_resampled_neighbour_copies_through_the_driver():
    from tests import test_gpu_patchmatch as g
    g.resampled_neighbour_copies_through_the_driver(160, 120)

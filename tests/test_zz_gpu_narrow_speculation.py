"""-m gpu, collected last: pm_sweep_widen_kernel (csrc/pm_wide_n.hip) -- the speculative sweep kernel with four or two hypotheses of a pixel per round (two or four
pixels per wave), the engine's choice for batches of 3-25 reference views since the end of round 3 -- on the cases of the eight-wide kernel's parity test: 8 / 4 / 1-3
sources, pyramid, geometric round, ignore masks, iteration budgets above and below a round, a nine-view batch.  Same bits as the oracle.

Its own file, sorted after the others: the kernel was written after the round's GPU budget was spent (what it has run on the device is the full-schedule identity check of
tools/small_batch_probe.py at 1920x1080, profiles/r03_small_batches_call24_26_narrower_speculation.log; these cases pass under the CPU emulator), so a surprise here must not
keep `pytest -x` from running the rest of the suite first."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hyps", ["2", "4", None])
def test_narrower_speculation_parity(nine_scene, small_scene, hyps):
    from tests import test_gpu_patchmatch as g
    g.test_wide_latency_mode_parity(nine_scene, small_scene, hyps=hyps)

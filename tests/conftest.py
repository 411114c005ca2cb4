import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The engine picks a speculative sweep kernel for batches of <= 32 reference views and for the short diagonals of larger ones (PMHipTuning::wideMaxViews / widePixels: eight-wide for one
# or two views, two-wide above); nearly every test case is that small, so under the product default the regular sweep kernel -- the one that carries the benchmark -- would hardly be
# exercised.  The suite therefore pins the regular kernel on every engine a test creates through the Python bindings (pmhip_set_tuning right after pmhip_create; the library
# reads no environment), runs every case of the `engine` fixture with both choices, and tests the speculative kernels by name (test_wide_latency_mode_parity, the one-call part of
# test_config2_full_size_matches_golden, tests/test_zz_gpu_narrow_speculation.py).  Programs that drive the C ABI themselves (tests/cpp) run the product's own choice.
def _pin_the_regular_sweep_kernel():
    from openmvs_amd import patchmatch as pm
    if getattr(pm.PatchMatchHIP, "_suite_pinned", False):
        return
    created = pm.PatchMatchHIP.__init__

    def init(self, device=0):
        created(self, device)
        self.tuning(wideMaxViews=-1)
    pm.PatchMatchHIP.__init__ = init
    pm.PatchMatchHIP._suite_pinned = True


_pin_the_regular_sweep_kernel()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "pinning: the oracle against the reference's own code compiled into oracle/_ref (tests/test_ref_*.py; `pytest -m pinning`, needs /root/reference to build)")


@pytest.hookimpl(tryfirst=True)
def pytest_itemcollected(item):
    # every case of tests/test_ref_*.py pins an oracle function to the reference's text: one command runs them all (`python -m pytest tests -m pinning`)
    if os.path.basename(str(item.fspath)).startswith("test_ref_"):
        item.add_marker(pytest.mark.pinning)


def _have_gpu():
    """True when the product library finds a device (pmhip_create succeeds); no torch involved."""
    try:
        from openmvs_amd import patchmatch
        lib = patchmatch.load_library()
        import ctypes
        h = ctypes.c_void_p()
        rc = lib.pmhip_create(0, ctypes.byref(h))
        if rc == 0:
            lib.pmhip_destroy(h)
        return rc == 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a host without a GPU skips the gpu-marked cases instead of failing them.  When the gpu cases are asked for by name
    (`-m gpu`, what the driver runs on the GPU box) nothing is skipped: a box whose device the library cannot open must fail loudly."""
    mexpr = (config.getoption("-m") or "").replace(" ", "")
    if (mexpr and "gpu" in mexpr and "notgpu" not in mexpr) or not any("gpu" in it.keywords for it in items):
        return
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no MI355X visible to libpmhip (pmhip_create != 0)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def small_scene():
    from openmvs_amd import synth
    return synth.make_scene(5, 160, 120, n_src=4)


@pytest.fixture(scope="session")
def nine_scene():
    from openmvs_amd import synth
    return synth.make_scene(9, 128, 96, n_src=8)


@pytest.fixture(scope="session", params=["sweep2", "speculative"])
def engine(request):
    """The one-call engine of the GPU parity cases, once with the regular sweep kernel (pm_sweep2_kernel, what the 100-view benchmark times) and once with the product's default
    choice for small batches (the speculative kernels: eight-wide for one or two views, two-wide above) -- selected through pmhip_set_tuning, not the environment."""
    from openmvs_amd.patchmatch import PatchMatchHIP
    e = PatchMatchHIP(0)
    e.Init(False)
    e.tuning(wideMaxViews=-1 if request.param == "sweep2" else 64, wideHyps=-1)     # (-1 also switches the per-launch use of the speculative kernels off)
    yield e
    e.close()

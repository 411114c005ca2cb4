import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_scene():
    from openmvs_amd import synth
    return synth.make_scene(5, 160, 120, n_src=4)


@pytest.fixture(scope="session")
def nine_scene():
    from openmvs_amd import synth
    return synth.make_scene(9, 128, 96, n_src=8)


@pytest.fixture(scope="session")
def engine():
    from openmvs_amd.patchmatch import PatchMatchHIP
    e = PatchMatchHIP(0)
    e.Init(False)
    yield e
    e.close()

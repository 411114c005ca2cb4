import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "isolated: run the test body in a child pytest process; a crash, hang or failure there is reported as xfail here")


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    """Kernels on their first device run are exercised in a child process: a GPU fault aborts the whole process (HSA raises SIGABRT), and that must
    not take the already verified tests of the session down with it.  The child runs the very same test (same node id) with xfail disabled."""
    if pyfuncitem.get_closest_marker("isolated") is None or os.environ.get("OPENMVS_AMD_ISOLATED_CHILD"):
        return None
    env = dict(os.environ, OPENMVS_AMD_ISOLATED_CHILD="1")
    cmd = [sys.executable, "-m", "pytest", pyfuncitem.nodeid, "-x", "-q", "--runxfail", "-p", "no:cacheprovider"]
    try:
        r = subprocess.run(cmd, cwd=str(pyfuncitem.config.rootpath), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.xfail("isolated run of %s timed out" % pyfuncitem.nodeid)
    if r.returncode != 0:
        pytest.xfail("isolated run of %s exited with %d:\n%s" % (pyfuncitem.nodeid, r.returncode, r.stdout.decode(errors="replace")[-2000:]))
    return True


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

"""The C++ host-side mirror (include/PatchMatchHIP.hpp) compiles against DepthData-like types and,
on a GPU, produces the oracle's depth map when called like SceneDensify.cpp:618-623 calls pmCUDA."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def _build(tmp):
    from openmvs_amd import build
    lib = build.build_lib("libpmhip.so")
    exe = os.path.join(tmp, "adapter_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp"),
                           "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return exe


def test_adapter_compiles_and_links(tmp_path):
    exe = _build(str(tmp_path))
    assert os.path.exists(exe)
    assert subprocess.run([exe]).returncode == 2          # usage error path, no GPU touched


@pytest.mark.gpu
def test_adapter_matches_oracle(tmp_path, small_scene):
    from oracle import pyoracle as po
    sc = small_scene
    exe = _build(str(tmp_path))
    ids = [0] + list(sc.neighbors[0])
    inp = tmp_path / "in.bin"; out = tmp_path / "out.bin"
    with open(inp, "wb") as f:
        for i in ids:
            f.write(np.ascontiguousarray(sc.gray[i], np.float32).tobytes())
            f.write(np.concatenate([sc.K[i].ravel(), sc.R[i].ravel(), sc.C[i].ravel()]).astype(np.float64).tobytes())
        f.write(np.array([sc.dmin[0], sc.dmax[0]], np.float32).tobytes())
    subprocess.check_call([exe, str(sc.width), str(sc.height), str(len(ids)), str(inp), str(out)])
    raw = np.fromfile(out, np.float32); n = sc.width * sc.height
    d = raw[:n].reshape(sc.height, sc.width); nrm = raw[n:4 * n].reshape(sc.height, sc.width, 3); c = raw[4 * n:].reshape(sc.height, sc.width)
    # the adapter numbers views by position (GetID() == index in this stand-in): reference view ID 0
    views, keep = po.make_views(sc.gray, sc.K, sc.R, sc.C, ids)
    od, on, oc = po.estimate_depth_map(views, len(ids), float(sc.dmin[0]), float(sc.dmax[0]), po.default_opt(seed=7, viewID=0))
    assert np.array_equal(d, od) and np.array_equal(nrm, on) and np.array_equal(c, oc)

"""include/DenseDepthMapsHIPMulti.hpp: the C++ host of the multi-GPU split (one engine and one host thread per device, one broadcast of the image set,
all-gathers at the round boundaries only).  The orchestration is checked by running it with several engines on ONE device through the local-copy
collective against the single-engine driver -- under the CPU emulator here, on the device in the gpu suite -- and the RCCL policy is compiled and
linked against librccl (no multi-GPU box is reachable from the build container; the driver's multi-GPU run uses bench.py)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_scene(path, sc, own=None):
    """own: {view: scene} -- these views take image, colours and K from another rendering of the same scene at another size (a scene whose views differ in size)."""
    n, w, h, ns = sc.n_views, sc.width, sc.height, sc.neighbors.shape[1]
    with open(path, "wb") as f:
        f.write(np.array([-n if own else n, w, h, ns], np.int32).tobytes())
        for i in range(n):
            src = own.get(i, sc) if own else sc
            if own:
                f.write(np.array([src.width, src.height], np.int32).tobytes())
            f.write(np.ascontiguousarray(src.gray[i], np.float32).tobytes()); f.write(np.ascontiguousarray(src.bgr[i], np.uint8).tobytes())
            f.write(np.concatenate([src.K[i].ravel(), sc.R[i].ravel(), sc.C[i].ravel()]).astype(np.float64).tobytes())
            f.write(np.array([sc.dmin[i], sc.dmax[i]], np.float32).tobytes()); f.write(np.ascontiguousarray(sc.neighbors[i], np.int32).tobytes())


def test_multi_engine_host_equals_single_engine_under_the_emulator(tmp_path):
    from openmvs_amd import synth
    from tests import emu
    lib = emu.build("libpmhip_emu.so")
    exe = str(tmp_path / "dense_multi_emu")
    subprocess.check_call([emu._clang(), "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "tests", "cpp", "hipemu"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "dense_multi.cpp"), "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib)])
    sc = synth.make_scene(5, 64, 48, n_src=4)
    inp = str(tmp_path / "scene.bin"); _write_scene(inp, sc)
    for engines in (2, 3):
        r = subprocess.run([exe, inp, str(engines), "31", "serial"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-400:]
        assert "engines == 1 engine" in r.stdout
    # views of different sizes (DepthMapsData::InitViews, SceneDensify.cpp:306-459): two of the five views smaller, one larger; blocks of 3 + 2 views, so maps of
    # their own size cross engines at the round boundaries, before the filter and on the way to the fusing engine
    small = synth.make_scene(5, 48, 36, n_src=4); big = synth.make_scene(5, 80, 60, n_src=4)
    inp2 = str(tmp_path / "scene_mixed.bin"); _write_scene(inp2, sc, own={1: small, 3: big, 4: small})
    r = subprocess.run([exe, inp2, "2", "31", "serial"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-400:]
    assert "2 engines == 1 engine: 5 views of different sizes" in r.stdout


def test_rccl_policy_compiles_and_links(tmp_path):
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no RCCL headers")
    from openmvs_amd import build
    lib = build.build_lib("libpmhip.so")
    src = tmp_path / "rccl_link.cpp"
    src.write_text('#define PMHIP_WITH_RCCL\n#define __HIP_PLATFORM_AMD__ 1\n#include "DenseDepthMapsHIPMulti.hpp"\n'
                   'int main(int argc, char**) { if (argc > 100) { MVS::DenseDepthMapsHIPMulti m(std::vector<int>{0, 1}); MVS::DenseDepthMapsHIP::PointCloud pc; m.ComputeDepthMaps(); m.FuseDepthMaps(pc); } return 0; }\n')
    exe = str(tmp_path / "rccl_link")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", str(src), "-o", exe, lib,
                           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl"])


@pytest.mark.gpu
def test_multi_engine_host_equals_single_engine_on_the_device(tmp_path, small_scene):
    from openmvs_amd import build
    lib = build.build_lib("libpmhip.so")
    exe = str(tmp_path / "dense_multi")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-D__HIP_PLATFORM_AMD__=1", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "dense_multi.cpp"),
                           "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    inp = str(tmp_path / "scene.bin"); _write_scene(inp, small_scene)
    r = subprocess.run([exe, inp, "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-400:]
    from openmvs_amd import synth
    sc = synth.make_scene(5, 128, 96, n_src=4); small = synth.make_scene(5, 96, 72, n_src=4); big = synth.make_scene(5, 160, 120, n_src=4)
    inp2 = str(tmp_path / "scene_mixed.bin"); _write_scene(inp2, sc, own={1: small, 3: big, 4: small})
    for engines in ("2", "3"):
        r = subprocess.run([exe, inp2, engines], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-400:]
        assert "views of different sizes" in r.stdout


@pytest.mark.gpu
def test_rccl_policy_runs_on_one_device(tmp_path, small_scene):
    """The RCCL policy of the multi-device host with the one rank a single-GPU box can give it: ncclCommInitAll, the image broadcast and the round-boundary
    exchanges all execute (as grouped ncclBroadcast calls on the engine's stream), and the results equal the single-engine driver's."""
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no RCCL headers")
    from openmvs_amd import build
    lib = build.build_lib("libpmhip.so")
    exe = str(tmp_path / "dense_multi_rccl")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-DPMHIP_WITH_RCCL", "-D__HIP_PLATFORM_AMD__=1", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "dense_multi.cpp"), "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-lamdhip64", "-lrccl"])
    inp = str(tmp_path / "scene.bin"); _write_scene(inp, small_scene)
    r = subprocess.run([exe, inp, "1", "31", "rccl"], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-600:]
    assert "1 engines == 1 engine" in r.stdout

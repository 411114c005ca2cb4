"""Test infrastructure: the product's HIP sources compiled for the host against tests/cpp/hipemu (a wave64 fiber emulator, see its header) and loaded
in place of libpmhip.so / libsgmhip.so, so that the kernels, their launch geometry, cross-lane traffic and atomics execute in the CPU test-suite and are
compared with the oracle there.  Only tests use this module; the product knows nothing about it and still fails loudly without a GPU.

The emulated libraries are the same translation units (pm_engine.hip, sgm_engine.hip with everything they include), built with the host compiler of
the ROCm LLVM (clang++ -x c++) and the product's floating-point contract (-ffp-contract=off)."""
import contextlib
import os
import shutil
import subprocess

import pytest

from openmvs_amd import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "cpp", "hipemu")
OUT = os.path.join(EMU, "_build")
CSRC = os.path.join(os.path.dirname(HERE), "openmvs_amd", "csrc")
LIBS = {"libpmhip_emu.so": "libpmhip.so", "libsgmhip_emu.so": "libsgmhip.so"}


def _clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("amdclang++"), shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    return None


def build(name: str) -> str:
    """Build (if stale) and return the path of an emulated library; skips the calling test when no clang++ is available."""
    srcs, deps = _build.LIBS[LIBS[name]]
    srcs_abs = [os.path.join(CSRC, s) for s in srcs]
    out = os.path.join(OUT, name)
    hdr = os.path.join(EMU, "hip", "hip_runtime.h")
    if not _build._stale(out, srcs_abs + [os.path.normpath(os.path.join(CSRC, d)) for d in deps] + [hdr]):
        return out
    cxx = _clang()
    if cxx is None:
        pytest.skip("no clang++ to build the emulated libraries")
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call([cxx, "-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value", "-Wno-unknown-attributes",
                           "-I", EMU] + srcs_abs + ["-o", out + ".tmp"], cwd=CSRC)
    os.replace(out + ".tmp", out)
    return out


@contextlib.contextmanager
def emulated(module, env_var, name):
    """Make `module.load_library()` (openmvs_amd.patchmatch / openmvs_amd.sgm) return the emulated library inside the block."""
    path = build(name)
    saved_lib, saved_env, saved_flag = module._LIB, os.environ.get(env_var), os.environ.get("OPENMVS_AMD_TEST_EMULATOR")
    module._LIB = None
    os.environ[env_var] = path
    os.environ["OPENMVS_AMD_TEST_EMULATOR"] = "1"                   # the product refuses emulated builds without it
    try:
        module.load_library()
        yield path
    finally:
        module._LIB = saved_lib
        for k, v in ((env_var, saved_env), ("OPENMVS_AMD_TEST_EMULATOR", saved_flag)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def counters(module):
    """(launches, fibers run, cross-lane exchanges, reads of non-participating lanes) of the emulated library behind `module`."""
    import ctypes
    out = (ctypes.c_uint64 * 4)()
    module.load_library().hipemu_counters(out)
    return tuple(int(v) for v in out)

"""Build the HIP shared libraries of the engine in-tree (hipcc, gfx950 only).

`python -m openmvs_amd.build` or `openmvs_amd.build.build_all()`.  hipcc cross-compiles without a
GPU; the resulting .so files sit next to this package (git-ignored, shipped to the GPU box).
-ffp-contract=off and correctly rounded divide/sqrt are part of the numerical contract with the
CPU oracle (see csrc/pm_math.h), not tuning knobs.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-value", "-Wno-pass-failed"]

LIBS = {
    "libpmhip.so": (["pm_engine.hip"], ["pm_kernels.hip", "pm_band.hip", "pm_wide_n.hip", "pm_filter.hip", "pm_fuse.hip", "pm_fuse.h", "pm_math.h", "../../include/pmhip.h"]),
    "libsgmhip.so": (["sgm_engine.hip"], ["sgm_kernels.hip", "sgm_kernels_sub.hip", "sgm_tsgm.hip", "sgm_post.hip", "sgm_post.h", "pm_math.h", "../../include/sgmhip.h"]),
}


# per-library extra flags.  libsgmhip: the SLP vectoriser pairs the per-tap multiplies of sgm_cost_px_kernel into v_pk_mul_f32, which costs it ~60 more
# VGPRs (copies into aligned register pairs) and pushes it into scratch; the packed form is no faster per flop on this part.
# libpmhip: the vectoriser pairs the bilinear and accumulation arithmetic of the tap rows into v_pk_mul_f32 / v_pk_add_f32 (no faster than two plain ones here) at the
# price of ~25 v_mov per row to line the operands up in register pairs; the one place a packed instruction pays (the FMAs of the perspective division) is written
# out as a vector type in pm_math.h.
LIB_FLAGS = {"libsgmhip.so": ["-fno-slp-vectorize"], "libpmhip.so": ["-fno-slp-vectorize"]}

HOST_LIBS = {"libdmapio.so": (["dmap_io.cpp"], ["../../include/dmapio.h"]),          # plain C++ (g++), no GPU code
             "libmvsfront.so": (["mvs_front.cpp", "opt_dense.cpp"], ["sml_text.h", "../../include/mvsfront.h", "../../include/optdense.h", "../../include/pmhip.h"])}


def build_host_lib(name: str, force: bool = False) -> str | None:
    srcs, deps = HOST_LIBS[name]
    srcs_abs = [os.path.join(_CSRC, s) for s in srcs]
    out = lib_path(name)
    if not force and not _stale(out, srcs_abs + [os.path.normpath(os.path.join(_CSRC, d)) for d in deps]):
        return out
    cxx = shutil.which("g++")
    if cxx is None:
        if os.path.exists(out):
            return out
        raise RuntimeError("g++ not found and %s is not built" % name)
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off"] + srcs_abs + ["-o", out], cwd=_CSRC)
    return out


def lib_path(name: str) -> str:
    return os.path.join(_HERE, name)


def _stale(out: str, deps: list[str]) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, out_name: str, extra_flags: list[str]) -> str:
    """Experiment helper: build `name` with extra -D flags into openmvs_amd/<out_name>."""
    srcs, _ = LIBS[name]
    out = lib_path(out_name)
    subprocess.check_call([HIPCC] + FLAGS + LIB_FLAGS.get(name, []) + list(extra_flags) + [os.path.join(_CSRC, s) for s in srcs] + ["-o", out], cwd=_CSRC)
    return out


def build_lib(name: str, force: bool = False, verbose: bool = False) -> str | None:
    srcs, deps = LIBS[name]
    srcs_abs = [os.path.join(_CSRC, s) for s in srcs]
    if not all(os.path.exists(s) for s in srcs_abs):
        return None
    out = lib_path(name)
    if not force and not _stale(out, srcs_abs + [os.path.normpath(os.path.join(_CSRC, d)) for d in deps]):
        return out
    if not os.path.exists(HIPCC):
        if os.path.exists(out):
            return out  # GPU box without a compiler in PATH: use the shipped binary
        raise RuntimeError("hipcc not found and %s is not built" % name)
    cmd = [HIPCC] + FLAGS + LIB_FLAGS.get(name, []) + srcs_abs + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=_CSRC)
    return out


def build_all(force: bool = False, verbose: bool = False) -> list[str]:
    return [p for p in (build_lib(n, force, verbose) for n in LIBS) if p] + [build_host_lib(n, force) for n in HOST_LIBS]


if __name__ == "__main__":
    for p in build_all(force=True, verbose=True):
        print("built", p)

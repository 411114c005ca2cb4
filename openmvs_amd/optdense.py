"""The reference's dense-reconstruction options (namespace OPTDENSE, libs/MVS/DepthMap.cpp:67-114) and their configuration file
(`DensifyPointCloud --dense-config-file`, apps/DensifyPointCloud/DensifyPointCloud.cpp:236-255): ctypes binding of include/optdense.h (libmvsfront.so).

    opt = optdense.load("dense.ini")            # init() + Load + update(); a missing file gives the defaults (and `opt.loaded` False), as in the reference
    opt.nNumViews = 8                           # what the command-line flags do after update()
    p = opt.params(seed=1)                      # -> PMHipParams for the estimator
    front = opt.front_options()                 # -> MVSFOptions for libmvsfront's view selection
    dense = opt.dense_options()                 # -> views.DenseOptions for the numpy front end
"""
from __future__ import annotations

import ctypes as C

from . import mvsfront as _front

_U, _I, _F = C.c_uint32, C.c_int32, C.c_float


class OptDense(C.Structure):
    """MVSFOptDense: the reference's variables, in the reference's order, under the reference's names (bool options are 0 / 1)."""
    _fields_ = ([(n, _U) for n in ("nResolutionLevel", "nMaxResolution", "nMinResolution", "nSubResolutionLevels", "nMinViews", "nMaxViews", "nMinViewsFuse", "nMinViewsFilter",
                                   "nMinViewsFilterAdjust", "nMinViewsTrustPoint", "nNumViews", "nPointInsideROI")]
                + [(n, _I) for n in ("bFilterAdjust", "bAddCorners", "bInitSparse", "bRemoveDmaps")]
                + [(n, _F) for n in ("fViewMinScore", "fViewMinScoreRatio", "fMinArea", "fMinAngle", "fOptimAngle", "fMaxAngle", "fDescriptorMinMagnitudeThreshold",
                                     "fDepthDiffThreshold", "fNormalDiffThreshold", "fPairwiseMul", "fOptimizerEps")]
                + [("nOptimizerMaxIters", _I), ("nSpeckleSize", _U), ("nIpolGapSize", _U), ("nIgnoreMaskLabel", _I), ("nOptimize", _U), ("nEstimateColors", _U),
                   ("nEstimateNormals", _U), ("fNCCThresholdKeep", _F), ("nEstimationIters", _U), ("nEstimationGeometricIters", _U), ("fEstimationGeometricWeight", _F),
                   ("nRandomIters", _U), ("nRandomMaxScale", _U)]
                + [(n, _F) for n in ("fRandomDepthRatio", "fRandomAngle1Range", "fRandomAngle2Range", "fRandomSmoothDepth", "fRandomSmoothNormal", "fRandomSmoothBonus")])
    loaded = True          # False when load() could not read the file (the reference's bValidConfig)
    unknown = 0            # entries of the file whose title is not an option

    def set(self, title: str, value) -> None:
        """One option by its title, from text (`istream >> value`, like OPTDENSE::update)."""
        if _lib().mvsf_optdense_set(C.byref(self), title.encode(), str(value).encode()) != 0:
            raise KeyError(title)

    def get(self, title: str) -> str:
        buf = C.create_string_buffer(64)
        if _lib().mvsf_optdense_get(C.byref(self), title.encode(), buf, 64) != 0:
            raise KeyError(title)
        return buf.value.decode()

    def save(self, path: str) -> None:
        if _lib().mvsf_optdense_save(path.encode(), C.byref(self)) != 0:
            raise OSError("cannot write %s" % path)

    def front_options(self) -> _front.MVSFOptions:
        f = _front.MVSFOptions()
        _lib().mvsf_optdense_front(C.byref(self), C.byref(f))
        return f

    def params(self, seed: int = 0):
        """PMHipParams for the estimator (the seed is ours: the reference seeds from random_device)."""
        from .patchmatch import PMHipParams
        p = PMHipParams()
        _lib().mvsf_optdense_estimator(C.byref(self), C.byref(p))       # every field but the seed
        p.seed = seed
        return p

    def dense_options(self):
        from .views import DenseOptions
        d = DenseOptions()
        for k in vars(d):
            v = getattr(self, k)
            setattr(d, k, bool(v) if k.startswith("b") else v)
        return d

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


_READY = False


def _lib():
    global _READY
    lib = _front.load_library()
    if not _READY:
        lib.mvsf_optdense_init.restype = None; lib.mvsf_optdense_front.restype = None; lib.mvsf_optdense_estimator.restype = None
        for n in EXPORTS:
            getattr(lib, n)
        _READY = True
    return lib


EXPORTS = ["mvsf_optdense_count", "mvsf_optdense_describe", "mvsf_optdense_init", "mvsf_optdense_set", "mvsf_optdense_get", "mvsf_optdense_load", "mvsf_optdense_save",
           "mvsf_optdense_front", "mvsf_optdense_estimator"]


def defaults() -> OptDense:
    """OPTDENSE::init()."""
    o = OptDense()
    _lib().mvsf_optdense_init(C.byref(o))
    return o


def load(path: str) -> OptDense:
    """OPTDENSE::init(); oConfig.Load(path); OPTDENSE::update().  A file that cannot be read leaves the defaults (`.loaded` False): DensifyPointCloud then writes the
    table to that path (`opt.save(path)`), DensifyPointCloud.cpp:253-254."""
    o = OptDense(); unknown = C.c_int()
    rc = _lib().mvsf_optdense_load(path.encode(), C.byref(o), C.byref(unknown))
    if rc not in (0, -2):
        raise ValueError("mvsf_optdense_load: %d" % rc)
    o.loaded = rc == 0
    o.unknown = unknown.value
    return o


def table():
    """[(variable, title, type, default text)] -- the option list of libs/MVS/DepthMap.cpp:69-114."""
    out = []
    for i in range(_lib().mvsf_optdense_count()):
        a, b, c, d = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_char_p()
        assert _lib().mvsf_optdense_describe(i, C.byref(a), C.byref(b), C.byref(c), C.byref(d)) == 0
        out.append((a.value.decode(), b.value.decode(), c.value.decode(), d.value.decode()))
    return out

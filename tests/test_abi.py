"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/pmhip.h declares,
and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b((?:pmhip|sgmhip)_[a-z_0-9]+)\s*\(", src)))


def test_pmhip_exports_every_declared_symbol():
    from openmvs_amd import patchmatch
    lib = patchmatch.load_library()
    names = _declared("pmhip.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(patchmatch.EXPORTS) == names


def test_sgmhip_exports_every_declared_symbol():
    from openmvs_amd import sgm
    lib = sgm.load_library()
    names = _declared("sgmhip.h")
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(sgm.EXPORTS) == names
    assert sgm.PIXEL_DTYPE.itemsize == 16 and C.sizeof(sgm.SGMHipStats) == 40
    # GenerateP2s needs no GPU: P2*(1+alpha) .. P2
    p = sgm.generate_p2s()
    assert p[0] == 60 and p[255] == 4


def test_struct_sizes_match_header():
    from openmvs_amd import patchmatch as pm
    assert C.sizeof(pm.PMHipParams) == 4 * 4 + 9 * 4 + 4
    assert C.sizeof(pm.PMHipView) == 8 + 8 + 21 * 8 + 8 + 21 * 8 + 16   # id, dw, dh + 4 bytes of tail padding
    assert C.sizeof(pm.PMHipKernelStats) == 64 and C.sizeof(pm.PMHipTuning) == 8 * 4


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from openmvs_amd.patchmatch import PatchMatchError, PatchMatchHIP
    with pytest.raises(PatchMatchError):
        PatchMatchHIP(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "openmvs_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                s = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyoracle" not in s and "pm_oracle" not in s.replace("oracle/pm_oracle.cpp", "") and "libpm_oracle" not in s, f

"""The resource figures DESIGN.md section 4.2 quotes for the dominant kernel, as the compiler reports them for gfx950 (hipcc cross-compiles: no GPU needed).
They decide how many waves a CU holds, which is what bounds pm_sweep_kernel; a change that silently costs a wave per SIMD or spills in bulk fails here."""
import os, shutil, sys
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_sweep_kernel_keeps_its_residency_budget():
    import kernel_resources as kr
    r = kr.resources()
    sweeps = {k: v for k, v in r.items() if "pm_sweep_kernel" in k}
    assert len(sweeps) == 22                      # 11 (lanes per pixel, views per lane) mappings x photometric / geometric
    for k, v in sweeps.items():
        lanes = int(k.split("pm_sweep_kernelILi")[1].split("E")[0])
        vpl = int(k.split("pm_sweep_kernelILi")[1].split("ELi")[1].split("E")[0])
        if lanes in (4, 8) and vpl == 1:                       # the mappings of 3..8 source views: three waves per SIMD and eleven one-wave workgroups per CU
            assert v["occupancy"] >= 3 and v["vgpr"] <= 168, (k, v)
            assert v["lds"] <= 15104, (k, v)   # windows 20 x (pixels per wave + 10) per view, weights, and the per-view constants (832 B; twice that in the geometric pass)
            assert v["scratch"] <= 60, (k, v)   # a few spilled dwords (round-1 kernel: 28 at 8 lanes per pixel; now 44 photometric, 60 geometric)
        else:
            assert v["occupancy"] >= (1 if vpl >= 4 else 2), (k, v)
        assert v["agpr"] == 0
    # the default sweep kernel since round 3 (pm_band.hip: visit state in LDS, quad images, no source windows): no scratch at all and little LDS, so
    # registers alone decide the residency (3 waves per SIMD; a 4-wave build was measured 3 % slower, profiles/r03_variants_call6_sweep2.log)
    sweep2 = {k: v for k, v in r.items() if "pm_sweep2_kernel" in k}
    assert len(sweep2) == 12
    for k, v in sweep2.items():
        assert v["occupancy"] >= 3 and v["vgpr"] <= 168 and v["scratch"] == 0 and v["agpr"] == 0, (k, v)
        assert v["lds"] <= 11008, (k, v)
    # the speculative kernels: eight-wide (one or two views) and the two- / four-wide template (3-25 views): three waves per SIMD, a few spilled dwords at most
    wide = {k: v for k, v in r.items() if "pm_sweep_wide_kernel" in k or "pm_sweep_widen_kernel" in k}
    assert len(wide) == 6
    for k, v in wide.items():
        assert v["occupancy"] >= 3 and v["vgpr"] <= 168 and v["scratch"] <= 48 and v["lds"] <= 4096, (k, v)
    band = {k: v for k, v in r.items() if "pm_band_kernel" in k}
    assert len(band) == 12
    for k, v in band.items():
        assert v["occupancy"] >= 3 and v["scratch"] <= 200, (k, v)     # spills of loop-invariant values at the head of a step only (checked in the ISA)

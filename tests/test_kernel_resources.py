"""The resource figures DESIGN.md section 4.2 quotes for the sweep kernels, as the compiler reports them for gfx950 (hipcc cross-compiles: no GPU needed).
Registers alone decide how many waves a SIMD holds here (no scratch, little LDS); a change that silently costs a wave per SIMD or starts spilling fails here."""
import os, shutil, sys
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_sweep_kernels_keep_their_residency_budget():
    import kernel_resources as kr
    r = kr.resources()
    # pm_sweep2_kernel<G, VPL, GEO, BUF> (pm_band.hip: visit state in LDS, branch-free optimistic tap rows as a two-deep pipeline): six (lanes per pixel, views per
    # lane) mappings x photometric / geometric x quad buffer / view pointer.  Three waves per SIMD (<= 168 VGPRs; the un-pipelined rows fitted four and were 7 % slower),
    # no scratch.
    sweep2 = {k: v for k, v in r.items() if "pm_sweep2_kernel" in k}
    assert len(sweep2) == 36      # + the tiled instantiations of the quad-buffer kernels (opt-in tiled sweeps)
    for k, v in sweep2.items():
        assert v["occupancy"] >= 3 and v["vgpr"] <= 168 and v["scratch"] == 0 and v["agpr"] == 0, (k, v)
        assert v["lds"] <= 163840 // 12, (k, v)                        # three one-wave workgroups per SIMD fit a CU's 160 KB of LDS
    # the instantiation the 100-view benchmark times on its photometric sweeps -- 4 lanes per pixel, 2 views per lane, quad buffer -- fits FOUR waves per SIMD (<= 128 VGPRs).
    # It is a narrow fit: a guarded-redo path with 16-byte loads cost it 9 registers, the fourth wave and 1.6 % of the photometric pass (profiles/r05_call4_ab_100.log)
    timed = [v for k, v in sweep2.items() if "ILi4ELi2ELb0ELb1ELb0E" in k]
    assert len(timed) == 1 and timed[0]["occupancy"] >= 4 and timed[0]["vgpr"] <= 128, timed
    assert timed[0]["lds"] <= 163840 // 16, timed                      # ... and sixteen of its workgroups a CU's LDS (round 6: the visit's prepared draws live there)
    # the speculative kernels: eight-wide (one or two views) and the two- / four-wide template (3-64 views): three waves per SIMD; the two-deep tap-row pipeline costs them
    # a few spilled dwords (measured worth it: 25 views 33.9 -> 37.5 Mpix/s, 13 views 23.6 -> 27.0; the pointer-path instantiations, which only batches with own-size
    # source views use, spill the most)
    wide = {k: v for k, v in r.items() if "pm_sweep_wide_kernel" in k or "pm_sweep_widen_kernel" in k}
    assert len(wide) == 16
    for k, v in wide.items():
        assert v["occupancy"] >= 3 and v["vgpr"] <= 168 and v["scratch"] <= 96 and v["lds"] <= 4096, (k, v)
    assert not any("pm_band_kernel" in k or "pm_sweep_kernel" in k for k in r)     # round 3's resident band kernel and round 2's LDS-window kernel are gone


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_no_instruction_touches_a_tap_row_register_before_its_wait():
    """The tap rows' 16-byte buffer loads are issued from inline assembly (pm_bufload5), outside the compiler's s_waitcnt bookkeeping: a copy, spill or reuse of one of
    their destination registers before the hand-written `s_waitcnt vmcnt(N)` would corrupt samples silently.  tools/isa_inflight_check.py models the vector-memory queue
    over the gfx950 assembly of every kernel with such loads; first a positive control on a hand-made sequence, then the shipped kernels."""
    import isa_inflight_check as ic
    fake = ["_Z4fakev:", "\tbuffer_load_dwordx4 v[10:13], v1, s[0:3], 0 idxen", "\tbuffer_load_dwordx4 v[14:17], v2, s[0:3], 0 idxen", "\tv_add_f32_e32 v20, v21, v22",
            "\ts_waitcnt vmcnt(1)", "\tv_mul_f32_e32 v30, v10, v11", "\tv_mov_b32_e32 v31, v15", "\ts_waitcnt vmcnt(0)", "\tv_mov_b32_e32 v32, v16", ".Lfunc_end0:"]
    got = ic.check(fake)
    assert got["_Z4fakev"][0] == 2 and got["_Z4fakev"][1] == ["v_mov_b32_e32 v31, v15"], got
    fake[3] = "\tscratch_store_dwordx4 off, v[10:13], off offset:16"              # a spill of an in-flight destination
    assert ic.check(fake)["_Z4fakev"][1] == [fake[3].strip()]                     # (the store is a queue entry itself: vmcnt(1) then covers both loads, the later v_mov is fine)
    r = ic.check()
    buf = [k for k in r if "pm_sweep" in k]
    assert len(buf) == 34, sorted(r)                                              # 24 pm_sweep2 (quad buffer; reference / tiled sweeps) + 2 eight-wide + 8 two- / four-wide instantiations
    for k, (n, bad) in r.items():
        assert n >= 15 and not bad, (k, bad[:3])


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_no_scratch_access_inside_a_tap_row_loop():
    """Whatever the register allocator spills in the sweep kernels, it does not spill inside the loops that carry the time: no scratch_load / scratch_store between the label
    and the backward branch of any loop that contains the tap rows' buffer loads."""
    import isa_mix as im
    lines = im.assembly()
    seen = 0
    for name, body in im.kernels(lines):
        if "pm_sweep" not in name:
            continue
        for loop in im.hot_loops(body):
            if len(loop) > 700:            # (an enclosing loop: the visit's trips; the tap-row loops themselves are ~400 instructions)
                continue
            seen += 1
            bad = [ln.strip() for ln in loop if ln.strip().startswith(("scratch_load", "scratch_store"))]
            assert not bad, (im.demangle(name), bad[:3])
    assert seen >= 34

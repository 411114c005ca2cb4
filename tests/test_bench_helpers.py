"""bench.py's bookkeeping that needs no GPU: the counter file it reads for `roofline.traffic` is the one tools/r05/make_traffic.py derives from the committed counter pass, and
the fields it forms from it are consistent with each other.  (The timed path of bench.py runs on the GPU box only; a crash in this glue there would cost the round its bench line.)"""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def test_committed_counter_file_is_what_the_counter_pass_gives(tmp_path):
    """profiles/traffic.json == tools/r06/make_traffic.py over profiles/r06_final_pmc (counter passes over a slice of the benchmark: one level-0 sweep of the photometric and one of the
    geometric kernels with the benchmark's 100 views resident) for the tree's kernel sources."""
    committed = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    keep = open(os.path.join(ROOT, "profiles", "traffic.json")).read()
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "r06", "make_traffic.py"), os.path.join(ROOT, "profiles", "r06_final_pmc")], stdout=subprocess.DEVNULL)
        again = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    finally:
        open(os.path.join(ROOT, "profiles", "traffic.json"), "w").write(keep)
    assert again["sweeps"] == committed["sweeps"] and again["families"] == committed["families"] and again["valu"] == committed["valu"]
    assert again["kernel_digest"] == committed["kernel_digest"], "the sweep kernels' sources changed after the counters were taken: re-run tools/r06/final.sh or expect roofline.traffic = null"
    sw = committed["sweeps"]
    assert sw["fabric_bytes_per_launch"] == 2 * sw["fetch_bytes_per_launch_raw"] + sw["write_bytes_per_launch"]
    assert abs(sw["fabric_bytes_per_step"] - sw["fabric_bytes_per_launch"] * sw["dispatches"]) <= 3 * sw["dispatches"]
    assert abs(sw["over_algorithmic"] - sw["fabric_bytes_per_launch"] / sw["algorithmic_bytes_per_launch"]) < 0.01 * sw["over_algorithmic"]
    fam = committed["families"]
    assert {"pm_sweep2_kernel", "pm_sweep_widen_kernel", "pm_init_kernel"} <= set(fam) and all(0.1 < fam[k]["valu_busy_of_simd_cycles"] < 1.0 for k in fam)


def test_roofline_fields_from_the_counter_file():
    b = _bench()
    alg = 2549567.3
    tf = b.traffic_fields(alg)
    assert tf["traffic"] and tf["traffic"] > 50 * alg and tf["traffic_measurement"]["measured"].startswith("offline")
    assert tf["traffic_measurement"]["dispatches"] == 43126                      # the benchmark's sweep launches per step (two view groups)
    v = b.valu_issue_fields(4.1, 1, 100 * 1920 * 1080 * 12)                      # live SQ sums of both kernel families of the slice, this run's wall time
    assert 0.4 < v["valu_issue"]["valu_busy_frac"] < 1.0 and "EXTRAPOLATION" not in v["valu_issue"]["note"]
    g = b.gather_issue_fields(3.8, 1)["gather_issue"]                            # the unit the sweeps run into: vector-memory wave-loads of a step against 256 units x one scattered wave-load per 64.5 cycles
    assert 0.5 < g["frac"] < 1.05 and 400 <= g["wave_loads_per_wave_visit"] < 520 and abs(g["peak"] - 256 * 2.4 / 64.5) < 0.01
    assert b.sweep_kernel_name(100, 8) == "pm_sweep2_kernel" and b.sweep_kernel_name(13, 8).startswith("pm_sweep_widen_kernel") and b.sweep_kernel_name(1, 8) == "pm_sweep_wide_kernel"
    assert b.usable_cores() >= 1

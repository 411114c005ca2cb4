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
    """profiles/traffic.json == tools/r05/make_traffic.py over profiles/r05_call6_pmc (FETCH_SIZE of every sweep launch of one benchmark step) for the tree's kernel sources."""
    committed = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    old = tmp_path / "old.json"
    old.write_text(json.dumps({"sq": {k: v for k, v in committed.get("sq_round4", {}).items() if k not in ("source", "kernel_digest")},
                               "source": committed.get("sq_round4", {}).get("source"), "kernel_digest": committed.get("sq_round4", {}).get("kernel_digest")}))
    keep = open(os.path.join(ROOT, "profiles", "traffic.json")).read()
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "r05", "make_traffic.py"), os.path.join(ROOT, "profiles", "r05_call6_pmc"), str(old)],
                              stdout=subprocess.DEVNULL)
        again = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    finally:
        open(os.path.join(ROOT, "profiles", "traffic.json"), "w").write(keep)
    assert again["sweeps"] == committed["sweeps"] and again["families"] == committed["families"]
    assert again["kernel_digest"] == committed["kernel_digest"], "the sweep kernels' sources changed after the counters were taken: re-run tools/r05/pmc_bench.sh or expect roofline.traffic = null"
    sw = committed["sweeps"]
    assert sw["fabric_bytes_per_launch"] == 2 * sw["fetch_bytes_per_launch_raw"] + sw["write_bytes_per_launch"]
    assert abs(sw["fabric_bytes_per_step"] - sw["fabric_bytes_per_launch"] * sw["dispatches"]) <= sw["dispatches"]
    assert abs(sw["over_algorithmic"] - sw["fabric_bytes_per_launch"] / sw["algorithmic_bytes_per_launch"]) < 0.01


def test_roofline_fields_from_the_counter_file():
    b = _bench()
    alg = 2549567.3
    tf = b.traffic_fields(alg)
    assert tf["traffic"] and tf["traffic"] > 50 * alg and tf["traffic_measurement"]["measured"].startswith("offline")
    assert tf["traffic_measurement"]["dispatches"] == 43126                      # the benchmark's sweep launches per step (two view groups)
    v = b.valu_issue_fields(4.3, 1, 100 * 1920 * 1080 * 12)                      # some pixel-visit count: the field must form, and say what it is
    assert not v or ("valu_busy_frac" in v["valu_issue"] and 0 < v["valu_issue"]["valu_busy_frac"] < 5)
    saved = os.environ.pop("PMHIP_WIDE", None)                                   # (tests/conftest.py pins the regular kernel for the suite; the product default is what a bench run sees)
    try:
        assert b.sweep_kernel_name(100, 8) == "pm_sweep2_kernel" and b.sweep_kernel_name(13, 8).startswith("pm_sweep_widen_kernel") and b.sweep_kernel_name(1, 8) == "pm_sweep_wide_kernel"
    finally:
        if saved is not None:
            os.environ["PMHIP_WIDE"] = saved
    assert b.usable_cores() >= 1

"""Run bench.py for several (library variant, view groups[, lanes per pixel]) settings given as lib:groups[:lanes]; prints one line each.  (tools/r04/probe_lanes.py
does the same inside one process, without bench.py's extra legs.)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
views = sys.argv[1] if len(sys.argv) > 1 else "48"
variants = [a.split(":") for a in sys.argv[2:]] or [["libpmhip.so", "2"]]
for spec in variants:
    lib, groups = spec[0], spec[1]
    lanes = spec[2] if len(spec) > 2 else ""
    widepx = spec[3] if len(spec) > 3 else ""
    widemax = spec[4] if len(spec) > 4 else ""
    widehyps = spec[5] if len(spec) > 5 else ""
    wide8 = spec[6] if len(spec) > 6 else ""
    env = dict(os.environ, PMHIP_LIB=os.path.join(root, "openmvs_amd", lib))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extras", "--no-tiled-leg", "--no-shard-rates", "--views-per-gpu", views, "--groups", groups] + (["--lanes", lanes] if lanes else []) + (["--wide-pixels", widepx] if widepx else []) + (["--wide-max-views", widemax] if widemax else []) + (["--wide-hyps", widehyps] if widehyps else []) + (["--wide8-pixels", wide8] if wide8 else []) + os.environ.get("TUNE_STEPS", "").split(), env=env, capture_output=True, text=True, timeout=400)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-18s groups=%-2s lanes=%-2s wpx=%-6s wmax=%-3s hyps=%-2s w8px=%-5s views=%s  %.2f Mpix/s  step %.0f ms  avg_launch %.1f us" % (lib, groups, lanes or "-", widepx or "-", widemax or "-", widehyps or "-", wide8 or "-", views, j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"]), flush=True)
    except Exception as ex:
        print(lib, groups, "FAILED", ex, r.stderr[-500:], flush=True)

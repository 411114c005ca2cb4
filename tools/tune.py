"""Run bench.py for several (library variant, PMHIP_GROUPS[, PMHIP_LANES[, PMHIP_BAND]]) settings given as lib:groups[:lanes[:band]]; prints one line each."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
views = sys.argv[1] if len(sys.argv) > 1 else "48"
variants = [a.split(":") for a in sys.argv[2:]] or [["libpmhip.so", "1"]]
for spec in variants:
    lib, groups = spec[0], spec[1]
    lanes = spec[2] if len(spec) > 2 else ""
    env = dict(os.environ, PMHIP_LIB=os.path.join(root, "openmvs_amd", lib), PMHIP_GROUPS=groups)
    if lanes:
        env["PMHIP_LANES"] = lanes
    band = spec[3] if len(spec) > 3 else ""
    if band:
        env["PMHIP_BAND"] = band
    if len(spec) > 4 and spec[4]:
        env["PMHIP_BAND_CHUNK"] = spec[4]
    if len(spec) > 5 and spec[5]:
        env["PMHIP_BAND_SLACK"] = spec[5]
    if len(spec) > 6 and spec[6]:
        env["PMHIP_DIAG2"] = spec[6]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extras", "--views-per-gpu", views], env=env, capture_output=True, text=True, timeout=400)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-18s groups=%-2s lanes=%-2s band=%-1s chunk=%-4s slack=%-3s diag2=%-1s views=%s  %.2f Mpix/s  step %.0f ms  avg_launch %.1f us" % (lib, groups, lanes or "-", band or "-", (spec[4] if len(spec) > 4 else "-"), (spec[5] if len(spec) > 5 else "-"), (spec[6] if len(spec) > 6 else "-"), views, j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"]), flush=True)
    except Exception as ex:
        print(lib, groups, "FAILED", ex, r.stderr[-500:], flush=True)

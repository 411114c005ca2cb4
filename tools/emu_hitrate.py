"""LDS window hit rate of the sweep kernel for a build variant, counted under the CPU emulator (no GPU needed; the counts are a property of the
geometry, not of the hardware):  python tools/emu_hitrate.py -DPM_TR=21 -DPM_TCX=10
Prints the share of lane tap-rows served from the LDS window and the number of tap rows for which the whole wave stayed on the LDS path (a wave with one
lane outside executes the global-load path as well)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OPENMVS_AMD_TEST_EMULATOR"] = "1"
flags = sys.argv[1:]
tag = "_".join(f.replace("-D", "").replace("=", "") for f in flags) or "default"
out = "/tmp/libpm_emu_hitrate_%s.so" % tag
if not os.path.exists(out):
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value",
                           "-Wno-unknown-attributes", "-DPM_PROFILE", "-I", os.path.join(ROOT, "tests", "cpp", "hipemu")] + flags + ["pm_engine.hip", "-o", out], cwd=os.path.join(ROOT, "openmvs_amd", "csrc"))
os.environ["PMHIP_LIB"] = out
from openmvs_amd import synth
from openmvs_amd.patchmatch import PatchMatchHIP, default_params
W, H, V = 240, 136, 9
sc = synth.make_scene(V, W, H, n_src=8, seed=5)
e = PatchMatchHIP(0); e.Init(True); e.scene_load(sc, 1)
p = default_params(seed=1, nSubResolutionLevels=1)
ids = [0, 4]
for v in ids: e.scene_reset_view(v)
e.prof_get(True)
e.scene_estimate(ids, -1, p); e.sync()
c = e.prof_get(True)
print(tag, "LDS-served lane-rows %.2f %% of %d; whole-wave rows %d of %d" % (100.0 * c[10] / max(1, c[11]), c[11], c[7], c[11] // 64), flush=True)

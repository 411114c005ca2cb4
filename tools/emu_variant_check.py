"""Parity of a build VARIANT of libpmhip (compile-time switches) under the CPU emulator, before any GPU time is spent on it:
    python tools/emu_variant_check.py -DPM_SMOOTH_IN_ROW0=1
compiles openmvs_amd/csrc/pm_engine.hip for the host against tests/cpp/hipemu with the given flags and runs a selection of the -m gpu parity test
bodies (8 / 4 / 1-3 sources, pyramid, geometric round, masks, option sets, a 9-view batch) on it, bit for bit against the oracle."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
tag = "_".join(f.replace("-D", "").replace("=", "") for f in flags) or "default"
out = "/tmp/libpm_emu_variant_%s.so" % tag
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value", "-Wno-unknown-attributes",
                       "-I", os.path.join(ROOT, "tests", "cpp", "hipemu")] + flags + ["pm_engine.hip", "-o", out], cwd=os.path.join(ROOT, "openmvs_amd", "csrc"))
os.environ["OPENMVS_AMD_TEST_EMULATOR"] = "1"; os.environ["PMHIP_LIB"] = out
from openmvs_amd import patchmatch, synth
from tests import test_gpu_patchmatch as g
small = synth.make_scene(5, 160, 120, n_src=4); nine = synth.make_scene(9, 128, 96, n_src=8)
e = patchmatch.PatchMatchHIP(0); e.tuning(wideMaxViews=-1); e.Init(False)      # the regular sweep kernel (the speculative ones: tests/test_emu_kernels.py)
g.test_single_view_parity_N8_and_N1(e, nine); print("8 / 1 / 2 / 3 sources ok", flush=True)
g.test_single_view_photometric_parity_N4(e, small, 2); print("4 sources, pyramid ok", flush=True)
g.test_single_call_with_ignore_mask(e, small); print("mask ok", flush=True)
for k in (0, 2, 3): g.test_non_default_options_parity(e, small, k)
print("option sets ok", flush=True)
g.test_geometric_round_parity_and_golden(e); print("geometric round + golden ok", flush=True)
g.test_scene_batch_full_schedule_matches_oracle(small); print("scene batch ok", flush=True)
e.close()
from tests import emu
launches, fibers, exchanges, inactive = emu.counters(patchmatch)
assert inactive == 0, "%d cross-lane reads of lanes that were not executing the operation" % inactive
print("launches %d, fibers %d, cross-lane exchanges %d, reads of inactive lanes %d" % (launches, fibers, exchanges, inactive))
print("variant", tag, ": all selected parity checks bit-exact under the emulator")

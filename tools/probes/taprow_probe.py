"""Runs tools/probes/taprow_probe.hip: SIMD time per fast tap row (5 taps) of the sweep kernel against the number of resident waves.
    python tools/probes/taprow_probe.py            (GPU box; builds the probe if hipcc is there, else uses the shipped .so)"""
import ctypes as C, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libtaprow_probe.so")
src = os.path.join(HERE, "taprow_probe.hip")
if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                           "-Wno-unused-value", "-Wno-pass-failed", src, "-o", so])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
lib = C.CDLL(so)
CUS, SIMDS = 256, 1024
rows = 4000
static_lds = 8 * (20 * 18 + 4) * 4 + 8 * 26 * 8
print("static LDS per workgroup %d B; %d tap rows per wave" % (static_lds, rows))
import ctypes
jit = float(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "build" else 0.0
print("jitter", jit)
for per_cu in (4, 11, 12):
    dyn = max(0, (160 * 1024) // per_cu - static_lds - 512) if per_cu < 12 else 0
    fit = (160 * 1024) // (static_lds + dyn + 0)
    blocks = CUS * min(fit, per_cu) * 4
    for mode in (0, 1):
        ms, okf = C.c_float(), C.c_float()
        rc = lib.taprow_probe(blocks, rows, dyn, mode, C.c_float(jit), C.byref(ms), C.byref(okf))
        waves_per_simd = min(fit, per_cu) / 4.0
        ns_row_simd = ms.value * 1e6 / (rows * blocks / SIMDS)
        print("%2d workgroups per CU wanted (LDS allows %2d; %.2f waves/SIMD), %s: %.3f ms for %d waves -> %.1f ns of SIMD time per row (= %.0f cycles at 2.4 GHz)%s" % (
            per_cu, fit, waves_per_simd, "tap rows " if mode == 0 else "empty loop", ms.value, blocks, ns_row_simd, ns_row_simd * 2.4, "" if rc == 0 else "  rc %d" % rc), flush=True)

"""Static instruction mix of one kernel of libpmhip.so / libsgmhip.so as hipcc emits it for gfx950 (no GPU needed):
    python tools/isa_stats.py pm_sweep_kernelILi8ELi1ELb0E [--loop] [-DFLAG ...]
prints, for the whole kernel and for its outermost loop (the per-evaluation trip of the sweep kernel), how many VALU / SALU / LDS / vector-memory /
scratch instructions, branches, waits, f64, transcendental, division-expansion, integer-multiply and SGPR-spill (v_readlane / v_writelane) instructions
there are.  Round 2 used these numbers next to the timing probes."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmvs_amd import build as _b
name = sys.argv[1]
flags = [a for a in sys.argv[2:] if a.startswith("-D")]
src = "sgm_engine.hip" if "sgm" in name else "pm_engine.hip"
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call([_b.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "--cuda-device-only", "-S",
                           "-Wno-unused-value", "-Wno-pass-failed"] + flags + [os.path.join(_b._CSRC, src), "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\d+" + re.escape(name) + r".*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
CLASSES = [("valu", r"^\tv_"), ("salu", r"^\ts_(?!waitcnt|load|buffer_load|cbranch|branch|barrier|nop)"), ("lds", r"^\tds_"), ("vmem load", r"^\t(global|flat|buffer)_load"),
           ("vmem store/atomic", r"^\t(global|flat|buffer)_(store|atomic)"), ("scalar load", r"^\ts_(load|buffer_load)"), ("scratch", r"^\tscratch_"), ("branch", r"^\ts_c?branch"),
           ("waitcnt", r"^\ts_waitcnt"), ("s_nop", r"^\ts_nop"), ("f64", r"_f64"), ("transcendental", r"^\tv_(rcp|rsq|sqrt|exp|log|sin|cos)_"), ("division parts", r"^\tv_div_(scale|fmas|fixup)"),
           ("int multiply", r"^\tv_(mul_hi|mul_lo|mad_u64|mad_i64)"), ("packed f32", r"^\tv_pk_"), ("dpp", r" (quad_perm|row_|wave_)"), ("sgpr spill moves", r"^\tv_(readlane|writelane)")]


def stats(ls, title):
    print(title, "(%d lines)" % len(ls))
    for n, pat in CLASSES:
        c = sum(1 for l in ls if re.search(pat, l))
        if c:
            print("   %-20s %6d" % (n, c))


stats(body, name)
loops = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
if loops:
    stats(body[loops[-1]:], "last outermost loop to the end of the kernel")

"""Static instruction mix of the sweep kernels' hot loop (the two-deep tap-row pipeline of pm_taps_fast, csrc/pm_kernels.hip) from the gfx950 assembly the product is built
from: per kernel, the loop that contains the inline-assembly tap-row loads (`buffer_load_dwordx4 ... idxen`), its instructions by class, and what that is per tap (the loop
body covers two rows = ten taps per trip, two trips + one row per 25-tap patch).  No GPU needed: `python tools/isa_mix.py [substring of a kernel name]`."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmvs_amd import build as _b


def assembly():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        flags = [f for f in _b.FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.check_call([_b.HIPCC] + flags + _b.LIB_FLAGS.get("libpmhip.so", []) + ["--cuda-device-only", "-S", os.path.join(_b._CSRC, "pm_engine.hip"), "-o", out],
                              stderr=subprocess.DEVNULL, cwd=_b._CSRC)
        return open(out).read().split("\n")


def classify(op):
    if op.startswith("v_pk_"): return "valu_packed"
    if op.startswith(("v_cmp", "v_cmpx")): return "valu_cmp"
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")): return "valu_trans"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "valu_lane"
    if op.startswith("v_"): return "valu"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def kernels(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name and ln.startswith(".Lfunc_end"):
            yield name, body
            name = None
            continue
        if name:
            body.append(ln)


def demangle(n):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            return subprocess.check_output([tool, n], stderr=subprocess.DEVNULL).decode().strip()
        except Exception:
            continue
    return n


def hot_loops(body):
    """Loops (label ... backward branch to it) that contain an idxen tap-row load; innermost first."""
    labels = {}
    code = []
    for ln in body:
        t = ln.split(";")[0].strip()
        if not t or t.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", t):
                labels[t[:-1]] = len(code)
            continue
        if t.endswith(":"):
            labels[t[:-1]] = len(code)
            continue
        code.append(t)
    loops = []
    for i, t in enumerate(code):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] <= i:
                seg = code[labels[tgt]:i + 1]
                if any("idxen" in s for s in seg):
                    loops.append(seg)
    loops.sort(key=len)
    return loops


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "pm_sweep2_kernel"
    for name, body in kernels(assembly()):
        dn = demangle(name)
        if want not in dn:
            continue
        loops = hot_loops(body)
        if not loops:
            continue
        seg = loops[0]
        mix = collections.Counter(classify(s.split()[0]) for s in seg)
        loads = sum(1 for s in seg if "idxen" in s)
        taps = loads                                   # one 16-byte load = one bilinear tap
        valu = mix["valu"] + mix["valu_packed"] + mix["valu_cmp"] + mix["valu_trans"] + mix["valu_lane"]
        ops = collections.Counter(s.split()[0] for s in seg if s.startswith("v_"))
        print(dn)
        print("  tap-row loop: %d instructions, %d tap loads (taps per trip); VALU %d = %.1f per tap (plain %d, packed %d, compare %d, transcendental %d, lane %d); "
              "LDS %d, VMEM %d, SALU %d, s_waitcnt %d, branches %d" % (len(seg), loads, valu, valu / max(1, taps), mix["valu"], mix["valu_packed"], mix["valu_cmp"], mix["valu_trans"],
                                                                       mix["valu_lane"], mix["lds"], mix["vmem"], mix["salu"], mix["s_waitcnt"], mix["branch"]))
        print("  most frequent VALU opcodes: " + ", ".join("%s x%d" % kv for kv in ops.most_common(14)))


if __name__ == "__main__":
    main()

"""Static check of the hand-scheduled tap-row loads (pm_bufload5 / pm_bufwait5, csrc/pm_kernels.hip): the 16-byte buffer loads are issued from inline assembly, outside
the compiler's s_waitcnt bookkeeping, so nothing may read, copy, spill or overwrite one of their destination registers before the hand-written `s_waitcnt vmcnt(N)` that
covers it.  This walks the gfx950 assembly of every kernel that contains such loads and models the vector-memory queue in program order (vmcnt retires in issue order on
gfx9-family parts): every VMEM instruction is an entry, an `s_waitcnt vmcnt(N)` retires all but the youngest N, and any other instruction that names a register of a
still-in-flight inline-asm load is a violation.  (A linear walk: the two-deep row pipeline leaves the same queue at the loop's back edge as at its entry.)
    python tools/isa_inflight_check.py            # prints per kernel: loads checked, violations
Used by tests/test_kernel_resources.py."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmvs_amd import build as _b

VMEM = re.compile(r"^\s*(buffer|global|flat|scratch)_(load|store|atomic)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def assembly(extra_flags=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        flags = [f for f in _b.FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.check_call([_b.HIPCC] + flags + _b.LIB_FLAGS.get("libpmhip.so", []) + ["--cuda-device-only", "-S"] + list(extra_flags) +
                              [os.path.join(_b._CSRC, "pm_engine.hip"), "-o", out], stderr=subprocess.DEVNULL, cwd=_b._CSRC)
        return open(out).read().split("\n")


def check(lines=None):
    """{kernel: (inline-asm loads seen, [violations])}"""
    lines = lines if lines is not None else assembly()
    res, name, queue, n_loads, bad = {}, None, [], 0, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m and not ln.startswith(".L"):
            name, queue, n_loads, bad = m.group(1), [], 0, []
            continue
        if name is None:
            continue
        if ln.startswith(".Lfunc_end"):
            if n_loads:
                res[name] = (n_loads, bad)
            name = None
            continue
        code = ln.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        w = re.match(r"s_waitcnt\b(.*)", code)
        if w:
            v = re.search(r"vmcnt\((\d+)\)", w.group(1))
            if v:
                keep = int(v.group(1))
                queue = queue[len(queue) - keep:] if keep else []
            elif "vmcnt" not in w.group(1) and re.fullmatch(r"\s*\d+\s*", w.group(1) or ""):
                queue = []            # a raw immediate: be conservative the other way round is impossible; hipcc prints the symbolic form
            continue
        inflight = set().union(*[q for q in queue if q]) if queue else set()
        if VMEM.match(code):
            hand = code.startswith("buffer_load_dwordx4") and "idxen" in code
            ops = code.split(None, 1)[1] if " " in code else ""
            first = ops.split(",")[0]
            touched = _regs(ops)
            if touched & inflight:
                bad.append(code)
            queue.append(_regs(first) if hand else set())     # only the inline-asm loads are outside the compiler's own bookkeeping
            n_loads += hand
            continue
        if inflight and (_regs(code) & inflight):
            bad.append(code)
    return res


if __name__ == "__main__":
    r = check()
    for k in sorted(r):
        print("%-80s loads %3d violations %d" % (k[:80], r[k][0], len(r[k][1])))
        for b in r[k][1][:5]:
            print("      ", b)
    sys.exit(1 if any(v[1] for v in r.values()) else 0)

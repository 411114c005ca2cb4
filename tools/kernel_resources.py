"""Register / LDS / scratch / occupancy of every kernel of libpmhip.so as the compiler reports them (no GPU needed):
    python tools/kernel_resources.py [-DPM_TCX=10 ...]
Used by tests/test_kernel_resources.py to pin the figures DESIGN.md quotes for the sweep kernel."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmvs_amd import build as _b


def resources(extra_flags=(), lib="libpmhip.so"):
    """{mangled kernel name: dict(vgpr, agpr, sgpr, scratch, occupancy, lds)} from -Rpass-analysis=kernel-resource-usage."""
    srcs, _ = _b.LIBS[lib]
    with tempfile.TemporaryDirectory() as td:
        # the product's flags, the per-library ones included (-fno-slp-vectorize changes the register allocation of every kernel here)
        cmd = [_b.HIPCC] + _b.FLAGS + _b.LIB_FLAGS.get(lib, []) + ["-Rpass-analysis=kernel-resource-usage"] + list(extra_flags) + [os.path.join(_b._CSRC, s) for s in srcs] + ["-o", os.path.join(td, "x.so")]
        p = subprocess.run(cmd, cwd=_b._CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
    out = {}
    for blk in re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]:
        name = blk.split("\n")[0].split(" ")[0]
        def g(key):
            m = re.search(key + r": (\d+)", blk)
            return int(m.group(1)) if m else -1
        out[name] = dict(vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("SGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                         occupancy=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]"))
    return out


if __name__ == "__main__":
    r = resources(sys.argv[1:])
    for k in sorted(r):
        v = r[k]
        print("%-70s VGPR %3d SGPR %3d scratch %3d B  occupancy %d  LDS %5d B" % (k[:70], v["vgpr"], v["sgpr"], v["scratch"], v["occupancy"], v["lds"]))

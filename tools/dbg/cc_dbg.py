import numpy as np, subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openmvs_amd import synth
h, w = 120, 160
sc = synth.make_scene(2, w, h, n_src=1)
d = sc.gt_depth[0].copy(); d[:4] = 0; d[-4:] = 0; d[:, :4] = 0; d[:, -4:] = 0
r = np.random.RandomState(0); d[r.rand(h, w) < 0.05] = 0
d.astype(np.float32).tofile("/tmp/cc_in.bin")
here = os.path.dirname(os.path.abspath(__file__))
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", os.path.join(here, "cc_dbg.hip"), "-o", "/tmp/cc_dbg"])
print(subprocess.run(["/tmp/cc_dbg", str(w), str(h), "0.007", "/tmp/cc_in.bin", "/tmp/cc_out.bin"], capture_output=True, text=True).stdout)
raw = np.fromfile("/tmp/cc_out.bin", np.int32); parent = raw[:w * h]; size = raw[w * h:]
# python model
par = list(range(w * h))
def find(a):
    while par[a] != a:
        par[a] = par[par[a]]; a = par[a]
    return a
th = np.float32(0.007)
sim = lambda a, b: abs(np.float32(a) - np.float32(b)) / np.float32(a) < th
for y in range(h):
    for x in range(w):
        if not d[y, x] > 0: continue
        for qx, qy in ((x + 1, y), (x, y + 1)):
            if qx < w and qy < h and d[qy, qx] > 0 and sim(d[y, x], d[qy, qx]) and sim(d[qy, qx], d[y, x]):
                a, b = find(x * h + y), find(qx * h + qy)
                if a != b: par[max(a, b)] = min(a, b)
root = np.array([find(i) for i in range(w * h)])
print("roots equal:", np.array_equal(root, parent), "n comps model", len(set(root)), "gpu", len(set(parent)), "mismatch", int((root != parent).sum()))
print("sizes ok:", np.array_equal(np.bincount(root, minlength=w * h), size))
bad = np.flatnonzero(root != parent)[:10]
print([(int(i), int(root[i]), int(parent[i])) for i in bad])

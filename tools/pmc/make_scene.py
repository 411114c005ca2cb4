"""Writes the bench scene (synth.make_scene_torch, the generator bench.py uses) as a flat file for tools/pmc/pmc_workload.cpp:
    python tools/pmc/make_scene.py <views> <width> <height> <out.bin>
Runs unprofiled (it uses torch on the GPU when there is one); the profiled process is the C++ program."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from openmvs_amd import synth
V, W, H, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dev = "cuda" if torch.cuda.is_available() else "cpu"
sc = synth.make_scene_torch(V, W, H, n_src=8, device=dev, gt_views=0)
gray = sc["gray"].cpu().numpy()
nb = np.asarray(sc["neighbors"], np.int32)
with open(out, "wb") as f:
    f.write(np.array([V, W, H, nb.shape[1]], np.int32).tobytes())
    for i in range(V):
        f.write(np.ascontiguousarray(gray[i], np.float32).tobytes())
        f.write(np.concatenate([np.asarray(sc["K"][i]).ravel(), np.asarray(sc["R"][i]).ravel(), np.asarray(sc["C"][i]).ravel()]).astype(np.float64).tobytes())
        f.write(np.array([sc["dmin"][i], sc["dmax"][i]], np.float32).tobytes()); f.write(np.ascontiguousarray(nb[i]).tobytes())
print("wrote", out, V, W, H)

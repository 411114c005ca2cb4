"""FuseDepthMaps timing probe (GPU box): V views of WxH with perturbed ground-truth maps; prints points, rounds and milliseconds
(device call including the download; add --oracle to time the sequential CPU oracle on the same input).
    python tools/probe_fuse.py [views=9] [width=1920] [height=1080] [--oracle]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvs_amd import synth
from openmvs_amd.patchmatch import PatchMatchHIP
from tests import fuse_cases as fc

args = [a for a in sys.argv[1:] if not a.startswith("--")]
V = int(args[0]) if len(args) > 0 else 9
W = int(args[1]) if len(args) > 1 else 1920
H = int(args[2]) if len(args) > 2 else 1080
sc = synth.make_scene(V, W, H, n_src=8, device="cuda")
maps = fc.make_maps(sc, seed=7)
e = PatchMatchHIP(0)
e.scene_load(sc, n_levels=0)
for v in range(V):
    e.scene_set_maps(v, maps[0][v], maps[1][v]); e.scene_set_conf(v, maps[2][v]); e.scene_set_color(v, sc.bgr[v])
order = sorted(range(V), key=lambda i: (-len(sc.neighbors[i]), i))
e.scene_fuse(order)                                   # allocations
for colors in (True, False):
    t = time.time(); r = e.scene_fuse(order, bEstimateColor=colors, bEstimateNormal=colors); dt = time.time() - t
    print("views %d %dx%d colours/normals %s: %d points from %d depths, %d rounds, %.1f ms (%.1f Mdepths/s)" % (
        V, W, H, colors, r["nPoints"], r["nDepths"], r["rounds"], dt * 1e3, r["nDepths"] / dt / 1e6), flush=True)
if "--oracle" in sys.argv:
    from oracle import pyoracle as po
    t = time.time(); ref = po.fuse_depth_maps(*maps, list(sc.bgr), sc.K, sc.R, sc.C, [list(x) for x in sc.neighbors], order=order); dt = time.time() - t
    print("sequential oracle: %d points, %.0f ms" % (ref["nPoints"], dt * 1e3))
e.close()

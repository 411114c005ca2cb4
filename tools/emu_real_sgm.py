"""One-off check (about 4.5 minutes, CPU only): the SGM chain on the reference's pipeline-test scene with the DEVICE code path executed under the wave64
emulator of the tests -- stereo rectification, seeded resident tSGM loop (sgmhip_tsgm_match), ProjectDisparity2DepthMap, pair fusion -- against the same
chain on the oracle backend.  Expected output: both pairs and the fused depth / confidence maps equal.

    OPENMVS_AMD_TEST_EMULATOR=1 SGMHIP_LIB=tests/cpp/hipemu/_build/libsgmhip_emu.so python tools/emu_real_sgm.py      (build the library first: python -m pytest tests/test_emu_kernels.py -k sgm_match)
"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from openmvs_amd import mvsi, views, sgm, sgm_pipeline, mvsfront
from tests.tsgm_backends import OracleBackend
SCENE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'data', 'scene')
sc = mvsi.load(os.path.join(SCENE, "scene.mvs")); cams = views.Cameras(sc)
bgr = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(SCENE, im.name)).convert("RGB"))[..., ::-1]) for im in sc.images]
own = np.repeat(np.arange(len(sc.vertices)), np.diff(sc.vertex_view_start)); ids = sc.vertex_views["image_id"]
seen = []
for i in range(len(sc.images)):
    s = np.zeros(len(sc.vertices), bool); s[own[ids == i]] = True; seen.append(s)
cam = lambda i: (cams.K[i], cams.R[i], cams.C[i])
ok, nb, pts, avg = views.select_neighbor_views(sc, cams, 0)
pts0 = np.nonzero(seen[0])[0].astype(np.uint32)
cf = mvsfront.SceneFront(os.path.join(SCENE, "scene.mvs"))
seed = lambda w, h: cf.triangulate_depth_map(0, pts0, (w, h), avg_depth=avg)[0]
dev = sgm.SemiGlobalMatcherHIP(0); orc = OracleBackend()
res = {}
for name, be in (("device", dev), ("oracle", orc)):
    t = time.time(); pairs = []
    for B in (2, 3):
        p = sgm_pipeline.match_pair(be, bgr[0], cam(0), bgr[B], cam(B), sc.vertices[seen[0] & seen[B]], min_resolution=160, seed_depth=seed)
        pairs.append(p)
    depth, conf = sgm_pipeline.fuse_pairs(be, pairs, 2)
    res[name] = (pairs, depth, conf); print(name, time.time() - t, (depth > 0).mean(), flush=True)
for i in range(2):
    print('pair', i, np.array_equal(res['device'][0][i]['disparity'], res['oracle'][0][i]['disparity']), np.array_equal(res['device'][0][i]['cost'], res['oracle'][0][i]['cost']))
print('fused depth equal', np.array_equal(res['device'][1], res['oracle'][1]), 'conf equal', np.array_equal(res['device'][2], res['oracle'][2]))

"""Cross-compile sweep-kernel variants for tools/gpu_call.sh's `variants` step (no GPU needed; the .so files travel with the gpurun snapshot):
    python tools/build_variants.py name=-DFLAG,-DFLAG ...      e.g.  tcx10=-DPM_TCX=10 tr19=-DPM_TR=19,-DPM_TCX=10"""
import concurrent.futures as cf, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvs_amd.build import build_variant
V = [(a.split("=", 1)[0], a.split("=", 1)[1].split(",")) for a in sys.argv[1:]]
with cf.ThreadPoolExecutor(max(1, len(V))) as ex:
    print(list(ex.map(lambda v: build_variant("libpmhip.so", "libpmhip_%s.so" % v[0], v[1]), V)))

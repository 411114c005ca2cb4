"""Cross-compile the sweep-kernel variants tools/round2_first_call.sh times (no GPU needed; the .so files travel with the gpurun snapshot)."""
import concurrent.futures as cf, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmvs_amd.build import build_variant
V = [("libpmhip_wb8.so", ["-DPM_WINBATCH=8"]), ("libpmhip_noxcd.so", ["-DPM_XCD_REMAP=0"]), ("libpmhip_tcx10.so", ["-DPM_TCX=10"]),
     ("libpmhip_tr16.so", ["-DPM_TR=16"]), ("libpmhip_tr16_w4.so", ["-DPM_TR=16", "-DPM_MINWAVES=4"])]
with cf.ThreadPoolExecutor(len(V)) as ex:
    print(list(ex.map(lambda v: build_variant("libpmhip.so", v[0], v[1]), V)))

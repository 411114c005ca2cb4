"""Generates tests/golden/sgm_golden_80x60.npz from the CPU SGM oracle (python tests/golden/make_sgm_golden.py).
Self-contained: inputs + oracle outputs.  Not an output of the reference binary (it cannot be built here)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import pyoracle as po  # noqa: E402
from tests import sgm_cases as sc  # noqa: E402

lb, lg, rg = sc.stereo_pair(80, 60, 4, seed=3)
px, n, mx = sc.ranges(80, 60, "ragged", -2, 22, seed=5)
d, c, costs, acc = po.sgm_match(lb, lg, rg, px, n, mx, 3, po.sgm_generate_p2s())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sgm_golden_80x60.npz"), left_bgr=lb, left_gray=lg, right_gray=rg,
                    idx=px["idx"], minDisp=px["minDisp"], maxDisp=px["maxDisp"], num_costs=n, max_num_disp=mx, disparity=d, cost=c, costs=costs, accums=acc)
print("written", n, mx)

#!/usr/bin/env python3
"""tools/calibrate.py -- write bench.py's head-calibration cache (/tmp/insmos_bench_calibration.json) so that `bench.py --timed-only`
can run first on a fresh GPU box (the calibration = the score threshold at which the random-weight head keeps 1500 candidates)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402

cfg = P.default_cfg()
model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
win = torch.from_numpy(bench.load_windows([0], 1886)[0]).cuda()
bench.calibrate_head(model, win, 1500, cache="/tmp/insmos_bench_calibration.json", tag="rank0_az1886_c1500")
print("calibration written")

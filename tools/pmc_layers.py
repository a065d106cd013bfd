#!/usr/bin/env python3
"""tools/pmc_layers.py -- workload for per-layer PMC collection: 2 warm-up windows + 1 window of cfg-2 (S0), and the
order of the sparse-conv launches of that window written to gpurun_out/conv_order.json.  Run under
`rocprofv3 --kernel-trace --pmc ...` (tools/pmc_layers.sh); tools/pmc_join.py joins the passes per layer."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402

cfg = P.default_cfg()
sd = P.random_state_dict(cfg, seed=0)
window = bench.load_window(0, 1886)
pts = torch.from_numpy(window).cuda()
model = InsMOSNet(cfg, state_dict=sd).cuda(0).eval()
bench.calibrate_head(model, pts, 1500)
eng = model.model.engine
for _ in range(3):
    model.forward([{"past_point_clouds": pts}], "test")
torch.cuda.synchronize()
order = [(l.name, l.K, l.cin, l.cout, int(n)) for (_, n, l, _r0) in eng._conv_log]
os.makedirs("gpurun_out", exist_ok=True)
json.dump(order, open("gpurun_out/conv_order.json", "w"))
print("conv launches per window:", len(order))

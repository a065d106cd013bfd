#!/usr/bin/env python3
"""tools/train_motionnet_bench.py -- time of one MotionNet training step (forward + loss + backward + SGD) on the S0 window."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.synth import make_labels  # noqa: E402
from insmos_amd.train_motionnet import MotionNetTrainer  # noqa: E402

n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 1886
cfg = P.default_cfg()
w = bench.load_window(0, n_az)
pts = torch.from_numpy(w).cuda()
gt = torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=0)).cuda()
tr = MotionNetTrainer(cfg, P.random_state_dict(cfg, 0))
lib = tr.engine.lib
for i in range(6):
    if i == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    loss = tr.loss(pts, gt)
    loss.backward()
    tr.sgd_step(0.01)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 4
print(f"MotionNet training step, {len(w)} points, {tr.engine.last_counts['me_voxels']} voxels: {dt * 1e3:.1f} ms "
      f"(loss {float(loss.detach()):.4f}); peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)

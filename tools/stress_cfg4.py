#!/usr/bin/env python3
"""tools/stress_cfg4.py -- BASELINE.json configs[3]: dense scene, 300k pts/scan (n_az=4710), voxel 0.05 m, N=10, one GPU.
Random weights (the 0.05 m grid needs NUM_BEV_FEATURES 640, not weight-compatible with the 0.1 m checkpoints).  Prints the
coordinate-set sizes (SURVEY.md section 8d lists the known answers) and the time per window."""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.engine import Engine  # noqa: E402

cfg = copy.deepcopy(P.default_cfg())
cfg["DATA"]["VOXEL_SIZE"] = [0.05, 0.05, 0.05]
cfg["MODEL"]["MAP_TO_BEV"]["NUM_BEV_FEATURES"] = 640
cfg["MODEL"]["DENSE_HEAD"]["TARGET_ASSIGNER_CONFIG"]["VOXEL_SIZE"] = [0.05, 0.05, 0.05]
sd = P.random_state_dict(cfg, 4)
t0 = time.perf_counter()
w = bench.load_window(0, 4710)
print(f"window: {len(w)} points ({time.perf_counter() - t0:.1f} s to generate)", flush=True)
pts = torch.from_numpy(w).cuda()
eng = Engine(cfg, sd, native=True)
for _ in range(2):
    logits, pred = eng.forward_window(pts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    logits, pred = eng.forward_window(pts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
c = eng.last_counts
print("4D voxels", c["me_voxels"], "current points", c["n_cur"], "3D voxels", c["unet_voxels"], "boxes", c["n_boxes"])
print(f"{dt * 1e3:.2f} ms per window ({1 / dt:.1f} windows/s), arena {eng._arena.numel() / 2**30:.2f} GiB, "
      f"logits finite: {bool(torch.isfinite(logits).all())}", flush=True)
# the same window as a launch set of B copies (the level-0 table of more than two such windows passes 2 GiB: the engine
# then splits the set by itself)
for B in (2, 4):
    wins = [pts] * B
    for _ in range(2):
        eng.forward_windows(wins)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        res = eng.forward_windows(wins)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (3 * B)
    same = all(torch.equal(r[0], logits) for r in res)
    print(f"launch set of {B}: {dt * 1e3:.2f} ms per window ({1 / dt:.1f} windows/s), sets of {eng.last_counts['batch']}, "
          f"bits as the single window: {same}", flush=True)

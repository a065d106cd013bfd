#!/usr/bin/env python3
"""tools/batched_motionnet_probe.py -- first measurement for DESIGN.md section 2: MotionNet of B S0 windows in one
set of launches (Engine.motionnet_windows) against the same windows one after the other, step path, one stream.

    python tools/batched_motionnet_probe.py [B=4] [n_az=1886]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_az = int(sys.argv[2]) if len(sys.argv) > 2 else 1886
cfg = P.default_cfg()
eng = Engine(cfg, P.random_state_dict(cfg, 0), "cuda:0")
wins = [torch.from_numpy(bench.load_window(i, n_az)).cuda() for i in range(B)]
lib = eng.lib


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    lib.insmos_prof_reset()
    lib.insmos_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    prof = bench.read_profile(lib)
    lib.insmos_prof_enable(0)
    conv = prof.get("sparse_conv_mfma", (0.0, 0))
    return dt, conv[0] / reps, conv[1] // reps


seq, seq_conv, seq_n = timed(lambda: [eng.motionnet(w) for w in wins])
bat, bat_conv, bat_n = timed(lambda: eng.motionnet_windows(wins))
a = [eng.motionnet(w).clone() for w in wins]
b = eng.motionnet_windows(wins)
same = all(torch.equal(x, y) for x, y in zip(a, b))
print(f"MotionNet, {B} windows of {len(wins[0])} points: one after the other {seq * 1e3:.2f} ms (conv kernels {seq_conv:.2f} ms in "
      f"{seq_n} launches), batched {bat * 1e3:.2f} ms (conv {bat_conv:.2f} ms in {bat_n} launches); identical bits: {same}")

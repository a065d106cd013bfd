#!/usr/bin/env python3
"""tools/bn_shape_probe.py -- segmented BatchNorm forward + backward time per layer shape and chunk length (4 segments, like a B = 4
training step): which chunk length each (rows, channels) wants."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd.autograd import BnPlan, batch_norm_train_seg

shapes = [(1_900_000, 8), (850_000, 8), (850_000, 16), (335_000, 16), (335_000, 32), (120_000, 32), (170_000, 16), (123_000, 32),
          (50_000, 64), (27_000, 128), (75_000, 128), (300_000, 256)]
S = 4
for n, c in shapes:
    x = torch.randn((n, c), device="cuda", requires_grad=True)
    g = torch.ones(c, device="cuda", requires_grad=True)
    b = torch.zeros(c, device="cuda", requires_grad=True)
    dy = torch.randn((n, c), device="cuda")
    cells = []
    for chunk in (128, 256, 512, 1024, 2048, 4096):
        os.environ["INSMOS_BN_CHUNK"] = str(chunk)
        q = n // S
        plan = BnPlan([(i * q, (i + 1) * q if i < S - 1 else n, i) for i in range(S)], n, S, "cuda")

        def step():
            y = batch_norm_train_seg(x, g, b, plan, relu=True, force_segmented=True)
            y.backward(dy)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        cells.append("%5d: %6.1f us" % (chunk, (time.perf_counter() - t0) * 1e5))
    bytes_ = n * c * 4 * 11
    print("n %8d c %3d (%.0f MB traffic -> %.0f us at 4 TB/s) | " % (n, c, bytes_ / 1e6, bytes_ / 4e6) + " | ".join(cells), flush=True)

#!/usr/bin/env python3
"""tools/cfg5_torch_ops.py -- which torch (aten) ops the cfg-5 training step launches beside the library's kernels: torch.profiler over
four steps, ops ranked by device time with their call counts, and the Python source lines that call the most expensive ones."""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.synth import make_labels  # noqa: E402
from insmos_amd.train_unet import InsMOSTrainer  # noqa: E402


def main():
    dev = "cuda:0"
    B = 4
    cfg = P.default_cfg()
    wins = bench.load_windows(list(range(B)), 1886)
    rng = np.random.default_rng(1000)
    batch = [{"past_point_clouds": torch.from_numpy(w).to(dev),
              "past_labels": [None, torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=i)).to(dev)],
              "gt_boxes": torch.from_numpy(bench.synthetic_gt_boxes(rng)).to(dev)} for i, w in enumerate(wins)]
    tr = InsMOSTrainer(cfg, P.random_state_dict(cfg, 0, cls_bias=-2.0, box_w_std=0.05), device=dev)
    opt = torch.optim.Adam(list(tr.params.values()), lr=float(cfg["TRAIN"]["LR"]))

    def step():
        opt.zero_grad(set_to_none=True)
        loss, tb, gt, pred = tr.forward(batch, "train")
        loss.backward()
        opt.step()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n = 4
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    ka = prof.key_averages()
    rows = sorted(ka, key=lambda e: -getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)))
    print("op, calls per step, self device ms per step")
    for e in rows[:28]:
        t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
        print("%-60s %7.1f %8.3f" % (e.key[:60], e.count / n, t / n / 1e3))
    print("\n-- by source line (aten ops only)")
    ks = prof.key_averages(group_by_stack_n=6)
    agg = {}
    for e in ks:
        if not e.key.startswith("aten::"):
            continue
        t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
        if t <= 0:
            continue
        src = next((s for s in e.stack if "insmos_amd" in s or "bench.py" in s), e.stack[0] if e.stack else "?")
        a = agg.setdefault((src.strip()[-90:], e.key), [0, 0.0])
        a[0] += e.count
        a[1] += t
    for (src, key), (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:30]:
        print("%8.3f ms %6.1f calls  %-22s %s" % (t / n / 1e3, c / n, key, src))


if __name__ == "__main__":
    main()

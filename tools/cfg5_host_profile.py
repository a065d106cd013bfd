#!/usr/bin/env python3
"""tools/cfg5_host_profile.py -- is the training step (bench.py --config cfg5) bound by the GPU or by the host?  Times the step three
ways on one box: as benchmarked; with the host made to wait for the GPU after every step only (the same thing, for reference); and
the HOST time alone (wall time of step() up to the point where everything is enqueued, no synchronisation) -- then cProfile of four
steps, top functions by own time.  If enqueue time ~ step time, the step is host-bound and kernel speed-ups cannot show."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

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
        t_f = time.perf_counter()
        loss.backward()
        t_b = time.perf_counter()
        opt.step()
        return t_f, t_b

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n = 8
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print("as benchmarked: %.2f ms per step" % ((time.perf_counter() - t0) / n * 1e3))
    enq, fwd, bwd, opt_t, tot = [], [], [], [], []
    for _ in range(n):
        torch.cuda.synchronize()
        a = time.perf_counter()
        t_f, t_b = step()
        b = time.perf_counter()
        torch.cuda.synchronize()
        c = time.perf_counter()
        enq.append(b - a); fwd.append(t_f - a); bwd.append(t_b - t_f); opt_t.append(b - t_b); tot.append(c - a)
    ms = lambda v: 1e3 * float(np.median(v))
    print("one step at a time: host returns after %.2f ms (forward %.2f incl. its read-backs, backward %.2f, optimiser %.2f), GPU done "
          "after %.2f ms" % (ms(enq), ms(fwd), ms(bwd), ms(opt_t), ms(tot)))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()))


if __name__ == "__main__":
    main()

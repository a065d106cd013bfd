#!/usr/bin/env python3
"""tools/train_step_bench.py -- time of one FULL training step (InsMOSTrainer: MotionNet + 3D branch in train mode, the
four losses, backward, Adam) on the S0 window, with the per-kernel-class breakdown from the library's own profiler.

    python tools/train_step_bench.py [n_az] [B]             # default 1886 (S0, 1.2 M points), B = 1 window per step
    python tools/train_step_bench.py 1886 4                 # cfg-5's batch: 4 windows per step in one set of launches per branch
    INSMOS_TRAIN_SEQUENTIAL=1 python tools/train_step_bench.py 1886 4   # the same batch walked item by item (the reference's loop)
    INSMOS_DW_MFMA=1 python tools/train_step_bench.py       # the MFMA dW kernel instead of the LDS one
    INSMOS_TRAIN_BF16=1 python tools/train_step_bench.py    # conv forward / d/dx with bf16 operands, fp32 accumulate (opt-in)
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.synth import make_labels  # noqa: E402
from insmos_amd.train_unet import InsMOSTrainer  # noqa: E402


def main():
    n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 1886
    NB = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = P.default_cfg()
    wins = bench.load_windows(list(range(NB)), n_az)
    w = wins[0]
    rng = np.random.default_rng(0)
    M = 40
    batch = []
    for wi, ww in enumerate(wins):
        gt_boxes = np.zeros((1, M, 8), np.float32)
        gt_boxes[0, :, 0] = rng.uniform(-55, 55, M)
        gt_boxes[0, :, 1] = rng.uniform(-45, 45, M)
        gt_boxes[0, :, 2] = rng.uniform(-1.5, -0.5, M)
        gt_boxes[0, :, 3:6] = rng.uniform([1.5, 0.6, 1.2], [4.5, 2.0, 1.8], (M, 3))
        gt_boxes[0, :, 6] = rng.uniform(-3.1, 3.1, M)
        gt_boxes[0, :, 7] = rng.integers(1, 4, M)
        batch.append({"past_point_clouds": torch.from_numpy(ww).cuda(),
                      "past_labels": [None, torch.from_numpy(make_labels(ww[ww[:, 4] == 0], seed=wi)).cuda()],
                      "gt_boxes": torch.from_numpy(gt_boxes).cuda()})
    tr = InsMOSTrainer(cfg, P.random_state_dict(cfg, 0, cls_bias=-2.0, box_w_std=0.05))
    opt = torch.optim.Adam(list(tr.params.values()), lr=float(cfg["TRAIN"]["LR"]),
                           weight_decay=float(cfg["TRAIN"].get("WEIGHT_DECAY", 0.0)))  # models/models.py:188-193
    lib = tr.motion.engine.lib
    steps, warm = 8, 4   # (the caching allocator needs a few steps to settle: with 2 + 4 the timed steps still hit hipMalloc)


    def run_steps(n):  # noqa: E306
        for _ in range(n):
            opt.zero_grad(set_to_none=True)
            out = tr.forward(batch, "train")
            out[0].backward()
            opt.step()
        return out


    run_steps(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, tb, _, _ = run_steps(steps)                      # the timed steps: per-kernel profiler OFF
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    lib.insmos_prof_reset()
    lib.insmos_prof_enable(1)                              # a second pass for the per-kernel breakdown (HIP events per launch)
    run_steps(steps)
    torch.cuda.synchronize()
    prof = bench.read_profile(lib)
    lib.insmos_prof_enable(0)
    if os.environ.get("INSMOS_TRAIN_CPROFILE"):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        run_steps(2)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
        pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    dev_ms = sum(ms for ms, _ in prof.values()) / steps
    print(f"full training step, {NB} window(s) of {len(w)} points{' (item by item)' if os.environ.get('INSMOS_TRAIN_SEQUENTIAL') == '1' else ''}: "
          f"{dt * 1e3:.1f} ms wall = {dt * 1e3 / NB:.1f} ms per window (profiler off), {dev_ms:.1f} ms in the library's kernels "
          f"(loss {float(loss.detach()):.4f}, {tb[0]}); "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB; dW kernel {os.environ.get('INSMOS_DW_KERNEL', '2 (default)')} "
          f"bf16_convs={tr.bf16_convs}", flush=True)
    for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"    {k:24s} {ms / steps:9.3f} ms/step  {cnt // steps:6d} launches/step")


if __name__ == "__main__":   # (bench.load_windows ray-casts missing windows in spawned worker processes)
    main()

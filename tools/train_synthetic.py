#!/usr/bin/env python3
"""tools/train_synthetic.py -- a minimal training loop on synthetic windows, to show how the pieces go together
(NOT the reference's scripts/train.py: no Lightning, no dataset, no augmentation -- those are out of scope).

    python tools/train_synthetic.py --steps 20 [--n-az 472] [--out /tmp/insmos_synth.ckpt]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_synthetic.py --steps 20

Optimiser and schedule as models/models.py:188-193 (Adam(lr, weight_decay) + StepLR(step_size=LR_EPOCH, gamma=LR_DECAY));
gradients are exchanged with insmos_amd.ddp.BucketedGradReducer (one RCCL all-reduce per 8 MB bucket) when WORLD_SIZE > 1.
The trained tensors go back into a Lightning-shaped checkpoint that insmos_amd.models.InsMOSNet / predict_mos load.
(Written at the end of round 1 after the GPU budget was spent: the pieces it calls are tested, this loop itself has not run.)"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import params as P  # noqa: E402
from insmos_amd.models import save_checkpoint  # noqa: E402
from insmos_amd.synth import make_labels, make_window  # noqa: E402
from insmos_amd.train_unet import InsMOSTrainer  # noqa: E402


def synth_item(seed, n_az, dev):
    w = make_window(seed=seed, n_scans=10, n_az=n_az)
    rng = np.random.default_rng(seed)
    cur = w[w[:, 4] == 0]
    m = 12
    gt = np.zeros((1, m, 8), np.float32)
    gt[0, :, 0:2] = cur[rng.integers(0, len(cur), m), :2]
    gt[0, :, 2] = rng.uniform(-1.5, -0.5, m)
    gt[0, :, 3:6] = rng.uniform([1.5, 0.6, 1.2], [4.5, 2.0, 1.8], (m, 3))
    gt[0, :, 6] = rng.uniform(-3.1, 3.1, m)
    gt[0, :, 7] = rng.integers(1, 4, m)
    return {"past_point_clouds": torch.from_numpy(w).to(dev), "gt_boxes": torch.from_numpy(gt).to(dev),
            "past_labels": [torch.from_numpy(make_labels(cur, seed=seed)).to(dev)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--steps-per-epoch", type=int, default=5, help="the LR schedule steps once per this many iterations")
    ap.add_argument("--n-az", type=int, default=472)
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = f"cuda:{local}"
    cfg = P.default_cfg()
    sd = P.random_state_dict(cfg, 0, cls_bias=-2.0, box_w_std=0.05)
    tr = InsMOSTrainer(cfg, sd, dev)
    opt = torch.optim.Adam(list(tr.params.values()), lr=float(cfg["TRAIN"]["LR"]), weight_decay=float(cfg["TRAIN"]["WEIGHT_DECAY"]))
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=int(cfg["TRAIN"]["LR_EPOCH"]), gamma=float(cfg["TRAIN"]["LR_DECAY"]))
    reducer = tr.make_reducer() if world > 1 else None
    for step in range(args.steps):
        batch = [synth_item(1000 * rank + step, args.n_az, dev)]
        opt.zero_grad(set_to_none=True)
        loss, tb, _, _ = tr.forward(batch, "train")
        loss.backward()
        if reducer is not None:
            reducer.reduce()
        opt.step()
        if (step + 1) % args.steps_per_epoch == 0:
            sched.step()              # StepLR counts (pseudo-)epochs, as models/models.py:188-193 does
        if rank == 0:
            print(f"step {step}: loss {float(loss.detach()):.4f}  " + "  ".join(f"{k} {v:.4f}" for k, v in tb[0].items()), flush=True)
    if rank == 0 and args.out:
        out = dict(sd)
        out.update(tr.unet.export_state_dict())
        M = P.ME_PREFIX
        for k, v in tr.motion.params.items():   # MotionNetTrainer keeps the reference's names apart from the ".bn" level
            name = M + (k if k.endswith(".kernel") or k == "final.bias" else k.rsplit(".", 1)[0] + ".bn." + k.rsplit(".", 1)[1])
            out[name] = v.detach().cpu().numpy().reshape(np.asarray(sd[name]).shape)
        for k, v in tr.motion.buffers.items():
            name = M + k.rsplit(".", 1)[0] + ".bn." + k.rsplit(".", 1)[1]
            out[name] = v.detach().cpu().numpy()
        save_checkpoint(args.out, cfg, out)
        print("wrote", args.out)


if __name__ == "__main__":
    main()

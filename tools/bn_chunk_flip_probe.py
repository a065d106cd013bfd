#!/usr/bin/env python3
"""tools/bn_chunk_flip_probe.py -- why tests/test_train_unet.py::test_batched_training_step_equals_the_item_by_item_walk moved with the
BatchNorm chunk length: the batched and the item-by-item step of that test, their point logits row by row, the predicted boxes of
both, and the MotionNet outputs (no boxes involved) -- a rounding-order difference that flips a discrete decision (a box at the
score / NMS threshold feeds other instance columns to the decoder) shows as a FEW rows far off and the rest within 1e-5."""
import copy, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from insmos_amd import params as P
from insmos_amd.synth import make_labels, make_window
from insmos_amd.train_unet import InsMOSTrainer
from test_train_unet import _gt_boxes

rng = np.random.default_rng(21)
cfg = copy.deepcopy(P.default_cfg())
cfg["MODEL"]["USE_MOTION_LOSS"] = True
sd = P.random_state_dict(cfg, 2, cls_bias=-1.0, box_w_std=0.05)
batch = []
for s, (ns, az) in zip((3, 4, 5), ((3, 96), (4, 80), (3, 128))):
    w = make_window(seed=s, n_scans=ns, n_az=az)
    batch.append({"past_point_clouds": torch.from_numpy(w).cuda(),
                  "past_labels": [None, torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=s)).cuda()],
                  "gt_boxes": torch.from_numpy(_gt_boxes(rng)).cuda()})
for chunk in (None, "1024"):
    if chunk:
        os.environ["INSMOS_BN_CHUNK"] = chunk
    else:
        os.environ.pop("INSMOS_BN_CHUNK", None)
    tr_b, tr_s = InsMOSTrainer(cfg, sd), InsMOSTrainer(cfg, sd)
    mb = tr_b.motion.forward_windows([b["past_point_clouds"] for b in batch])
    ms = [tr_s.motion.forward_windows([b["past_point_clouds"]])[0] for b in batch]
    print("chunk", chunk or "by width", "| MotionNet logits, batched vs alone, max |d| per window:",
          [float((a - b).abs().max()) for a, b in zip(mb, ms)])
    tr_b, tr_s = InsMOSTrainer(cfg, sd), InsMOSTrainer(cfg, sd)
    _, _, _, pb = tr_b.forward(batch, "train")
    ps = [tr_s.forward([it], "train")[3][0] for it in batch]
    for w, (a, b) in enumerate(zip(pb, ps)):
        d = (a - b).abs().max(1).values
        print("   window %d: %d point rows, max |d| %.2e, rows > 1e-4: %d, rows > 1e-5: %d" % (w, len(d), float(d.max()), int((d > 1e-4).sum()), int((d > 1e-5).sum())))

"""oracle/calibrate.py -- TEST INFRASTRUCTURE (like everything under oracle/: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it).  Helpers for end-to-end parity: a seeded checkpoint whose detection head actually fires.

With default-initialised random weights the CenterHead never exceeds SCORE_THRESH (bias -log 99,
center_head.py:60-63), so the instance branch would go untested.  `detecting_state_dict`
  1. makes every BEV BatchNorm shift negative, so an empty BEV region stays exactly zero through the
     ReLU stack (the empty-map background, including the zero-padded borders, then scores exactly the bias);
  2. calibrates the classification head with the ORACLE so that a few hundred cells near occupied
     voxels pass the threshold, placing the threshold in the widest gap of the sorted scores -- the
     discrete steps (threshold, top-k order, NMS) are then well-posed for a CPU-vs-GPU comparison;
  3. gives the box head car-sized boxes near the ground so the point-in-box features are exercised."""
import math

import numpy as np

from insmos_amd import params as P
from oracle import ref_model as M


def detecting_state_dict(cfg, window, seed=0, target=(300, 700), margin=1.0, box_w_std=0.05):
    sd = P.random_state_dict(cfg, seed, cls_bias=0.0, box_w_std=box_w_std)
    rng = np.random.default_rng(seed + 77)
    B = M.UN_P + "bev_backbone."
    stems = [B + "blocks.0.2"] + [B + f"blocks.0.{5 + 3 * k}" for k in range(cfg["MODEL"]["BACKBONE_2D"]["LAYER_NUMS"][0])]
    stems.append(B + "deblocks.0.1")
    for st in stems:
        c = sd[st + ".bias"].shape
        sd[st + ".bias"] = -np.abs(rng.normal(0, 0.002, c)).astype(np.float32)
        sd[st + ".running_mean"] = np.abs(rng.normal(0, 0.002, c)).astype(np.float32)
    sd[M.UN_P + "center_head.conv_box.bias"] = np.array(
        [0.0, 0.0, -1.0, math.log(4.0), math.log(2.0), math.log(1.6), 0.3, 0.5], np.float32)
    _, _, dbg = M.forward_window(sd, cfg, window, want_debug=True)
    best = dbg["unet"]["cls"].max(axis=1)  # bias 0, scale 1: exactly 0 on the empty background
    s = np.sort(best[best > 0])[::-1]
    lo, hi = target
    hi = min(hi, len(s) - 1)
    assert hi > lo, "window too empty to calibrate a detecting head"
    gaps = s[lo:hi] - s[lo + 1:hi + 1]
    r = lo + int(np.argmax(gaps))
    q = 0.5 * (s[r] + s[r + 1])  # threshold in the widest gap: r+1 cells pass
    scale = margin / q
    key_w = M.UN_P + "center_head.conv_cls.weight"
    sd[key_w] = (sd[key_w] * scale).astype(np.float32)
    ncls = cfg["MODEL"]["DENSE_HEAD"]["NUM_CLASS"]
    sd[M.UN_P + "center_head.conv_cls.bias"] = np.full(ncls, math.log(0.1 / 0.9) - margin, np.float32)
    return sd

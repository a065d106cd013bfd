#!/usr/bin/env python3
"""tools/bf16x3_experiment.py -- the split-bf16 x 3 convolution EXPERIMENT (insmos_conv_precision(3)), end to end.

NOT the product path and never a benchmark line: the library's inference path is exact fp32.  This script answers two
questions about replacing every fp32 MFMA group (4 x v_mfma_f32_16x16x4_f32, 128 cycles per 16-channel chunk) by three bf16
MFMAs on a (hi, lo) split of both operands (48 cycles): what it does to the outputs of the full forward, and what it does to
windows/s.  Same W S0 windows, same weights, same launch-set shape as bench.py.

    python tools/bf16x3_experiment.py [W=24] [steps=8] [out.json]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402


def timed(model, batch, steps):
    for _ in range(2):
        out = model.forward(batch, "test")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model.forward(batch, "test")
    torch.cuda.synchronize()
    return out, steps * len(batch) / (time.perf_counter() - t0)


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cfg = P.default_cfg()
    model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
    wins = [torch.from_numpy(w).cuda() for w in bench.load_windows(list(range(W)), 1886)]
    bench.calibrate_head(model, wins[0], 1500, cache="/tmp/insmos_bench_calibration.json", tag="rank0_az1886_c1500")
    batch = [{"past_point_clouds": p} for p in wins]
    eng = model.model.engine
    res = {"windows": W, "steps": steps, "windows_per_launch": model.model.windows_per_launch,
           "launch_sets_in_flight": model.model.windows_in_flight}
    out0, rate0 = timed(model, batch, steps)
    eng.set_conv_precision(3)
    try:
        out3, rate3 = timed(model, batch, steps)
    finally:
        eng.set_conv_precision(0)
    out0b, rate0b = timed(model, batch, steps)
    res["fp32_windows_per_s"] = round(rate0, 1)
    res["fp32_windows_per_s_again"] = round(rate0b, 1)
    res["bf16x3_windows_per_s"] = round(rate3, 1)
    res["fp32_unchanged_after_experiment"] = all(torch.equal(a, b) for a, b in zip(out0[2], out0b[2]))
    # forward(batch, 'test') -> ([[pred_dict]] per window, recall dicts, logits per window)
    dl, flips, npts, nb0, nb3, scale = [], 0, 0, 0, 0, []
    for a, b in zip(out0[2], out3[2]):
        a, b = a.float(), b.float()
        if a.shape != b.shape:
            res.setdefault("shape_mismatch", 0)
            res["shape_mismatch"] += 1
            continue
        dl.append(float((a - b).abs().max()))
        scale.append(float(a.abs().mean()))
        flips += int((a.argmax(1) != b.argmax(1)).sum())
        npts += a.shape[0]
    for a, b in zip(out0[0], out3[0]):
        nb0 += len(a[0]["pred_boxes"])
        nb3 += len(b[0]["pred_boxes"])
    res["max_abs_logit_diff"] = max(dl) if dl else None
    res["mean_abs_logit"] = float(np.mean(scale)) if scale else None
    res["label_flips"] = flips
    res["points"] = npts
    res["boxes_fp32"], res["boxes_bf16x3"] = nb0, nb3
    res["note"] = ("EXPERIMENT: split-bf16 x 3 on every convolution with Cin % 16 == 0 and a neighbour table (incl. the dense BEV "
                   "kernel); Cin 4/8 layers, 1x1 layers and the fused deconv+heads stay fp32.  Not the product path.")
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

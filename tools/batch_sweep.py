#!/usr/bin/env python3
"""tools/batch_sweep.py -- windows/s of InsMOS_Model.forward over the launch-set shape: windows per launch set x launch
sets in flight, same 16 S0 windows, inputs resident (the bench.py step without the bookkeeping).

    python tools/batch_sweep.py [W=16] [steps=6] [grid="1x1,1x4,..."]     # one line per (windows_per_launch x in_flight)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402

def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    grid = [(1, 1), (1, 4), (2, 1), (2, 2), (2, 4), (4, 1), (4, 2), (4, 3), (8, 1), (8, 2)]
    if len(sys.argv) > 3:
        grid = [tuple(int(v) for v in g.split("x")) for g in sys.argv[3].split(",")]
    cfg = P.default_cfg()
    model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
    wins = [torch.from_numpy(w).cuda() for w in bench.load_windows(list(range(W)), 1886)]
    bench.calibrate_head(model, wins[0], 1500, cache="/tmp/insmos_bench_calibration.json", tag="rank0_az1886_c1500")
    batch = [{"past_point_clouds": p} for p in wins]
    ref = None
    for wpl, fl in grid:
        model.model.windows_per_launch, model.model.windows_in_flight = wpl, fl
        try:
            for _ in range(2):
                out = model.forward(batch, "test")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                out = model.forward(batch, "test")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        except Exception as e:  # noqa: BLE001
            print(f"wpl {wpl} in_flight {fl}: FAILED {e}", flush=True)
            continue
        same = ""
        if ref is None:
            ref = out
        else:
            same = " identical to (1,1): %s" % all(torch.equal(a, b) for a, b in zip(ref[2], out[2]))
        print(f"windows_per_launch {wpl} sets_in_flight {fl}: {steps * W / dt:8.1f} windows/s  ({1000 * dt / (steps * W):.3f} ms/window)"
              f"{same}  [INSMOS_SPLIT_TAP_MOD={os.environ.get('INSMOS_SPLIT_TAP_MOD', '1')}]", flush=True)


if __name__ == "__main__":
    main()

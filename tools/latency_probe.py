#!/usr/bin/env python3
"""tools/latency_probe.py -- latency of ONE window through the native runner (Engine.forward_window, nothing else in flight) for each
setting of the second stream (insmos_forward_streams: bit 0 level-0 table, 1 the 3D coordinate phase, 2 inv_conv_out, 3 one-hots)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from insmos_amd import params as P
from insmos_amd.models import InsMOSNet


def main():
    cfg = P.default_cfg()
    model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
    pts = torch.from_numpy(bench.load_window(0, 1886)).cuda()
    bench.calibrate_head(model, pts, 1500, cache="/tmp/insmos_bench_calibration.json", tag="rank0_az1886_c1500")
    eng = model.model.engine
    lib = eng.lib
    for mask in (0, 15, 1, 2, 4, 8, 3, 0, 15):
        lib.insmos_forward_streams(mask)
        for _ in range(3):
            eng.forward_window(pts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            eng.forward_window(pts)
        torch.cuda.synchronize()
        print("second-stream mask %2d: %.3f ms per window" % (mask, (time.perf_counter() - t0) / n * 1e3), flush=True)
    lib.insmos_forward_streams(-1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/b1_ab.py -- A/B of the single-window (B = 1) latency knobs inside one process: INSMOS_LEVEL_CHAIN (the level-down chain,
one read-back instead of four) x insmos_bev_cosplit (128-channel BEV layers as two 64-channel workgroups per patch).  Each setting
twice, interleaved; ms per Engine.forward_window with nothing else in flight."""
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
    lib.insmos_forward_streams(15)
    ref = None
    for rep in range(2):
        for chain, cosplit in ((0, 0), (1, 0), (0, 1024), (1, 1024)):
            os.environ["INSMOS_LEVEL_CHAIN"] = str(chain)
            lib.insmos_bev_cosplit(cosplit)
            for _ in range(4):
                out = eng.forward_window(pts)
            torch.cuda.synchronize()
            logits = out[0].clone()   # (logits (Ncur, 3), pred dict)
            if logits is not None:
                if ref is None:
                    ref = logits
                assert torch.equal(ref, logits), "outputs moved with the launch-shape knobs"
            t0 = time.perf_counter()
            n = 30
            for _ in range(n):
                eng.forward_window(pts)
            torch.cuda.synchronize()
            print("level chain %d, bev cosplit %4d: %.3f ms per window%s" % (chain, cosplit, (time.perf_counter() - t0) / n * 1e3,
                                                                              "" if logits is None else " (logits bit-equal)"), flush=True)
    lib.insmos_bev_cosplit(-1)
    lib.insmos_forward_streams(-1)
    os.environ.pop("INSMOS_LEVEL_CHAIN", None)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/two_stream_probe.py -- a measurement script, not a test (pytest.ini keeps collection to tests/): throughput of W windows in flight (one host thread + HIP stream + Engine each)."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.engine import Engine  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402


def main():
    cfg = P.default_cfg()
    sd = P.random_state_dict(cfg, seed=0)
    pts = torch.from_numpy(bench.load_window(0, 1886)).cuda()
    model = InsMOSNet(cfg, state_dict=sd).cuda(0).eval()
    bench.calibrate_head(model, pts, 1500)
    sd = model.state_dict()
    STEPS = 30
    for W in (1, 2, 3, 4):
        engines = [Engine(cfg, sd, "cuda:0") for _ in range(W)]
        streams = [torch.cuda.Stream() for _ in range(W)]

        def worker(i, n):
            with torch.cuda.stream(streams[i]):
                for _ in range(n):
                    engines[i].forward_window(pts)
                streams[i].synchronize()
        for i in range(W):
            worker(i, 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(i, STEPS)) for i in range(W)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"W={W}: {W * STEPS / dt:.1f} windows/s  ({dt / (W * STEPS) * 1e3:.2f} ms per window)", flush=True)


if __name__ == "__main__":
    main()

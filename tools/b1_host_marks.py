#!/usr/bin/env python3
"""tools/b1_host_marks.py -- host timeline of one window alone (B = 1) through the native runner: when the host finished enqueueing
each section and when each count read-back returned (insmos_forward_host_marks; run with INSMOS_HOST_MARKS=1), median of 20 windows,
no profiler attached.  Next to it: the total latency and the GPU-side time of the same call (HIP events around it)."""
import ctypes, os, sys
os.environ.setdefault("INSMOS_HOST_MARKS", "1")
import numpy as np
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
    buf = ctypes.create_string_buffer(4096)
    for _ in range(5):
        eng.forward_window(pts)
    torch.cuda.synchronize()
    rows, names = [], None
    for _ in range(20):
        eng.forward_window(pts)
        torch.cuda.synchronize()
        assert lib.insmos_forward_host_marks(buf, 4096) == 0
        items = [it.rsplit(":", 1) for it in buf.value.decode().split(";") if it]
        names = [a for a, _ in items]
        rows.append([float(b) for _, b in items])
    med = np.median(np.array(rows), 0)
    prev = 0.0
    for nm, t in zip(names, med):
        print("%8.1f us  (+%6.1f)  %s" % (t, t - prev, nm))
        prev = t
    lib.insmos_forward_streams(-1)


if __name__ == "__main__":
    main()

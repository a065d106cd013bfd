#!/usr/bin/env python3
"""tools/driver_bench.py -- I/O-inclusive rate of the predict_mos counterpart: a synthetic SemanticKITTI-style sequence
(n scans of ~120k points written as .bin + poses/calib) is predicted end to end: disk read, H2D, pose alignment and stacking,
forward, output stage, D2H and the three output files per scan."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import params as P  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402
from insmos_amd.predict_mos import predict_sequence  # noqa: E402
from insmos_amd.synth import make_scan, make_world  # noqa: E402

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 80
root = tempfile.mkdtemp(prefix="insmos_seq_", dir=os.environ.get("INSMOS_BENCH_DIR"))
seq = os.path.join(root, "08")
os.makedirs(os.path.join(seq, "velodyne"))
rng = np.random.default_rng(0)
world = make_world(rng, 60, 30)
t0 = time.perf_counter()
base = []
for i in range(12):  # 12 distinct scans, reused cyclically (generating every scan would dominate the GPU-box time)
    p = make_scan(rng, 1.0 * i, 1886, world)
    base.append(np.hstack([p, rng.uniform(0, 1, (len(p), 1)).astype(np.float32)]).astype(np.float32))
for i in range(n_scans):
    base[i % 12].tofile(os.path.join(seq, "velodyne", "%06d.bin" % i))
open(os.path.join(seq, "poses.txt"), "w").write("".join("1 0 0 0 0 1 0 0 0 0 1 %.3f\n" % (1.0 * (i % 12)) for i in range(n_scans)))
open(os.path.join(seq, "calib.txt"), "w").write("Tr: 0 -1 0 0 0 0 -1 0 1 0 0 0\n")
print(f"wrote {n_scans} scans in {time.perf_counter() - t0:.1f} s", flush=True)
cfg = P.default_cfg()
model = InsMOSNet(cfg, seed=0).cuda(0).eval()
out = os.path.join(root, "preb_out")
import insmos_amd.predict_mos as PM  # noqa: E402
real_write = PM.write_outputs


import collections  # noqa: E402
from insmos_amd import data as D  # noqa: E402
T = collections.defaultdict(float)


def timed(obj, name, key):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        T[key] += time.perf_counter() - t
        return r
    setattr(obj, name, g)


timed(D.SequenceWindows, "window", "window() incl. waiting for the reader")
timed(D.SequenceWindows, "_load", "reader thread: file -> pinned -> H2D issue")
timed(PM, "output_stage", "output_stage")
timed(PM.OutputWriter, "submit", "writer.submit")
timed(type(model), "forward", "model.forward (launch sets)")


def run(tag):
    T.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = predict_sequence(model, cfg, seq, 8, out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"[{tag}] {n} scans in {dt:.2f} s: {n / dt:.1f} scans/s ({dt / n * 1e3:.2f} ms per scan)", flush=True)
    for k, v in T.items():
        print(f"[   {k}: {v:.2f} s]", flush=True)


with torch.no_grad():
    predict_sequence(model, cfg, seq, 8, out, limit=40)  # warm-up (short histories, first windows, arenas)
    run("full: read + align + forward + output stage + D2H + 3 files per scan")
    PM.write_outputs = lambda *a, **k: [a[4].cpu(), a[5].cpu()]
    run("no file writes (D2H only)")
    PM.write_outputs = lambda *a, **k: None
    run("no D2H, no files")
    PM.write_outputs = real_write
    # the same 4 windows, resident, forwarded repeatedly (what bench.py times)
    sw = D.SequenceWindows(cfg, seq, None, "cuda:0")
    ws = [sw.window(20 + i)[0].clone() for i in range(4)]
    batch = [{"past_point_clouds": w} for w in ws]
    for _ in range(3):
        model.forward(batch, "test")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model.forward(batch, "test")
    torch.cuda.synchronize()
    print(f"[resident windows: {(time.perf_counter() - t0) / 40 * 1e3:.2f} ms per window; counts {model.model.engine.last_counts}]", flush=True)
shutil.rmtree(root, ignore_errors=True)

#!/usr/bin/env python3
"""tools/b1_trace.py -- workload + analysis for the single-window (B = 1) timeline: run under
`rocprofv3 --kernel-trace --output-format csv -d DIR -o b1 -- python tools/b1_trace.py`, then `python tools/b1_trace.py --analyze DIR`:
per window, the wall time between the first and last kernel, the union of kernel-busy time (any stream), the idle gaps, and the top
kernels by time."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workload():
    import torch
    import bench
    from insmos_amd import params as P
    from insmos_amd.models import InsMOSNet
    cfg = P.default_cfg()
    model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
    pts = torch.from_numpy(bench.load_window(0, 1886)).cuda()
    bench.calibrate_head(model, pts, 1500, cache="/tmp/insmos_bench_calibration.json", tag="rank0_az1886_c1500")
    eng = model.model.engine
    for _ in range(4):
        eng.forward_window(pts)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        eng.forward_window(pts)
        torch.cuda.synchronize()
    print("5 windows: %.3f ms each" % ((time.perf_counter() - t0) * 200))


def analyze(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    # windows = separated by gaps > 200 us after the warm-up; take the last 5 groups
    groups, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - max(x[1] for x in cur[-50:]) > 150_000:
            groups.append(cur); cur = []
        cur.append(r)
    groups.append(cur)
    groups = [g for g in groups if len(g) > 100][-5:]
    for g in groups[-2:]:
        t0, t1 = g[0][0], max(x[1] for x in g)
        busy, hi = 0, t0
        gaps = []
        for s, e, _ in g:
            if s > hi:
                gaps.append(s - hi)
            if e > hi:
                busy += e - max(s, hi); hi = e
        ksum = sum(e - s for s, e, _ in g)
        print("window: %d kernels, wall %.1f us, busy(union) %.1f us, kernel sum %.1f us, idle %.1f us in %d gaps (>%d us: %d, sum %.1f us)" % (
            len(g), (t1 - t0) / 1e3, busy / 1e3, ksum / 1e3, sum(gaps) / 1e3, len(gaps), 10, sum(1 for x in gaps if x > 10_000),
            sum(x for x in gaps if x > 10_000) / 1e3))
    # the gaps of the last window, largest first: what ran before and after each (the read-backs and other host round trips)
    g = groups[-1]
    hi, prev = g[0][0], g[0]
    gl = []
    for r in g:
        if r[0] > hi:
            gl.append((r[0] - hi, (hi - g[0][0]) / 1e3, prev[2].split("(")[0][-44:], r[2].split("(")[0][-44:]))
        if r[1] > hi:
            hi, prev = r[1], r
    for d, at, a, b in sorted(gl, reverse=True)[:24]:
        print("  gap %6.1f us at %7.1f us  after %-44s before %s" % (d / 1e3, at, a, b))
    agg = {}
    for s, e, n in g:
        k = n.split("(")[0][-60:]
        a = agg.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
        print("  %8.1f us %4d  %s" % (t / 1e3, c, k))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
        analyze(sys.argv[2])
    else:
        workload()

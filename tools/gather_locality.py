#!/usr/bin/env python3
"""tools/gather_locality.py -- how local are the gathers of the 81-tap 4D layers?  (CPU, oracle tables; design input for the
LDS-staged small-channel kernel)  For every block of R consecutive output rows (rows are in (t, Morton) order)
and every time offset dt, over the 27 spatial taps: the share of PRESENT neighbour rows that fall inside ONE window of W
consecutive input rows (window placed at the block's own row range shifted to the time slice t + dt), the window size a block
would need to catch 90 / 99 % of them, and the number of DISTINCT neighbour rows per block."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd.synth import make_window  # noqa: E402
from oracle import ref_ops as R  # noqa: E402


def main():
    n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 1886
    w = make_window(seed=0, n_scans=10, n_az=n_az)
    c0, k0, _ = R.me_quantize(w[:, [0, 1, 2, 4]], [0.1, 0.1, 0.1, 0.1])
    print("level 0 voxels", len(c0))
    for lvl in (1, 2, 0):
        if lvl == 0:
            c, k = c0, k0
        else:
            c, k, _ = R.me_stride_down(c0, k0, lvl)
        n = len(c)
        offs = R.me_kernel_offsets([3, 3, 3, 3], [1 << lvl] * 3 + [1])
        nbr = R.me_nbr(c, k, offs)                                   # (81, n)
        t = c[:, 3]
        # first row of every time slice
        ts = np.unique(t)
        first = {int(v): int(np.searchsorted(t, v)) for v in ts}
        for Rb in (64, 256):
            nb = (n + Rb - 1) // Rb
            for dt in (-1, 0, 1):
                taps = [kk for kk in range(81) if offs[kk][3] == dt]
                sub = nbr[taps]                                      # (27, n)
                hits = {W: 0 for W in (128, 256, 512, 1024, 2048)}
                tot = 0
                distinct = []
                span90 = []
                for b in range(0, nb, max(1, nb // 400)):            # a sample of blocks
                    r0, r1 = b * Rb, min(n, (b + 1) * Rb)
                    v = sub[:, r0:r1].ravel()
                    v = v[v >= 0]
                    if len(v) == 0:
                        continue
                    tot += len(v)
                    distinct.append(len(np.unique(v)))
                    # the block's own position inside its time slice, transported to slice t + dt
                    tb = int(t[r0])
                    if (tb + dt) not in first:
                        continue
                    nxt = first.get(tb + 1, n) - first[tb]
                    rel = (r0 - first[tb]) / max(nxt, 1)
                    size_d = (first.get(tb + dt + 1, n) - first[tb + dt])
                    centre = first[tb + dt] + int(rel * size_d) + Rb // 2
                    med = int(np.median(v))
                    for W in hits:
                        lo = med - W // 2
                        hits[W] += int(((v >= lo) & (v < lo + W)).sum())
                    d = np.sort(np.abs(v - med))
                    span90.append(2 * int(d[int(0.9 * (len(d) - 1))]))
                print(f"level {lvl} rows {n} block {Rb} dt {dt:+d}: distinct rows/block mean {np.mean(distinct):.0f} max {np.max(distinct)}; "
                      f"gathers/block mean {tot / max(len(distinct), 1):.0f}; window hit-rate " +
                      " ".join(f"W{W}={hits[W] / max(tot, 1):.3f}" for W in hits) +
                      f"; 90%-span median {int(np.median(span90))} p90 {int(np.percentile(span90, 90))}", flush=True)


if __name__ == "__main__":
    main()

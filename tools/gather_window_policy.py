#!/usr/bin/env python3
"""tools/gather_window_policy.py -- design input for the LDS-staged 81-tap kernel (csrc/spconv_lds.hip): per 64-row block and time
offset, how many PRESENT neighbour rows fall outside a staged window of CAP rows for a few placements of the window, and how the
per-(block, dt) miss counts are distributed (they go to an overflow area of OVF rows; more than that -> the block falls back to
global gathers).  CPU, oracle tables of the S0 window."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd.synth import make_window  # noqa: E402
from oracle import ref_ops as R  # noqa: E402


def main():
    w = make_window(seed=0, n_scans=10, n_az=1886)
    c0, k0, _ = R.me_quantize(w[:, [0, 1, 2, 4]], [0.1, 0.1, 0.1, 0.1])
    for lvl in (1, 2, 0):
        c, k = (c0, k0) if lvl == 0 else R.me_stride_down(c0, k0, lvl)[:2]
        n = len(c)
        offs = R.me_kernel_offsets([3, 3, 3, 3], [1 << lvl] * 3 + [1])
        nbr = R.me_nbr(c, k, offs).astype(np.int64)
        Rb = 64
        nb = n // Rb
        for dt in (-1, 0, 1):
            taps = [kk for kk in range(81) if offs[kk][3] == dt]
            centre = [kk for kk in taps if (offs[kk][:3] == 0).all()][0]
            sub = nbr[taps][:, :nb * Rb].reshape(27, nb, Rb)                    # (tap, block, row)
            cen = nbr[centre][:nb * Rb].reshape(nb, Rb)
            valid = sub >= 0
            big = np.iinfo(np.int64).max
            vmin = np.where(valid, sub, big).min(axis=(0, 2))
            vmax = np.where(valid, sub, -1).max(axis=(0, 2))
            cmin = np.where(cen >= 0, cen, big).min(1)
            cmax = np.where(cen >= 0, cen, -1).max(1)
            has = vmax >= 0
            for CAP in (256, 512, 1024):
                # placement: centred on the centre tap's own neighbours when the block has any, else on the smallest index
                mid = np.where(cmax >= 0, (np.minimum(cmin, vmax) + cmax) // 2, vmin)
                lo = np.maximum(mid - CAP // 2, 0)
                miss = valid & ((sub < lo[None, :, None]) | (sub >= (lo + CAP)[None, :, None]))
                mcount = miss.sum(axis=(0, 2))
                tot = valid.sum()
                full = ((vmax - vmin) < CAP) | ~has
                print(f"level {lvl} dt {dt:+d} CAP {CAP}: span<CAP for {full.mean():.3f} of blocks; centre-tap window: "
                      f"lane miss rate {miss.sum() / tot:.3f}, blocks with 0 misses {np.mean(mcount == 0):.3f}, "
                      f"<=32 {np.mean(mcount <= 32):.3f}, <=64 {np.mean(mcount <= 64):.3f}, <=128 {np.mean(mcount <= 128):.3f}, "
                      f"max {mcount.max()}; miss taps/block mean {(miss.any(2).sum(0)).mean():.1f}", flush=True)


if __name__ == "__main__":
    main()

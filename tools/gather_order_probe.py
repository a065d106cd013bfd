#!/usr/bin/env python3
"""tools/gather_order_probe.py -- review item 3 of round 4, CPU part (no GPU minutes): would ANOTHER ROW ORDER of a 4D level make
the gathers of the 81-tap C <= 16 layers touch fewer cache lines?  DESIGN.md 3.9's probes say those layers are bound by the lines a
gather instruction touches (64 lanes, one neighbour row each in the row-lane kernel; 16 rows x 64 B in the MFMA tiles), not by HBM.

For the oracle's level-L tables of one synthetic window and a set of candidate row orders (the canonical (t, Morton) order and
private permutations of it, applied to output rows AND input rows alike, as section 3.5 does for the 3D levels) this prints

  * lines/gather(64): mean number of distinct 128-byte lines among the present neighbour rows of a 64-row tile under one tap
    (row pitch 32 B = 8 channels: 4 rows per line; 64 B = 16 channels: 2 rows per line) -- the row-lane kernel's unit;
  * lines/gather(16): the same for a 16-row group (MFMA tiles, 64 B chunks);
  * x-adjacency: the share of present (row, dx = +1 neighbour) pairs whose neighbour is the NEXT row -- what an (index, 3-bit mask)
    x-triple table entry would need;
  * active (16-row group, tap) slots per group -- the MFMA passes a tile pays (an order must not lose here what it wins above).

    python tools/gather_order_probe.py [n_az=944] [levels=1,2]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd.synth import make_window  # noqa: E402
from oracle import ref_ops as R  # noqa: E402


def orders(c, lvl):
    """name -> permutation `old_of_new` of the level's rows.  c: (n, 4) int32 [x, y, z, t] in canonical (t, Morton) order."""
    n = len(c)
    x, y, z, t = (c[:, i].astype(np.int64) >> (lvl if i < 3 else 0) for i in range(4))
    x, y, z = x - x.min(), y - y.min(), z - z.min()
    out = {"(t, Morton) canonical": np.arange(n)}

    def lex(*keys):   # last key = most significant (np.lexsort convention reversed for readability)
        return np.lexsort(tuple(reversed(keys)))
    out["(t, z, y, x) raster"] = lex(t, z, y, x)
    out["(t, y, z, x) raster"] = lex(t, y, z, x)
    for bs in (8, 16, 32):
        b = bs.bit_length() - 1
        out[f"(t, yx-block {bs}x{bs} raster, z, y, x)"] = lex(t, y >> b, x >> b, z, y, x)
        out[f"(t, yx-block {bs}x{bs} raster, y, z, x)"] = lex(t, y >> b, x >> b, y, z, x)
    # Morton over blocks, raster inside: keep the canonical order of the block's FIRST row as the block rank
    for bs in (4, 8):
        b = bs.bit_length() - 1
        blk = (t << 48) | ((z >> b) << 32) | ((y >> b) << 16) | (x >> b)
        _, first = np.unique(blk, return_index=True)
        rank_of_blk = dict()
        ub, inv = np.unique(blk, return_inverse=True)
        blk_rank = first[inv]                     # canonical position of the block's first row = Morton rank of the block
        out[f"(t, Morton of {bs}^3 blocks, z, y, x inside)"] = lex(blk_rank, z, y, x)
    return out


def measure(nbr, offs, old_of_new, pitch_rows_per_line):
    """nbr (K, n) in canonical rows.  Returns lines/gather for 64- and 16-row tiles, x-adjacency, active slots per 16-row group."""
    n = nbr.shape[1]
    new_of_old = np.empty(n, np.int64)
    new_of_old[old_of_new] = np.arange(n)
    K = nbr.shape[0]
    res = {}
    tab = nbr[:, old_of_new]                                   # output rows permuted
    tab = np.where(tab >= 0, new_of_old[np.maximum(tab, 0)], -1)   # entries renamed
    kx = [k for k in range(K) if tuple(offs[k]) == (offs[:, 0].max(), 0, 0, 0)][0]
    pres = tab[kx] >= 0
    res["x_adj"] = float((tab[kx][pres] == np.nonzero(pres)[0] + 1).mean()) if pres.any() else 0.0
    for tile in (64, 16):
        nt = n // tile
        lines = tab[:, :nt * tile].reshape(K, nt, tile)
        present = lines >= 0
        ln = np.where(present, lines // pitch_rows_per_line, -1)
        ln.sort(axis=2)
        distinct = ((ln[:, :, 1:] != ln[:, :, :-1]) & (ln[:, :, 1:] >= 0)).sum(axis=2) + (ln[:, :, 0] >= 0)
        active = present.any(axis=2)
        res[f"lines{tile}"] = float(distinct[active].mean())
        res[f"rows{tile}"] = float(present.sum(axis=2)[active].mean())
        if tile == 16:
            res["slots16"] = float(active.sum(axis=0).mean())
    return res


def main():
    n_az = int(sys.argv[1]) if len(sys.argv) > 1 else 944
    levels = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2"])]
    w = make_window(seed=0, n_scans=10, n_az=n_az)
    c0, k0, _ = R.me_quantize(w[:, [0, 1, 2, 4]], [0.1, 0.1, 0.1, 0.1])
    print(f"window n_az={n_az}: {len(w)} points, level-0 voxels {len(c0)}")
    for lvl in levels:
        c, k, _ = R.me_stride_down(c0, k0, lvl) if lvl else (c0, k0, None)
        offs = R.me_kernel_offsets([3, 3, 3, 3], [1 << lvl] * 3 + [1])
        nbr = R.me_nbr(c, k, offs).astype(np.int64)
        n = len(c)
        print(f"\nlevel {lvl}: {n} rows, {float((nbr >= 0).sum()) / n:.1f} valid taps per row")
        for rpl, what in ((4, "8 channels (32-B rows)"), (2, "16 channels (64-B rows)")):
            print(f"  -- {what}")
            print("  %-52s %9s %9s %9s %9s %7s %8s" % ("row order", "lines/64", "rows/64", "lines/16", "rows/16", "x-adj", "slots/16"))
            for name, perm in orders(c, lvl).items():
                m = measure(nbr, offs, perm, rpl)
                print("  %-52s %9.2f %9.2f %9.2f %9.2f %7.3f %8.2f" % (name, m["lines64"], m["rows64"], m["lines16"], m["rows16"],
                                                                    m["x_adj"], m["slots16"]))


if __name__ == "__main__":
    main()

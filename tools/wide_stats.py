#!/usr/bin/env python3
"""tools/wide_stats.py -- MFMA passes issued / useful on the 3D layers of ONE window in the DEVICE's row order: valid (row, tap)
pairs, active (16-row group, tap) slots (what the output-stationary tiles execute), the 32-row union (csrc/spconv_wide.hip before
its per-group skip) and per-tap compaction of an R-row block's rows into dense groups of 16 (the tap-compacted form).

    python tools/wide_stats.py [out.txt]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.engine import NbrTable  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    cfg = P.default_cfg()
    model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
    win = torch.from_numpy(bench.load_windows([0], 1886)[0]).cuda()
    bench.calibrate_head(model, win, 1500, cache="/tmp/insmos_bench_calibration.json", tag="rank0_az1886_c1500")
    eng = model.model.engine
    eng.forward_window(win, native=False)
    lines = ["layer,K,cin,cout,rows,pairs_per_row,slots16_x16_per_pair,union32_x32_per_pair," +
             ",".join("compact_R%d" % r for r in (64, 128, 256, 512)) + ",taps_per_block_R128,groups_per_tapvisit_R128"]
    seen = {}
    for nbr, n_out, layer, row0 in eng._conv_log:
        if nbr is None or layer.K < 27:
            continue
        tab = nbr.nbr if isinstance(nbr, NbrTable) else nbr
        key = (tab.data_ptr(), row0)
        if key not in seen:
            pres = (tab[:, row0:n_out] >= 0)            # (K, n)
            K, n = pres.shape
            pairs = int(pres.sum().item())
            ng = n // 16
            a16 = pres[:, :ng * 16].reshape(K, ng, 16).any(2)
            p16 = int(pres[:, :ng * 16].sum().item())
            n32 = n // 32
            a32 = pres[:, :n32 * 32].reshape(K, n32, 32).any(2)
            p32 = int(pres[:, :n32 * 32].sum().item())
            comp = []
            extra = (0.0, 0.0)
            for R in (64, 128, 256, 512):
                nb = n // R
                if nb == 0:
                    comp.append(float("nan"))
                    continue
                cnt = pres[:, :nb * R].reshape(K, nb, R).sum(2)
                grp = (cnt + 15) // 16
                comp.append(float(grp.sum().item()) * 16 / max(1, int(cnt.sum().item())))
                if R == 128:
                    extra = (float((cnt > 0).sum(0).float().mean().item()), float(grp.sum().item()) / max(1, int((cnt > 0).sum().item())))
            seen[key] = (n, pairs / max(1, n), float(a16.sum().item()) * 16 / max(1, p16), float(a32.sum().item()) * 32 / max(1, p32), comp, extra)
        n, ppr, s16, u32, comp, extra = seen[key]
        lines.append("%s,%d,%d,%d,%d,%.2f,%.3f,%.3f,%s,%.1f,%.2f" % (layer.name, layer.K, layer.cin, layer.cout, n, ppr, s16, u32,
                                                                  ",".join("%.3f" % c for c in comp), extra[0], extra[1]))
    txt = "\n".join(lines)
    print(txt)
    if out:
        with open(out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()

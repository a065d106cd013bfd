#!/usr/bin/env python3
"""tools/regroup_order_probe.py -- which ROW ORDER inside a window makes 16-row groups share taps best?  For the SubMConv3d tables
of UNet levels 2..4 of one S0 window (device order = signature-sorted 4096-row blocks): issued / useful MFMA passes of the 16-row
tiles under alternative sort keys of the 27-bit tap signature (the row permutation only permutes the table's columns)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import params as P  # noqa: E402
from insmos_amd.engine import NbrTable  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402


def ratio(pres):
    K, n = pres.shape
    ng = n // 16
    p = pres[:, :ng * 16]
    return float(p.reshape(K, ng, 16).any(2).sum().item()) * 16 / max(1, int(p.sum().item()))


def main():
    cfg = P.default_cfg()
    model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
    win = torch.from_numpy(bench.load_windows([0], 1886)[0]).cuda()
    bench.calibrate_head(model, win, 1500, cache="/tmp/insmos_bench_calibration.json", tag="rank0_az1886_c1500")
    eng = model.model.engine
    eng.forward_window(win, native=False)
    done = set()
    for nbr, n_out, layer, row0 in eng._conv_log:
        if layer.name not in ("conv2.1.0", "conv3.1.0", "conv4.1.0", "conv1.0.0"):
            continue
        tab = nbr.nbr if isinstance(nbr, NbrTable) else nbr
        if tab.data_ptr() in done:
            continue
        done.add(tab.data_ptr())
        pres = tab[:, :n_out] >= 0
        K, n = pres.shape
        if isinstance(nbr, NbrTable) and nbr.mask16 is not None:
            m = nbr.mask16.view(torch.int32).reshape(-1, 4).to(torch.int64) & 0xFFFFFFFF
            ks = torch.arange(K, device=m.device)
            has = ((m[:, ks >> 5] >> (ks & 31)) & 1).bool().T          # (K, groups)
            pres_m = pres & has.repeat_interleave(16, dim=1)[:, :n]
            print(layer.name, "rows", n, "unmasked pairs", int(pres.sum()), "masked pairs", int(pres_m.sum()))
            pres = pres_m
        w = (1 << torch.arange(K, device=pres.device, dtype=torch.int64))
        sig = (pres.to(torch.int64) * w[:, None]).sum(0)             # (n,)
        freq = pres.float().mean(1)                                  # (K,)
        print(layer.name, "pairs/row %.2f" % (float(pres.sum()) / n), "device order: issued/useful %.3f" % ratio(pres))
        print("   tap frequency:", " ".join("%.2f" % f for f in freq.tolist()))

        def by(key, name):
            order = torch.argsort(key, stable=True)
            print("   %-44s %.3f" % (name, ratio(pres[:, order])))
            return order

        by(sig, "numeric signature (whole window)")
        gray = sig.clone()
        # rank in Gray order: inverse Gray code of the signature
        g = sig.clone()
        sh = 1
        while sh < 32:
            g = g ^ (g >> sh)
            sh <<= 1
        by(g, "inverse-Gray rank of the signature")
        pc = pres.sum(0).to(torch.int64)
        by(pc * (1 << 27) + sig, "popcount, then signature")
        # bits re-ordered: most balanced taps (frequency nearest 0.5) most significant
        for nm, score in (("balanced taps high", -(freq - 0.5).abs()), ("frequent taps high", freq), ("rare taps high", -freq)):
            o = torch.argsort(score, descending=True)                # most significant first
            key = torch.zeros(n, dtype=torch.int64, device=pres.device)
            for i, k in enumerate(o.tolist()):
                key |= pres[k].to(torch.int64) << (K - 1 - i)
            by(key, "signature, " + nm)
            gk = key.clone()
            sh = 1
            while sh < 32:
                gk = gk ^ (gk >> sh)
                sh <<= 1
            by(gk, "inverse-Gray of (" + nm + ")")
        # greedy: repeatedly take the unassigned row with the smallest signature, then the 15 rows nearest in Hamming distance
        if n <= 20000:
            P_ = pres.T.float()                                      # (n, K)
            left = torch.ones(n, dtype=torch.bool, device=pres.device)
            order = []
            pcf = P_.sum(1)
            while int(left.sum()) > 0:
                idx = torch.nonzero(left)[:, 0]
                seed = idx[torch.argmax(pcf[idx])]
                # cost of adding row r to the group = taps of r outside the union (start: seed's taps) + taps of union r lacks * 0
                union = P_[seed].clone()
                grp = [int(seed)]
                left[seed] = False
                for _ in range(15):
                    idx = torch.nonzero(left)[:, 0]
                    if idx.numel() == 0:
                        break
                    extra = (P_[idx] * (1 - union)).sum(1) * 16 - (P_[idx] * union).sum(1)    # new slots cost 16 lanes; shared taps are gains
                    j = idx[torch.argmin(extra)]
                    grp.append(int(j))
                    union = torch.maximum(union, P_[j])
                    left[j] = False
                order.extend(grp)
            print("   %-44s %.3f" % ("greedy grouping (whole window, reference only)", ratio(pres[:, torch.tensor(order, device=pres.device)])))


if __name__ == "__main__":
    main()

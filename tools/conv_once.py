#!/usr/bin/env python3
"""tools/conv_once.py -- run chosen S0 layers alone at launch-set size (the window's real tables replicated B times), a few
launches each: the target of a `rocprofv3 --pmc ...` pass (tools/pmc_conv_layers.sh), which then holds the counters of exactly
these launches, in this order.

    python tools/conv_once.py layer[,layer...] [B=8] [reps=3]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import _lib, params as P  # noqa: E402
from insmos_amd.engine import NbrTable  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402


def main():
    layers = sys.argv[1].split(",")
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lib = _lib.load()
    cfg = P.default_cfg()
    model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
    pts = torch.from_numpy(bench.load_window(0, 1886)).cuda()
    bench.calibrate_head(model, pts, 1500, cache="/tmp/insmos_bench_calibration.json", tag="rank0_az1886_c1500")
    eng = model.model.engine
    eng.dense_bev_kernel = False          # BEV layers through the table kernel too (they are in the log with their table)
    eng.forward_window(pts, native=False)
    torch.cuda.synchronize()
    log = {l.name: (nbr, n, l, r0) for (nbr, n, l, r0) in eng._conv_log if nbr is not None}
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name in layers:
        nbr, n_out, layer, row0 = log[name]
        tab = nbr.nbr if isinstance(nbr, NbrTable) else nbr
        mask = nbr.mask16 if isinstance(nbr, NbrTable) else None
        n_in = int(tab.max().item()) + 1
        n_o16 = (n_out + 15) // 16 * 16
        big = torch.full((tab.shape[0], n_o16 * B), -1, dtype=torch.int32, device="cuda")
        for b in range(B):
            big[:, b * n_o16:b * n_o16 + n_out] = torch.where(tab >= 0, tab + b * n_in, tab)
        bmask = mask.view(torch.int32).reshape(-1, 4).repeat(B, 1).contiguous() if mask is not None else None
        x = torch.randn((n_in * B, layer.cin), device="cuda")
        out = torch.empty((n_o16 * B, layer.cout), device="cuda")
        pairs = int((big >= 0).sum().item())
        torch.cuda.synchronize()
        for _ in range(reps):
            rc = lib.insmos_sparse_conv(x.data_ptr(), n_in * B, layer.cin, layer.cin, big.data_ptr(),
                                        bmask.data_ptr() if bmask is not None else None, layer.K, n_o16 * B, layer.w.data_ptr(),
                                        layer.b.data_ptr(), out.data_ptr(), layer.cout, layer.cout, None, 0, 0, 0, 1, st)
            assert rc == 0, rc
        torch.cuda.synchronize()
        print(f"LAYER {name} K{layer.K} {layer.cin}->{layer.cout} rows {n_o16 * B} pairs {pairs} gflop {pairs * layer.flops_per_pair / 1e9:.3f}",
              flush=True)


if __name__ == "__main__":
    main()

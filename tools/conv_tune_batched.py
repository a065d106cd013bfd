#!/usr/bin/env python3
"""tools/conv_tune_batched.py -- tile-shape sweep of insmos_sparse_conv at LAUNCH-SET sizes: the real neighbour tables of
one S0 window (step path), replicated B times (rows and neighbour indices shifted per copy -- what a launch set of B such
windows hands the kernel), timed with the model-chosen variant and with every forced unsplit (COT, JT, ring) variant.

    python tools/conv_tune_batched.py [B=4] [layer,layer,...]
"""
import ctypes
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import _lib, params as P  # noqa: E402
from insmos_amd.engine import NbrTable  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402

DEFAULT = ["block1.0.conv1", "block7.0.conv1", "block6.0.conv1", "block3.0.conv2", "block8.0.conv1", "conv2.1.0", "conv3.1.0",
           "conv4.1.0", "conv_up_m4.0", "conv_up_m3.0", "conv_up_instance_block.0", "inv_conv4.0", "bev1", "bev0"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    layers = sys.argv[2].split(",") if len(sys.argv) > 2 else DEFAULT
    lib = _lib.load()
    cfg = P.default_cfg()
    model = InsMOSNet(cfg, state_dict=P.random_state_dict(cfg, seed=0)).cuda().eval()
    pts = torch.from_numpy(bench.load_window(0, 1886)).cuda()
    bench.calibrate_head(model, pts, 1500, cache="/tmp/insmos_bench_calibration.json", tag="rank0_az1886_c1500")
    eng = model.model.engine
    eng.forward_window(pts, native=False)
    torch.cuda.synchronize()
    log = {l.name: (nbr, n, l, r0) for (nbr, n, l, r0) in eng._conv_log if nbr is not None}
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name in layers:
        nbr, n_out, layer, row0 = log[name]
        tab = nbr.nbr if isinstance(nbr, NbrTable) else nbr
        mask = nbr.mask16 if isinstance(nbr, NbrTable) else None
        n_in = int(tab.max().item()) + 1
        n_o16 = (n_out + 15) // 16 * 16        # copies start on a 16-row boundary (their tap masks stay valid)
        K = tab.shape[0]
        big = torch.full((K, n_o16 * B), -1, dtype=torch.int32, device="cuda")
        for b in range(B):
            big[:, b * n_o16:b * n_o16 + n_out] = torch.where(tab >= 0, tab + b * n_in, tab)
        bmask = None
        if mask is not None:
            bmask = mask.view(torch.int32).reshape(-1, 4).repeat(B, 1).contiguous()
        N_out, N_in = n_o16 * B, n_in * B
        x = torch.randn((N_in, layer.cin), device="cuda")
        out = torch.empty((N_out, layer.cout), device="cuda")
        pairs = int((big[:, row0 * 0:] >= 0).sum().item())
        gf = 2.0 * pairs * (layer.flops_per_pair / 2) / 1e9

        def run():
            rc = lib.insmos_sparse_conv(x.data_ptr(), N_in, layer.cin, layer.cin, big.data_ptr(),
                                        bmask.data_ptr() if bmask is not None else None, K, N_out, layer.w.data_ptr(),
                                        layer.b.data_ptr(), out.data_ptr(), layer.cout, layer.cout, None, 0, 0, 0, 1, st)
            assert rc == 0, rc

        def timed(reps=10):
            for _ in range(2):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1000.0 / reps

        lib.insmos_debug_conv_force(0, 0, 0)
        base = timed()
        res = []
        if layer.cin % 16 == 0:
            ntile = (layer.cout + 15) // 16
            for cot, jt, ring in itertools.product((1, 2, 4, 8), (1, 2, 4), (2, 3)):
                if ntile % cot or (cot == 8 and jt == 4 and ring == 3):
                    continue
                lib.insmos_debug_conv_force(cot, jt, ring)
                res.append((timed(6), cot, jt, ring))
            lib.insmos_debug_conv_force(0, 0, 0)
        res.sort()
        print(f"{name} K{K} {layer.cin}->{layer.cout} rows {N_out} (x{B}): chosen {base:.1f} us ({gf / base * 1e3:.1f} TF/s) | unsplit best "
              + ", ".join(f"({c},{j},R{r}) {t:.1f}" for t, c, j, r in res[:5]) + (f" | worst {res[-1][0]:.1f}" if res else ""), flush=True)


if __name__ == "__main__":
    main()

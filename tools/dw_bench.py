#!/usr/bin/env python3
"""tools/dw_bench.py -- the three d/dW kernels of insmos_sparse_conv_backward_weight on layer shapes of the S0 training step
(random tables at LiDAR-like fill; one window).  Prints microseconds per call per kernel (2 = row-compacting MFMA kernel,
1 = first MFMA design, 0 = LDS slabs).

    python tools/dw_bench.py [out.csv]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import _lib  # noqa: E402

SHAPES = [  # (name, K, cin, cout, rows, fill)
    ("4D C8 81-tap", 81, 8, 8, 172000, 0.24), ("4D C16", 81, 16, 16, 65000, 0.30), ("4D 16->32", 81, 16, 32, 18500, 0.33),
    ("4D C32", 81, 32, 32, 16000, 0.35), ("4D 48->32", 81, 48, 32, 38000, 0.35), ("3D C16 L1", 27, 16, 16, 36000, 0.45),
    ("3D C32 L2", 27, 32, 32, 25000, 0.45), ("3D C64 L3", 27, 64, 64, 10100, 0.45), ("3D C128 L4", 27, 128, 128, 5200, 0.45),
    ("3D 256->128", 27, 256, 128, 5200, 0.45), ("3D 144->128", 27, 144, 128, 5200, 0.45), ("BEV 3x3 C128", 9, 128, 128, 18750, 0.97),
    ("BEV 256->128", 9, 256, 128, 18750, 0.97), ("1x1 16->3", 1, 16, 3, 36000, 1.0),
]


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    rows = []
    rng = np.random.default_rng(0)
    for name, K, cin, cout, n, fill in SHAPES:
        nbr = None
        if K > 1:
            # spatially coherent: runs of 16 rows share their taps, neighbours are nearby rows
            base = np.arange(n, dtype=np.int64)[None, :] + rng.integers(-40, 40, size=(K, 1))
            t = np.clip(base, 0, n - 1).astype(np.int32)
            keep = np.repeat(rng.uniform(size=(K, (n + 15) // 16)) < min(1.0, fill * 1.5), 16, axis=1)[:, :n] & \
                (rng.uniform(size=(K, n)) < 1 / 1.5 if fill < 0.9 else np.ones((K, n), bool))
            t[~keep] = -1
            nbr = torch.from_numpy(t).to(dev)
        x = torch.randn((n, cin), device=dev)
        dy = torch.randn((n, cout), device=dev)
        dw = torch.empty((K, cin, cout), device=dev)
        res, outs = [], []
        for mode in (2, 1, 0):
            _lib.check(lib.insmos_debug_dw_kernel(mode), "mode")
            ws = torch.empty(int(lib.insmos_sparse_conv_backward_weight_ws_floats(n, K, cin, cout)), device=dev)

            def call():
                _lib.check(lib.insmos_sparse_conv_backward_weight(x.data_ptr(), n, cin, cin, dy.data_ptr(), cout, cout,
                                                                  nbr.data_ptr() if nbr is not None else None, K, n,
                                                                  dw.data_ptr(), 0, ws.data_ptr(), st), "dw")
            for _ in range(2):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 5 * 1000)
            outs.append(dw.clone())
        lib.insmos_debug_dw_kernel(2)
        err = max(float((outs[0] - outs[2]).abs().max()), float((outs[1] - outs[2]).abs().max())) / float(outs[2].abs().mean() + 1e-9)
        pairs = int((nbr >= 0).sum()) if nbr is not None else n
        rows.append((name, K, cin, cout, n, pairs, *res, err))
        print(f"{name:16s} K={K:3d} {cin:3d}->{cout:3d} rows={n:7d} pairs={pairs:9d}  rows-kernel {res[0]:8.1f} us  mfma-v1 {res[1]:8.1f} us  "
              f"lds {res[2]:8.1f} us   {2 * pairs * cin * cout / res[0] / 1e6:6.1f} TFLOP/s  rel.diff {err:.1e}", flush=True)
    print("total: rows-kernel %.2f ms, mfma-v1 %.2f ms, lds %.2f ms" % tuple(sum(r[6 + i] for r in rows) / 1000 for i in range(3)))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("layer,K,cin,cout,rows,pairs,us_rows_kernel,us_mfma_v1,us_lds,rel_diff\n")
            for r in rows:
                f.write(",".join(str(v) for v in r) + "\n")


if __name__ == "__main__":
    main()

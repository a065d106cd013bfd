#!/usr/bin/env python3
"""tools/conv_concurrency.py -- how much MFMA throughput do concurrent launches of one conv layer reach?
(dense BEV 3x3 128->128, 18750 sites; and a masked 27-tap 128->128 layer with 6717 rows) on 1, 2, 3, 4 streams."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import _lib  # noqa: E402
from insmos_amd.engine import ConvLayer  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(0)
D = torch.device("cuda:0")
H, W = 125, 150
nb = torch.empty((9, H * W), dtype=torch.int32, device=D)
lib.insmos_dense_nbr2d(H, W, nb.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
cases = [("bev 9x128->128 n18750", nb, None, H * W, 128, 128)]
n = 6717
a = rng.integers(0, n, size=(27, n)).astype(np.int32)
a[rng.uniform(size=a.shape) > 0.57] = -1
v = np.zeros((27, (n + 15) // 16 * 16), bool); v[:, :n] = a >= 0
any16 = v.reshape(27, -1, 16).any(2)
m = np.zeros((any16.shape[1], 4), np.uint32)
for k in range(27):
    m[:, k >> 5] |= any16[k].astype(np.uint32) << np.uint32(k & 31)
cases.append(("L4 27x128->128 n6717", torch.from_numpy(a).to(D), torch.from_numpy(m.view(np.int32)).to(D), n, 128, 128))
for name, nbr, mask, n_out, cin, cout in cases:
    K = nbr.shape[0]
    layer = ConvLayer(lib, (rng.normal(size=(K, cin, cout)) * 0.05).astype(np.float32), None, cin, cout, D)
    pairs = int((nbr >= 0).sum())
    flops = 2.0 * pairs * cin * cout
    for S in (1, 2, 3, 4, 6):
        streams = [torch.cuda.Stream() for _ in range(S)]
        xs = [torch.randn((n_out, cin), device=D) for _ in range(S)]
        outs = [torch.empty((n_out, cout), device=D) for _ in range(S)]

        def burst(reps):
            for r in range(reps):
                for i, st in enumerate(streams):
                    rc = lib.insmos_sparse_conv(xs[i].data_ptr(), n_out, cin, cin, nbr.data_ptr(),
                                                mask.data_ptr() if mask is not None else None, K, n_out, layer.w.data_ptr(),
                                                layer.b.data_ptr(), outs[i].data_ptr(), cout, cout, None, 0, 0, 0, 1,
                                                ctypes.c_void_p(st.cuda_stream))
                    assert rc == 0
        burst(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        burst(30)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}: {S} streams: {dt / (30 * S) * 1e6:7.1f} us per launch, {flops * 30 * S / dt / 1e12:6.1f} TFLOP/s", flush=True)

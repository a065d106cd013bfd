#!/usr/bin/env python3
"""tools/conv_tune.py -- sweep tile shape / ring depth of insmos_sparse_conv on representative layer shapes
(dense BEV 3x3, small sparse 27-tap levels) on the GPU box.  Tuning aid, not part of the product path."""
import ctypes
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import _lib  # noqa: E402
from insmos_amd.engine import ConvLayer  # noqa: E402

D = "cuda:0"


def time_conv(lib, layer, x, nbr, mask, n_out, reps=20):
    out = torch.empty((n_out, layer.cout), device=D)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        rc = lib.insmos_sparse_conv(x.data_ptr(), x.shape[0], x.stride(0), layer.cin, nbr.data_ptr(),
                                    mask.data_ptr() if mask is not None else None, layer.K, n_out, layer.w.data_ptr(),
                                    layer.b.data_ptr(), out.data_ptr(), layer.cout, layer.cout, None, 0, 0, 0, 1, st)
        assert rc == 0, rc
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / reps


def main():
    lib = _lib.load()
    rng = np.random.default_rng(0)
    shapes = []
    H, W = 125, 150
    nb = torch.empty((9, H * W), dtype=torch.int32, device=D)
    lib.insmos_dense_nbr2d(H, W, nb.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    shapes.append(("bev 9x128->128 n18750", nb, None, H * W, H * W, 128, 128))
    shapes.append(("bev0 9x256->128 n18750", nb, None, H * W, H * W, 256, 128))
    for name, n, cin, cout, dens in (("L4 27x128->128 n6717", 6717, 128, 128, 0.57), ("L3 27x64->64 n12564", 12564, 64, 64, 0.45),
                                     ("L2 27x32->32 n30867", 30867, 32, 32, 0.42), ("m4 27x256->128 n6717", 6717, 256, 128, 0.57),
                                     ("ME b6 81x48->32 n83805", 83805, 48, 32, 0.27), ("ME b7 81x32->16 n211916", 211916, 32, 16, 0.24),
                                     ("ME b7c2 81x16->16 n211916", 211916, 16, 16, 0.24), ("ME b8 81x16->8 n468007", 468007, 16, 8, 0.2),
                                     ("U1 27x32->16 n42280", 42280, 32, 16, 0.23), ("U1 27x16->16 n42280", 42280, 16, 16, 0.23)):
        K = 81 if "81x" in name else 27
        a = rng.integers(0, n, size=(K, n)).astype(np.int32)
        # spatially coherent occupancy: whole 16-row groups share their taps
        grp = rng.uniform(size=(K, (n + 15) // 16)) < min(1.0, dens * 1.6)
        keep = np.repeat(grp, 16, axis=1)[:, :n] & (rng.uniform(size=(K, n)) < 0.62)
        a[~keep] = -1
        ng = (n + 15) // 16
        v = np.zeros((K, ng * 16), bool); v[:, :n] = a >= 0
        any16 = v.reshape(K, ng, 16).any(2)
        m = np.zeros((ng, 4), np.uint32)
        for k in range(K):
            m[:, k >> 5] |= (any16[k].astype(np.uint32) << np.uint32(k & 31))
        shapes.append((name, torch.from_numpy(a).to(D), torch.from_numpy(m.view(np.int32)).to(D), n, n, cin, cout))
    for name, nbr, mask, n_in, n_out, cin, cout in shapes:
        K = nbr.shape[0]
        layer = ConvLayer(lib, (rng.normal(size=(K, cin, cout)) * 0.05).astype(np.float32), None, cin, cout, torch.device(D))
        x = torch.randn((n_in, cin), device=D)
        lib.insmos_debug_conv_force(0, 0, 0)
        base = time_conv(lib, layer, x, nbr, mask, n_out)
        res = []
        for cot, jt, ring in itertools.product((1, 2, 4, 8), (1, 2, 4), (2, 3, 4)):
            if (cout // 16) % cot or (cot == 8 and (ring == 4 or (jt == 4 and ring == 3))):
                continue
            lib.insmos_debug_conv_force(cot, jt, ring)
            res.append((time_conv(lib, layer, x, nbr, mask, n_out), cot, jt, ring))
        lib.insmos_debug_conv_force(0, 0, 0)
        res.sort()
        pairs = int((nbr >= 0).sum())
        gf = 2.0 * pairs * cin * cout / 1e9
        print(f"{name}: model-chosen {base:.1f} us ({gf / base * 1e3:.1f} TF/s) | best " +
              ", ".join(f"({c},{j},R{r}) {t:.1f}" for t, c, j, r in res[:6]) + f" | worst {res[-1][0]:.1f}", flush=True)


if __name__ == "__main__":
    main()

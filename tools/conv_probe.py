#!/usr/bin/env python3
"""tools/conv_probe.py -- which stream bounds the sparse-conv kernel?  Re-times chosen S0 layers (real neighbour
tables captured from one forward) with probe builds of the kernel: dbg bit 0 = no weight loads, bit 1 = no gathers,
bit 2 = no MFMAs.  Tuning aid, not part of the product path."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from insmos_amd import _lib, params as P  # noqa: E402
from insmos_amd.engine import NbrTable  # noqa: E402
from insmos_amd.models import InsMOSNet  # noqa: E402

REP = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # replicate the window's tables REP times (launch-set size)
LAYERS = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else [
    "block1.0.conv1", "block8.0.conv1", "block8.0.conv2", "block7.0.conv1", "block6.0.conv1", "block3.0.conv2",
    "conv2.1.0", "conv3.1.0", "conv4.1.0", "conv_up_m4.0", "bev1", "bev0"]
DBGS = [0, 1, 2, 3, 4, 7]

lib = _lib.load()
cfg = P.default_cfg()
sd = P.random_state_dict(cfg, seed=0)
pts = torch.from_numpy(bench.load_window(0, 1886)).cuda()
model = InsMOSNet(cfg, state_dict=sd).cuda(0).eval()
bench.calibrate_head(model, pts, 1500)
eng = model.model.engine
model.forward([{"past_point_clouds": pts}], "test")
torch.cuda.synchronize()
log = {l.name: (nbr, n, l) for (nbr, n, l, _r0) in eng._conv_log if nbr is not None}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print("%-18s %3s %4s %4s %7s | " % ("layer", "K", "cin", "cout", "n_out") + " ".join("%8s" % s for s in
      ("full", "noW", "noB", "mfma", "noMFMA", "idx", "allK", "allKmfma", "unif", "unifmfma")))
for name in LAYERS:
    nbr, n_out, layer = log[name]
    tab = nbr.nbr if isinstance(nbr, NbrTable) else nbr
    mask = nbr.mask16 if isinstance(nbr, NbrTable) else None
    n_in = int(tab.max().item()) + 1
    if REP > 1:  # REP copies of the window back to back, each on a 16-row boundary (tap masks stay valid)
        n_o16 = (n_out + 15) // 16 * 16
        big = torch.full((tab.shape[0], n_o16 * REP), -1, dtype=torch.int32, device="cuda")
        for b in range(REP):
            big[:, b * n_o16:b * n_o16 + n_out] = torch.where(tab >= 0, tab + b * n_in, tab)
        tab = big
        if mask is not None:
            mask = mask.view(torch.int32).reshape(-1, 4).repeat(REP, 1).contiguous()
        n_out, n_in = n_o16 * REP, n_in * REP
    x = torch.randn((n_in, layer.cin), device="cuda")
    out = torch.empty((n_out, layer.cout), device="cuda")
    res = []
    if mask is not None:  # balance probe: every tile walks all K taps (equal work); compare with K / mean active taps
        m = mask.view(torch.int32).cpu().numpy().view("uint32")
        pc = np.unpackbits(m.view(np.uint8), axis=1).sum(1)
        act = float(pc.mean())
        na = int(round(act))
        um = np.zeros((m.shape[0], 4), np.uint32)  # uniform probe: every group walks the first round(act) taps
        for w in range(4):
            bits = min(32, max(0, na - 32 * w))
            um[:, w] = (1 << bits) - 1 if bits < 32 else 0xFFFFFFFF
        umask = torch.from_numpy(um.view(np.int32)).cuda()
    else:
        act = float(layer.K)
        umask = None
    for dbg, use_mask in [(d, True) for d in DBGS] + [(0, False), (3, False), (0, 2), (3, 2)]:
        lib.insmos_debug_conv_force(0, 0, 16 * dbg)

        def run():
            rc = lib.insmos_sparse_conv(x.data_ptr(), n_in, layer.cin, layer.cin, tab.data_ptr(),
                                        (umask.data_ptr() if use_mask == 2 and umask is not None else mask.data_ptr()) if (mask is not None and use_mask) else None, layer.K, n_out, layer.w.data_ptr(),
                                        layer.b.data_ptr(), out.data_ptr(), layer.cout, layer.cout, None, 0, 0, 0, 1, st)
            assert rc == 0, rc
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1000 / 20)
    lib.insmos_debug_conv_force(0, 0, 0)
    print("%-18s %3d %4d %4d %7d | " % (name, layer.K, layer.cin, layer.cout, n_out) + " ".join("%8.1f" % r for r in res)
          + " | act/grp %5.1f  allK->scaled full %7.1f mfma %7.1f" % (act, res[-4] * act / layer.K, res[-3] * act / layer.K),
          flush=True)

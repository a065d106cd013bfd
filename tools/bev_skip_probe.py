#!/usr/bin/env python3
"""tools/bev_skip_probe.py -- time of the dense BEV kernel against the constant-region-skipping variant on a synthetic map:
(a) occupancy like the S0 window's (14 % of the sites, clustered), (b) everything constant, (c) nothing constant."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import _lib, params as P
from insmos_amd.engine import ConvLayer

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B, H, W = 8, 125, 150
rng = np.random.default_rng(0)
occ = np.zeros((B, H, W), bool)
for b in range(B):
    for _ in range(60):
        y, x = rng.integers(0, H), rng.integers(0, W)
        occ[b, max(0, y - 2):y + 3, max(0, x - 4):x + 5] |= rng.uniform(size=occ[b, max(0, y - 2):y + 3, max(0, x - 4):x + 5].shape) < 0.7
print("occupancy %.3f" % occ.mean())
ys, xs = np.nonzero(occ.reshape(B * H, W))
coords = torch.from_numpy(np.stack([ys // H, np.zeros_like(ys), ys % H, xs], 1).astype(np.int32)).to(dev)
dist = torch.empty(B * H * W, dtype=torch.uint8, device=dev)
ws = torch.empty(int(lib.insmos_bev_distance_map_ws_bytes(B, H, W)), dtype=torch.uint8, device=dev)
_lib.check(lib.insmos_bev_distance_map(coords.data_ptr(), len(coords), B, H, W, 6, dist.data_ptr(), ws.data_ptr(), ws.numel(), st), "dist")
far = torch.full_like(dist, 7)
near = torch.zeros_like(dist)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for cin, layer_idx in ((256, 0), (128, 1), (128, 3), (128, 5)):
    w = (rng.normal(size=(128, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    layer = ConvLayer(lib, P.conv2d_weight_to_taps(w), np.zeros(128, np.float32), cin, 128, dev)
    x = torch.randn((B * H * W, cin), device=dev)
    o = torch.empty((B * H * W, 128), device=dev)
    cv = torch.zeros(128, device=dev)

    def run(d):
        if d is None:
            return lib.insmos_bev_conv3x3(x.data_ptr(), B, H, W, cin, cin, layer.w.data_ptr(), layer.b.data_ptr(), o.data_ptr(), 128, 128, 1, st)
        return lib.insmos_bev_conv3x3_skip(x.data_ptr(), B, H, W, cin, cin, layer.w.data_ptr(), layer.b.data_ptr(), o.data_ptr(), 128, 128, 1,
                                           d.data_ptr(), layer_idx, cv.data_ptr(), st)

    res = []
    for name, d in (("dense", None), ("skip:real", dist), ("skip:all-constant", far), ("skip:none-constant", near), ("dense", None), ("skip:real", dist)):
        for _ in range(3):
            assert run(d) == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run(d)
        torch.cuda.synchronize()
        res.append("%s %.1f us" % (name, (time.perf_counter() - t0) * 1e5))
    lib.insmos_bev_skip_executed_pairs(dist.data_ptr(), B, H, W, layer_idx, cnt.data_ptr(), st)
    torch.cuda.synchronize()
    print("cin %d layer %d: executed pairs %.3f of dense | " % (cin, layer_idx, cnt.item() / (9.0 * B * H * W)) + " | ".join(res), flush=True)

#!/usr/bin/env python3
"""tools/bn_kernel_probe.py <rows> <channels> [reps=20] -- forward + backward of ONE segmented BatchNorm layer (4 segments) in a loop:
run under `rocprofv3 --kernel-trace --stats` for the four kernels' durations at that shape (tools/bn_kernel_probe.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd.autograd import BnPlan, batch_norm_train_seg
n, c = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
S = 4
x = torch.randn((n, c), device="cuda", requires_grad=True)
g = torch.ones(c, device="cuda", requires_grad=True)
b = torch.zeros(c, device="cuda", requires_grad=True)
dy = torch.randn((n, c), device="cuda")
q = n // S
plan = BnPlan([(i * q, (i + 1) * q if i < S - 1 else n, i) for i in range(S)], n, S, "cuda")
for _ in range(reps):
    y = batch_norm_train_seg(x, g, b, plan, relu=True, force_segmented=True)
    y.backward(dy)
torch.cuda.synchronize()
print("rows", n, "channels", c, "chunks", plan.table(c).n_chunks, "MB per pass %.1f" % (n * c * 4 / 1e6))

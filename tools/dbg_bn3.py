import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import autograd as A
rng = np.random.default_rng(0)
for n, c in ((75000, 256), (75000, 128), (40000, 256), (75000, 272), (150000, 64)):
    x = (rng.normal(size=(n, c)) * rng.uniform(0.5, 3, c) + rng.normal(size=c) * 2).astype(np.float32)
    gm, bt = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(size=c).astype(np.float32)
    gy = rng.normal(size=(n, c)).astype(np.float32)
    xr = torch.from_numpy(x).double().requires_grad_(True)
    gr, br = torch.from_numpy(gm).double().requires_grad_(True), torch.from_numpy(bt).double().requires_grad_(True)
    yr = torch.relu(F.batch_norm(xr, None, None, gr, br, training=True, eps=1e-3))
    (yr * torch.from_numpy(gy).double()).sum().backward()
    for new in (True, False):
        xt = torch.from_numpy(x).cuda().requires_grad_(True)
        g_, b_ = torch.from_numpy(gm).cuda().requires_grad_(True), torch.from_numpy(bt).cuda().requires_grad_(True)
        if new:
            y = A.batch_norm_train_seg(xt, g_, b_, A.BnPlan.whole(n, "cuda"), None, None, eps=1e-3, relu=True)
        else:
            y = A.batch_norm_train(xt, g_, b_, None, None, eps=1e-3, relu=True)
        (y * torch.from_numpy(gy).cuda()).sum().backward()
        rel = lambda a, b: float((a.cpu().double() - b).abs().max()) / (float(b.abs().max()) + 1e-12)
        d = (xt.grad.cpu().double() - xr.grad).abs()
        bad_rows = torch.nonzero(d.max(1).values > 1e-3 * float(xr.grad.abs().max())).flatten()
        print(n, c, "new" if new else "old", "vs float64: y %.2e dx %.2e dgamma %.2e dbeta %.2e" %
              (rel(y.detach(), yr.detach()), rel(xt.grad, xr.grad), rel(g_.grad, gr.grad), rel(b_.grad, br.grad)),
              "bad dx rows:", len(bad_rows), bad_rows[:4].tolist(), bad_rows[-2:].tolist())

import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import autograd as A
rng = np.random.default_rng(0)
for n, c in ((4598, 16), (5942, 32), (3547, 64), (2380, 128), (1732, 128), (18750, 128), (9000, 24), (75000, 256)):
    x = (rng.normal(size=(n, c)) * rng.uniform(0.5, 3, c) + rng.normal(size=c) * 2).astype(np.float32)
    gm, bt = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(size=c).astype(np.float32)
    gy = rng.normal(size=(n, c)).astype(np.float32)
    res = []
    for new in (True, False):
        xt = torch.from_numpy(x).cuda().requires_grad_(True)
        g_, b_ = torch.from_numpy(gm).cuda().requires_grad_(True), torch.from_numpy(bt).cuda().requires_grad_(True)
        if new:
            y = A.batch_norm_train_seg(xt, g_, b_, A.BnPlan.whole(n, "cuda"), None, None, eps=1e-3, relu=True)
        else:
            y = A.batch_norm_train(xt, g_, b_, None, None, eps=1e-3, relu=True)
        (y * torch.from_numpy(gy).cuda()).sum().backward()
        res.append((y.detach(), xt.grad, g_.grad, b_.grad))
    rel = lambda a, b: float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
    print(n, c, "y %.2e dx %.2e dgamma %.2e dbeta %.2e" % tuple(rel(a, b) for a, b in zip(res[0], res[1])),
          "dgamma new/old norm %.5f" % (float(res[0][2].norm()) / float(res[1][2].norm())))

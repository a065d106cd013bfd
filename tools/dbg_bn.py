import os, sys, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import params as P, autograd as A
from insmos_amd.synth import make_window, make_labels
from insmos_amd.train_unet import InsMOSTrainer
import insmos_amd.train_unet as TU, insmos_amd.train_motionnet as TM

def gtb(rng, m=6):
    gt = np.zeros((1, m, 8), np.float32)
    gt[0, :, 0] = rng.uniform(-30, 30, m); gt[0, :, 1] = rng.uniform(-20, 20, m); gt[0, :, 2] = rng.uniform(-1.5, -0.5, m)
    gt[0, :, 3] = rng.uniform(1.5, 4.5, m); gt[0, :, 4] = rng.uniform(0.6, 2.0, m); gt[0, :, 5] = rng.uniform(1.2, 1.8, m)
    gt[0, :, 6] = rng.uniform(-3.1, 3.1, m); gt[0, :, 7] = rng.integers(1, 4, m)
    return gt
cfg = copy.deepcopy(P.default_cfg()); cfg["MODEL"]["USE_MOTION_LOSS"] = True
sd = P.random_state_dict(cfg, 2, cls_bias=-1.0, box_w_std=0.05)
rng = np.random.default_rng(21)
batch = []
for s, (ns, az) in zip((3, 4, 5), ((3, 96), (4, 80), (3, 128))):
    w = make_window(seed=s, n_scans=ns, n_az=az)
    batch.append({"past_point_clouds": torch.from_numpy(w).cuda(), "past_labels": [None, torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=s)).cuda()],
                  "gt_boxes": torch.from_numpy(gtb(rng)).cuda()})

def old_bn(x, g, b, plan, rm=None, rv=None, momentum=0.1, eps=1e-5, relu=False):
    assert plan.S == 1
    return A.batch_norm_train(x, g, b, rm, rv, momentum, eps, relu)

def run(items, patch_old):
    seg_new_u, seg_new_m = TU.batch_norm_train_seg, TM.batch_norm_train_seg
    if patch_old:
        TU.batch_norm_train_seg = old_bn; TM.batch_norm_train_seg = old_bn
    try:
        tr = InsMOSTrainer(cfg, sd)
        loss, tb, _, preds = tr.forward(items, "train")
        loss.backward()
        return float(loss.detach()), tb, {k: v.grad.clone() for k, v in tr.params.items() if v.grad is not None}
    finally:
        TU.batch_norm_train_seg, TM.batch_norm_train_seg = seg_new_u, seg_new_m

# (1) one window: new fused BN vs the old kernels
l_new, tb_new, g_new = run(batch[:1], False)
l_old, tb_old, g_old = run(batch[:1], True)
print("B=1 loss new %.6f old %.6f" % (l_new, l_old), tb_new[0], tb_old[0])
worst = sorted(((float((g_new[k] - g_old[k]).abs().max()) / (float(g_old[k].abs().max()) + 1e-12), k) for k in g_new), reverse=True)[:8]
print("worst relative grad differences new vs old BN:", worst)
# (2) batched vs sequential per-item numbers
lb, tbb, gb = run(batch, False)
seq = [run([it], False) for it in batch]
print("batched loss %.6f, sequential mean %.6f" % (lb, sum(s[0] for s in seq) / 3))
for i in range(3):
    print(i, "batched", tbb[i]); print(i, "single ", seq[i][1][0])

import os, sys, zlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd import params as P, autograd as A
from insmos_amd.synth import make_window
from insmos_amd.train_unet import InsMOSTrainer
import insmos_amd.train_unet as TU, insmos_amd.train_motionnet as TM
g = np.load("tests/golden/train_wiring.npz")
cfg = P.default_cfg()
window = make_window(seed=21, n_scans=3, n_az=96)
sd = P.random_state_dict(cfg, 9, cls_bias=-1.0, box_w_std=0.05)
def old_bn(x, gm, b, plan, rm=None, rv=None, momentum=0.1, eps=1e-5, relu=False):
    return A.batch_norm_train(x, gm, b, rm, rv, momentum, eps, relu)
for patch in (False, True):
    if patch:
        TU.batch_norm_train_seg = old_bn; TM.batch_norm_train_seg = old_bn
    tr = InsMOSTrainer(cfg, sd)
    batch = [{"past_point_clouds": torch.from_numpy(window).cuda(), "past_labels": [torch.from_numpy(g["gt_labels"]).cuda()],
              "gt_boxes": torch.from_numpy(g["gt_boxes"]).cuda()}]
    loss, tb, _, _ = tr.forward(batch, "train")
    loss.backward()
    grads = {str(n): (float(g["grad_norms"][i]), g["grad_samples"][i]) for i, n in enumerate(g["grad_names"])}
    rows = []
    for stem, v in tr.unet.params.items():
        name = P.UNET_PREFIX + stem
        gr = tr.unet.to_reference_layout(stem, v.grad).astype(np.float64).reshape(-1)
        nrm = np.sqrt((gr * gr).sum())
        rows.append((abs(nrm - grads[name][0]) / grads[name][0], name, nrm, grads[name][0]))
    rows.sort(reverse=True)
    print("old BN" if patch else "new BN", "losses", tb[0], "golden", g["losses"])
    for r in rows[:6]:
        print("   %.4f %s ours %.6g golden %.6g" % r)
    print("   n rows of levels:", tr.unet._last_tables["win_rows"] if hasattr(tr.unet, "_last_tables") else None)

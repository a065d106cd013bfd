"""CPU: the training graphs the GPU tests use as their float64 yardstick, checked against the reference's OWN training code.

tests/golden/train_wiring.npz (make_golden.py:train_wiring_golden) holds what the reference's model classes produce for one
training pass -- MotionNet / VoxelGenerate / MeanVFE / UNetV2(..., 'train') / CenterHead.get_loss / MOSLoss imported from the
reference as written, torch autograd through them, over the differentiable oracle-backed stand-ins of oracle/shims: the four
losses, the boxes the pass predicted, and the norm + 12 entries of EVERY parameter's gradient.  Here the restated graphs --
`_ref_graph` of tests/test_train_unet.py (the very function the HIP 3D-branch trainer is compared with on the GPU) and the
MotionNet graph of tests/test_train_slice.py, restated below -- are evaluated in float64 over the oracle's kernel maps and
must reproduce those numbers.  Together with the GPU tests (HIP == restated graph) this pins the HIP training step's wiring,
losses and gradients to the reference's code; MinkowskiEngine / spconv primitive semantics stay dep-knowledge.
"""
import os
import types
import zlib

import numpy as np
import torch
import torch.nn.functional as F

from insmos_amd import params as P
from insmos_amd.synth import make_window
from oracle import ref_model as M
from oracle import ref_ops as R


def _samples(name, numel, n=12):
    return np.random.default_rng(zlib.crc32(name.encode())).integers(0, numel, n)  # as make_golden.py:_grad_samples


def _check_grad(g, name, grad_ref, bad):
    """g: our gradient already in the reference's layout; grad_ref: (norm, 12 sampled entries)."""
    g = np.asarray(g, np.float64).reshape(-1)
    norm_ref, samp_ref = grad_ref
    scale = max(float(np.abs(g).max()), 1e-12)
    norm = float(np.sqrt((g * g).sum()))
    samp = g[_samples(name, g.size)]
    e_norm = abs(norm - norm_ref) / max(norm_ref, 1e-12)
    e_samp = float(np.abs(samp - samp_ref).max()) / scale
    if e_norm > 5e-3 or e_samp > 5e-3:
        bad.append((name, e_norm, e_samp))
    return max(e_norm, e_samp)


def _golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_wiring.npz"))
    grads = {str(n): (float(g["grad_norms"][i]), g["grad_samples"][i]) for i, n in enumerate(g["grad_names"])}
    cfg = P.default_cfg()
    window = make_window(seed=21, n_scans=3, n_az=96)
    sd = P.random_state_dict(cfg, 9, cls_bias=-1.0, box_w_std=0.05)
    return g, grads, cfg, window, sd


def test_motionnet_training_graph_vs_reference_code(golden_dir):
    g, grads, cfg, window, sd = _golden(golden_dir)
    _, dbg = M.motionnet_forward(sd, window, want_debug=True)   # the oracle's kernel maps (and an eval forward, unused)
    tabs = dict(n125=dbg["nbr125"], n81=dbg["nbr81"], dn=dbg["nbr_dn"], up=dbg["nbr_up"])
    inverse = dbg["inverse"]
    MP = P.ME_PREFIX
    pr, ref_name = {}, {}

    def add(ours, theirs, taps=False):
        v = np.asarray(sd[MP + theirs], np.float64)
        if taps:
            v = v[None] if v.ndim == 2 else v
        pr[ours] = torch.from_numpy(v.copy()).requires_grad_(True)
        ref_name[ours] = MP + theirs

    for cname, bname, kv, ci, co in P.ME_CONVS:
        add(cname + ".kernel", cname + ".kernel", True)
        add(bname + ".weight", bname + ".bn.weight")
        add(bname + ".bias", bname + ".bn.bias")
    for name, ci, co in P.ME_BLOCKS:
        for c, n in ((".conv1", ".norm1"), (".conv2", ".norm2")):
            add(name + c + ".kernel", name + c + ".kernel", True)
            add(name + n + ".weight", name + n + ".bn.weight")
            add(name + n + ".bias", name + n + ".bn.bias")
        if ci != co:
            add(name + ".downsample.0.kernel", name + ".downsample.0.kernel", True)
            add(name + ".downsample.1.weight", name + ".downsample.1.bn.weight")
            add(name + ".downsample.1.bias", name + ".downsample.1.bn.bias")
    add("final.kernel", "final.kernel", True)
    add("final.bias", "final.bias")

    def conv(x, wname, nbr, bias=None):
        ww = pr[wname]
        if nbr is None:
            y = x @ ww[0]
        else:
            y = torch.zeros((nbr.shape[1], ww.shape[2]), dtype=torch.float64)
            for k in range(nbr.shape[0]):
                o = np.nonzero(nbr[k] >= 0)[0]
                if len(o):
                    y = y.index_add(0, torch.from_numpy(o), x[torch.from_numpy(nbr[k][o].astype(np.int64))] @ ww[k])
        return y if bias is None else y + pr[bias].reshape(1, -1)

    def bn(x, name, relu):
        y = F.batch_norm(x, None, None, pr[name + ".weight"], pr[name + ".bias"], training=True, eps=1e-5)
        return torch.relu(y) if relu else y

    def block(name, x, nbr):
        out = bn(conv(x, name + ".conv1.kernel", nbr), name + ".norm1", True)
        out = bn(conv(out, name + ".conv2.kernel", nbr), name + ".norm2", False)
        res = x
        if (name + ".downsample.0.kernel") in pr:
            res = bn(conv(x, name + ".downsample.0.kernel", None), name + ".downsample.1", False)
        return torch.relu(out + res)

    n0 = tabs["n81"][0].shape[1]
    out_p1 = bn(conv(torch.full((n0, 1), 0.5, dtype=torch.float64), "conv0p1s1.kernel", tabs["n125"]), "bn0", True)
    out = bn(conv(out_p1, "conv1p1s2.kernel", tabs["dn"][0]), "bn1", True)
    b1 = block("block1.0", out, tabs["n81"][1])
    out = bn(conv(b1, "conv2p2s2.kernel", tabs["dn"][1]), "bn2", True)
    b2 = block("block2.0", out, tabs["n81"][2])
    out = bn(conv(b2, "conv3p4s2.kernel", tabs["dn"][2]), "bn3", True)
    out = block("block3.0", out, tabs["n81"][3])
    out = bn(conv(out, "convtr5p8s2.kernel", tabs["up"][2]), "bntr5", True)
    out = block("block6.0", torch.cat([out, b2], 1), tabs["n81"][2])
    out = bn(conv(out, "convtr6p4s2.kernel", tabs["up"][1]), "bntr6", True)
    out = block("block7.0", torch.cat([out, b1], 1), tabs["n81"][1])
    out = bn(conv(out, "convtr7p2s2.kernel", tabs["up"][0]), "bntr7", True)
    out = block("block8.0", torch.cat([out, out_p1], 1), tabs["n81"][0])
    motion = conv(out, "final.kernel", None, "final.bias")
    cur = np.nonzero((window[:, 4] / np.float32(0.1)) == 0)[0]
    feat_cur = motion[torch.from_numpy(inverse[cur].astype(np.int64))]
    # the train-mode MotionNet output is what the reference hands to the 3D branch (motionnet.py:48)
    np.testing.assert_allclose(feat_cur.detach().numpy(), g["current_point"][:, 4:7], atol=2e-4)
    np.testing.assert_array_equal(window[cur, :4], g["current_point"][:, :4])
    z = feat_cur.clone()
    z[:, 0] = -float("inf")
    loss = F.nll_loss(torch.log(torch.softmax(z, 1).clamp(min=1e-8)), torch.from_numpy(g["gt_labels"]).long(),
                      weight=torch.tensor([0.0, 0.5, 0.5], dtype=torch.float64))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["losses"][3])) < 2e-5 * max(1.0, float(g["losses"][3]))
    bad, worst = [], 0.0
    for ours, theirs in ref_name.items():
        worst = max(worst, _check_grad(pr[ours].grad.numpy(), theirs, grads[theirs], bad))
    print("MotionNet: worst relative gradient deviation from the reference's autograd %.2e over %d tensors" % (worst, len(ref_name)))
    assert not bad, sorted(bad, key=lambda t: -max(t[1], t[2]))[:6]
    assert len(ref_name) == sum(1 for n in grads if n.startswith(MP))


def test_unet_training_graph_vs_reference_code(golden_dir):
    from insmos_amd.train_unet import UNetV2Trainer
    from test_train_unet import _ref_graph
    g, grads, cfg, window, sd = _golden(golden_dir)
    cur = g["current_point"]
    vs, rng = cfg["DATA"]["VOXEL_SIZE"], cfg["DATA"]["POINT_CLOUD_RANGE"]
    _, _, dbg = M.unet_forward(sd, cfg, cur, want_debug=True)      # the oracle's voxelisation and kernel maps
    assert len(dbg["voxel_features"]) == int(g["n_voxels"])
    NS = types.SimpleNamespace
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    coords = {l: tt(np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)) for l, c in dbg["coords"].items()}
    feat8 = np.zeros((len(dbg["voxel_features"]), 8), np.float32)
    feat8[:, :7] = dbg["voxel_features"]
    T = dict(subm={l: NS(nbr=tt(t)) for l, t in dbg["subm"].items()}, down={l: NS(nbr=tt(t)) for l, t in dbg["down"].items()},
             inv={l: NS(nbr=tt(t)) for l, t in dbg["inv"].items()}, down5=NS(nbr=tt(dbg["down5"])), inv5=NS(nbr=tt(dbg["inv5"])),
             coords=coords, feat=tt(feat8), pcid=tt(dbg["pc_voxel_id"].astype(np.int64)))
    # dense 3x3 (pad 1) table of the BEV map: tap ky*3+kx reads (y+ky-1, x+kx-1)   (include/insmos_hip.h: insmos_dense_nbr2d)
    grid = np.round((np.array(rng[3:6], float) - np.array(rng[0:3], float)) / np.array(vs)).astype(np.int64)
    H, W, D = int(grid[1]) // 8, int(grid[0]) // 8, 2
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    nbr_bev = np.full((9, H * W), -1, np.int32)
    for ky in range(3):
        for kx in range(3):
            y2, x2 = yy + ky - 1, xx + kx - 1
            ok = (y2 >= 0) & (y2 < H) & (x2 >= 0) & (x2 < W)
            nbr_bev[ky * 3 + kx][ok.reshape(-1)] = (y2 * W + x2)[ok]
    eng = NS(nbr_bev=tt(nbr_bev), bevD=D, bevH=H, bevW=W, n_bev_layers=int(cfg["MODEL"]["BACKBONE_2D"]["LAYER_NUMS"][0]))
    # instance one-hots of the boxes the reference's pass predicted; box scaling as spconv_unet.py:324-329 on the CPU
    b = g["pred_boxes"].astype(np.float32).copy()
    for d in range(3):
        b[:, d] = ((b[:, d] - np.float32(rng[d])) / np.float32(vs[d])) / np.float32(8)
        b[:, 3 + d] = (b[:, 3 + d] / np.float32(vs[d])) / np.float32(8)
    boxes8 = np.concatenate([b, g["pred_labels"].astype(np.float32).reshape(-1, 1)], 1).astype(np.float32)
    oh = {}
    for lvl in (4, 3, 2, 1):
        oh[lvl] = tt(R.boxes_to_onehot(dbg["coords"][lvl][:, [2, 1, 0]], boxes8, 3, True).astype(np.float32))
        boxes8[:, 0:6] *= np.float32(2)
    assert sum(int(v.sum()) for v in oh.values()) > 0
    tc = cfg["MODEL"]["DENSE_HEAD"]["TARGET_ASSIGNER_CONFIG"]
    heat, anno, ind, mask = R.center_assign_targets(g["gt_boxes"][0], grid, np.array(rng), tc["VOXEL_SIZE"], tc["OUT_SIZE_FACTOR"], 3,
                                                    tc["MAX_OBJS"], tc["GAUSSIAN_OVERLAP"], tc["MIN_RADIUS"])
    assert int(mask.sum()) == 7
    targets = {"heatmaps": [tt(heat)[None]], "anno_boxes": [tt(anno)[None]], "inds": [tt(ind)[None]], "masks": [tt(mask)[None]]}
    tr = UNetV2Trainer(cfg, sd, device="cpu", engine=NS(bevH=H, bevW=W))
    total, l_cls, l_loc, l_mos, pr = _ref_graph(tr, T, oh, targets, len(cur), tt(g["gt_labels"]), eng)
    ref_cls, ref_loc, ref_mos = (float(v) for v in g["losses"][:3])
    print("losses ours/reference: cls %.5f/%.5f loc %.5f/%.5f mos %.5f/%.5f" % (l_cls, ref_cls, l_loc, ref_loc, l_mos, ref_mos))
    assert abs(l_cls - ref_cls) < 1e-4 * ref_cls and abs(l_loc - ref_loc) < 1e-4 * ref_loc and abs(l_mos - ref_mos) < 1e-4 * ref_mos
    bad, worst, n = [], 0.0, 0
    for stem, v in pr.items():
        name = P.UNET_PREFIX + stem
        worst = max(worst, _check_grad(tr.to_reference_layout(stem, v.grad), name, grads[name], bad))
        n += 1
    print("UNetV2: worst relative gradient deviation from the reference's autograd %.2e over %d tensors" % (worst, n))
    assert not bad, sorted(bad, key=lambda t: -max(t[1], t[2]))[:6]
    assert n == sum(1 for k in grads if k.startswith(P.UNET_PREFIX))

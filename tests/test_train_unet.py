"""-m gpu: the 3D branch in training mode (insmos_amd/train_unet.py) and the full training loss.

  * wiring: with BatchNorm on running statistics the training graph must reproduce the inference path's logits and boxes
    (the inference path is the oracle-checked one) -- this pins layer order, concatenations, the pair-sum reduction, the
    BEV scatter, the deconv-as-4-taps table and the voxel -> point gather; that test lives in test_gpu_model.py
    (test_training_graph_on_running_stats_reproduces_inference), next to the oracle-calibrated checkpoint it needs;
  * numerics: in train mode (batch statistics) the loss and the gradient of EVERY parameter against the same graph
    restated with torch index ops in float64 on the CPU (independent code, same kernel maps / targets / one-hots);
  * InsMOSTrainer: the reference's 'train' return signature, gradients reach both branches, SGD lowers the loss,
    export_state_dict round-trips into the inference model.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gt_boxes(rng, m=6):
    gt = np.zeros((1, m, 8), np.float32)
    gt[0, :, 0] = rng.uniform(-30, 30, m)
    gt[0, :, 1] = rng.uniform(-20, 20, m)
    gt[0, :, 2] = rng.uniform(-1.5, -0.5, m)
    gt[0, :, 3] = rng.uniform(1.5, 4.5, m)
    gt[0, :, 4] = rng.uniform(0.6, 2.0, m)
    gt[0, :, 5] = rng.uniform(1.2, 1.8, m)
    gt[0, :, 6] = rng.uniform(-3.1, 3.1, m)
    gt[0, :, 7] = rng.integers(1, 4, m)
    return gt


def _ref_graph(tr, T, oh, targets, cur_n, gt_labels, eng):
    """The training graph of UNetV2Trainer.forward + losses in float64 on the CPU, written independently with index ops."""
    import torch.nn.functional as F
    pr = {k: v.detach().cpu().double().requires_grad_(True) for k, v in tr.params.items()}
    tab = lambda t: t.nbr.cpu().numpy()
    subm, down, inv = {l: tab(t) for l, t in T["subm"].items()}, {l: tab(t) for l, t in T["down"].items()}, \
        {l: tab(t) for l, t in T["inv"].items()}
    down5, inv5 = tab(T["down5"]), tab(T["inv5"])
    nbr_bev = eng.nbr_bev.cpu().numpy()

    def conv(x, w, nbr, bias=None):
        if nbr is None:
            y = x @ w[0]
        else:
            y = torch.zeros((nbr.shape[1], w.shape[2]), dtype=torch.float64)
            for k in range(nbr.shape[0]):
                o = np.nonzero(nbr[k] >= 0)[0]
                if len(o):
                    y = y.index_add(0, torch.from_numpy(o), x[torch.from_numpy(nbr[k][o].astype(np.int64))] @ w[k])
        return y if bias is None else y + bias

    def bn(x, stem, relu):
        y = F.batch_norm(x, None, None, pr[stem + ".weight"], pr[stem + ".bias"], training=True, eps=1e-3)
        return torch.relu(y) if relu else y

    def cbr(c, b, x, nbr):
        return bn(conv(x, pr[c + ".weight"], nbr), b, True)

    def cbr_cat(c, b, x, o, nbr):
        return bn(conv(torch.cat([x, o], 1), pr[c + ".weight"], nbr), b, True)

    def basic(stem, x, nbr):
        out = bn(conv(x, pr[stem + ".conv1.weight"], nbr), stem + ".bn1", True)
        out = bn(conv(out, pr[stem + ".conv2.weight"], nbr), stem + ".bn2", False)
        return torch.relu(out + x)

    def ur(lvl, lat, bottom, nbr):
        trans = basic(f"conv_up_t{lvl}", lat, nbr)
        cat = torch.cat([bottom, trans], 1)
        m = cbr(f"conv_up_m{lvl}.0", f"conv_up_m{lvl}.1", cat, nbr)
        return m + cat.reshape(cat.shape[0], m.shape[1], 2).sum(2)

    feat = T["feat"][:, :tr.in_ch].cpu().double()
    x0 = cbr("conv_input.0", "conv_input.1", feat, subm[1])
    xc = {1: cbr("conv1.0.0", "conv1.0.1", x0, subm[1])}
    for l in (2, 3, 4):
        a = cbr(f"conv{l}.0.0", f"conv{l}.0.1", xc[l - 1], down[l])
        b = cbr(f"conv{l}.1.0", f"conv{l}.1.1", a, subm[l])
        xc[l] = cbr(f"conv{l}.2.0", f"conv{l}.2.1", b, subm[l])
    enc = cbr("conv_out.0", "conv_out.1", xc[4], down5)
    D, H, W = eng.bevD, eng.bevH, eng.bevW
    c5 = T["coords"][5].cpu().long()
    # HeightCompression: dense (C, D, H, W) viewed as (C * D, H, W) -> channel c * D + d at site (y, x)
    C = enc.shape[1]
    site = (c5[:, 2] * W + c5[:, 3])[:, None].expand(-1, C)
    cols = torch.arange(C)[None, :] * D + c5[:, 1:2]
    flat = torch.zeros((H * W, C * D), dtype=torch.float64).index_put((site, cols), enc)
    f = cbr("bev_backbone.blocks.0.1", "bev_backbone.blocks.0.2", flat, nbr_bev)
    for k in range(eng.n_bev_layers):
        f = cbr(f"bev_backbone.blocks.0.{4 + 3 * k}", f"bev_backbone.blocks.0.{5 + 3 * k}", f, nbr_bev)
    # ConvTranspose2d(2, 2): out[2y+ky, 2x+kx] = in[y, x] @ W[ky*2+kx]
    wd = pr["bev_backbone.deblocks.0.0.weight"]
    up = torch.stack([f @ wd[k] for k in range(4)], 1).reshape(H, W, 2, 2, -1).permute(0, 2, 1, 3, 4).reshape(4 * H * W, -1)
    up = bn(up, "bev_backbone.deblocks.0.1", True)
    cls = up @ pr["center_head.conv_cls.weight"][0] + pr["center_head.conv_cls.bias"]
    box = up @ pr["center_head.conv_box.weight"][0] + pr["center_head.conv_box.bias"]
    o = {l: oh[l].cpu().double() for l in oh}
    x = conv(enc, pr["inv_conv_out.weight"], inv5)
    x = cbr_cat("conv_up_instance_block.0", "conv_up_instance_block.1", x, o[4], subm[4])
    m = ur(4, x, x, subm[4])
    x = cbr("inv_conv4.0", "inv_conv4.1", m, inv[4])
    x = cbr_cat("conv_up_instance_block_up4.0", "conv_up_instance_block_up4.1", x, o[3], subm[3])
    m = ur(3, xc[3], x, subm[3])
    x = cbr("inv_conv3.0", "inv_conv3.1", m, inv[3])
    x = cbr_cat("conv_up_instance_block_up3.0", "conv_up_instance_block_up3.1", x, o[2], subm[2])
    m = ur(2, xc[2], x, subm[2])
    x = cbr("inv_conv2.0", "inv_conv2.1", m, inv[2])
    x = cbr_cat("conv_up_instance_block_up2.0", "conv_up_instance_block_up2.1", x, o[1], subm[1])
    m = ur(1, xc[1], x, subm[1])
    x = cbr("conv_up_out.0.0", "conv_up_out.0.1", m, subm[1])
    seg = cbr_cat("conv_up_instance_block_up1.0", "conv_up_instance_block_up1.1", x, o[1], subm[1])
    vox = seg @ pr["mos_seg_layer.weight"][0] + pr["mos_seg_layer.bias"]
    pcid = T["pcid"][:cur_n].cpu()
    z = (vox[pcid.clamp(min=0)] * (pcid >= 0)[:, None].double()).clone()
    z[:, 0] = -float("inf")
    loss_mos = F.nll_loss(torch.log(torch.softmax(z, 1).clamp(min=1e-8)), gt_labels.cpu().long(),
                          weight=torch.tensor([0.0, 0.5, 0.5], dtype=torch.float64))
    # CenterHead.get_loss (center_head.py:279-331) in float64
    heat = targets["heatmaps"][0][0].cpu().double().permute(1, 2, 0).reshape(-1, 3)
    p = torch.sigmoid(cls).clamp(1e-4, 1 - 1e-4)
    pos = (heat == 1).double()
    focal = (-(p + 1e-12).log() * (1 - p) ** 2 * pos - (1 - p + 1e-12).log() * p ** 2 * (1 - heat) ** 4).sum()
    loss_cls = focal / max(float(pos.sum()), 1.0)
    anno, ind, mask = targets["anno_boxes"][0][0].cpu().double(), targets["inds"][0][0].cpu(), targets["masks"][0][0].cpu().double()
    loss_loc = 2.0 * ((box[ind] - anno).abs() * mask[:, None]).sum() / (mask.sum() + 1e-4)
    total = loss_cls + loss_loc + loss_mos
    total.backward()
    return float(total.detach()), float(loss_cls.detach()), float(loss_loc.detach()), float(loss_mos.detach()), pr


def test_unet_training_step_vs_float64_restatement():
    from insmos_amd import params as P
    from insmos_amd.engine import Engine
    from insmos_amd.synth import make_labels, make_window
    from insmos_amd.train_unet import UNetV2Trainer
    rng = np.random.default_rng(4)
    cfg = P.default_cfg()
    sd = P.random_state_dict(cfg, 6, cls_bias=-1.0, box_w_std=0.05)
    w = make_window(seed=8, n_scans=3, n_az=96)
    eng = Engine(cfg, sd, "cuda:0")
    eng.keep_current_points = True
    eng.forward_window(torch.from_numpy(w).cuda(), native=False)
    cur = eng.last_current_points.clone()
    gt_labels = torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=8)).cuda()
    gt_boxes = torch.from_numpy(_gt_boxes(rng)).cuda()
    tr = UNetV2Trainer(cfg, sd, engine=eng)
    loss, tb, out = tr.loss(cur, gt_boxes, gt_labels)
    loss.backward()
    T = eng._un_tables
    # the one-hots the HIP pass used (constants of the graph): recomputed from its predicted boxes
    nv = {l: int(T["coords"][l].shape[0]) for l in (1, 2, 3, 4)}
    pd = out["pred_dicts"][0]
    K = len(pd["pred_boxes"])
    pb = torch.zeros((eng.post_max, 7), device="cuda")
    pl = torch.zeros((eng.post_max,), dtype=torch.int64, device="cuda")
    pb[:K], pl[:K] = pd["pred_boxes"], pd["pred_labels"]
    cnt = torch.tensor([K, 0, 0, 0], dtype=torch.int32, device="cuda")
    scratch = torch.empty((int(eng.lib.insmos_boxes_to_onehot_scratch_ints(eng.post_max, max(nv.values()))),),
                          dtype=torch.int32, device="cuda")
    oh = {}
    for lvl, mult in ((4, 1.0), (3, 2.0), (2, 4.0), (1, 8.0)):
        o = torch.zeros((nv[lvl], 16), device="cuda")
        eng.instance_onehot(pb, pl, cnt, T["coords"][lvl], nv[lvl], mult, o, 16, 0, scratch)
        oh[lvl] = o[:, :3]
    total_r, cls_r, loc_r, mos_r, pr = _ref_graph(tr, T, oh, out["targets"], cur.shape[0], gt_labels, eng)
    print("boxes used for the instance one-hots:", K, "one-hot voxels:", {l: int(oh[l].sum()) for l in oh})
    print("loss hip %.6f ref %.6f | cls %.6f/%.6f loc %.6f/%.6f mos %.6f/%.6f" %
          (float(loss.detach()), total_r, tb["rpn_loss_cls"], cls_r, tb["rpn_loss_loc"], loc_r, tb["loss_mos"], mos_r))
    bad, worst = [], 0.0
    for k, v in tr.params.items():
        r = pr[k].grad
        if r is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
            continue
        assert v.grad is not None, k
        g, r = v.grad.cpu().numpy().astype(np.float64), r.numpy()
        scale = max(np.abs(r).max(), 1e-6)
        err = float(np.abs(g - r).max() / scale)
        worst = max(worst, err)
        if err > 5e-3:
            bad.append((k, err, float(scale)))
    print("worst relative gradient error %.2e over %d parameters" % (worst, len(tr.params)))
    assert abs(tb["rpn_loss_cls"] - cls_r) < 1e-4 * max(1.0, abs(cls_r))
    assert abs(tb["rpn_loss_loc"] - loc_r) < 1e-4 * max(1.0, abs(loc_r))
    assert abs(tb["loss_mos"] - mos_r) < 1e-4 * max(1.0, abs(mos_r))
    assert not bad, sorted(bad, key=lambda t: -t[1])[:8]


def test_insmos_trainer_train_mode_signature_and_descent():
    from insmos_amd import params as P
    from insmos_amd.models import InsMOSNet
    from insmos_amd.synth import make_labels, make_window
    from insmos_amd.train_unet import InsMOSTrainer
    rng = np.random.default_rng(11)
    cfg = P.default_cfg()
    sd = P.random_state_dict(cfg, 2, cls_bias=-1.0, box_w_std=0.05)
    batch = []
    for s in (3, 4):
        w = make_window(seed=s, n_scans=3, n_az=96)
        batch.append({"past_point_clouds": torch.from_numpy(w).cuda(),
                      "past_labels": [None, torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=s)).cuda()],
                      "gt_boxes": torch.from_numpy(_gt_boxes(rng)).cuda()})
    tr = InsMOSTrainer(cfg, sd)
    losses = []
    for step in range(4):
        loss, tb_list, gt_list, pred_list = tr.forward(batch, "train")
        assert tuple(loss.shape) == (1,) and len(tb_list) == len(gt_list) == len(pred_list) == 2
        assert set(tb_list[0]) == {"loss_mos", "loss_motion_encoder", "rpn_loss_cls", "rpn_loss_loc", "rpn_loss"}
        assert pred_list[0].shape == (gt_list[0].shape[0], 3)
        loss.backward()
        if step == 0:
            missing = [k for k, v in tr.params.items() if v.grad is None]
            assert not missing, missing[:5]
            assert all(bool(torch.isfinite(v.grad).all()) for v in tr.params.values())
        losses.append(float(loss.detach()))
        gnorm = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in tr.params.values())))
        tr.sgd_step(0.2 / max(gnorm, 1e-12))  # a normalised step of length 0.2 in parameter space
    print("training losses", losses)
    assert losses[-1] < losses[0]
    # the drop-in module serves the same mode (models/models.py:313-345): InsMOS_Model.forward(list, 'train') == the trainer's
    model = InsMOSNet(cfg, state_dict=sd).cuda().eval()
    loss_m, tb_m, gt_m, pred_m = model.forward(batch, "train")
    ref = InsMOSTrainer(cfg, sd)
    loss_r, tb_r, _, pred_r = ref.forward(batch, "train")
    assert torch.equal(loss_m, loss_r) and tb_m == tb_r and all(torch.equal(a, b) for a, b in zip(pred_m, pred_r))
    loss_m.backward()
    assert all(v.grad is not None for v in model.model.trainer.params.values())
    # the trained 3D branch goes back into a checkpoint the inference model loads
    sd2 = dict(sd)
    sd2.update(tr.unet.export_state_dict())
    model = InsMOSNet(cfg, state_dict=sd2).cuda().eval()
    with torch.no_grad():
        _, _, logits = model.forward([{"past_point_clouds": batch[0]["past_point_clouds"]}], "test")
    assert bool(torch.isfinite(logits[0]).all())


@pytest.mark.parametrize("chunk", ["1024", None])
def test_batched_training_step_equals_the_item_by_item_walk(chunk, monkeypatch):
    """(`chunk` = INSMOS_BN_CHUNK: with ONE pinned chunk length the predictions of the two walks agree row for row to 1e-4 -- the
    tight bound, which a regression in the segmented BatchNorm kernels or the chunk tables cannot pass; the adaptive lengths of
    BnPlan.chunk_rows move one last-bit ReLU decision in window 0, see below, and get the documented loose bound.)
    cfg-5's batch dimension: B windows per training step in ONE set of launches per branch (window index folded into the 4D time
    coordinate / spconv's batch column, per-window BatchNorm statistics, per-window losses) against the reference's walk over the
    batch list (models/models.py:313-345; here the same trainer with one window per call): the same per-item losses, the same
    predictions, the same gradient of the mean loss for every parameter (the sums run in another order: tolerance, not bits) and
    the same running statistics afterwards."""
    import copy
    from insmos_amd import params as P
    from insmos_amd.synth import make_labels, make_window
    from insmos_amd.train_unet import InsMOSTrainer
    if chunk is None:
        monkeypatch.delenv("INSMOS_BN_CHUNK", raising=False)
    else:
        monkeypatch.setenv("INSMOS_BN_CHUNK", chunk)
    rng = np.random.default_rng(21)
    cfg = copy.deepcopy(P.default_cfg())
    cfg["MODEL"]["USE_MOTION_LOSS"] = True
    sd = P.random_state_dict(cfg, 2, cls_bias=-1.0, box_w_std=0.05)
    batch = []
    for s, (ns, az) in zip((3, 4, 5), ((3, 96), (4, 80), (3, 128))):          # windows of different sizes
        w = make_window(seed=s, n_scans=ns, n_az=az)
        batch.append({"past_point_clouds": torch.from_numpy(w).cuda(),
                      "past_labels": [None, torch.from_numpy(make_labels(w[w[:, 4] == 0], seed=s)).cuda()],
                      "gt_boxes": torch.from_numpy(_gt_boxes(rng)).cuda()})
    tr_b, tr_s = InsMOSTrainer(cfg, sd), InsMOSTrainer(cfg, sd)
    loss_b, tb_b, gt_b, pred_b = tr_b.forward(batch, "train")
    loss_b.backward()
    # item by item: three one-window steps of the same model, losses averaged like models/models.py:365
    loss_s = torch.zeros(1, device="cuda")
    tb_s, pred_s = [], []
    for item in batch:
        l1, tb1, _, p1 = tr_s.forward([item], "train")
        loss_s = loss_s + l1
        tb_s.append(tb1[0])
        pred_s.append(p1[0])
    loss_s = loss_s / len(batch)
    loss_s.backward()
    assert abs(float(loss_b.detach()) - float(loss_s.detach())) < 1e-5 * max(1.0, abs(float(loss_s.detach())))
    for a, b in zip(tb_b, tb_s):
        assert set(a) == set(b) == {"loss_mos", "loss_motion_encoder", "rpn_loss_cls", "rpn_loss_loc", "rpn_loss"}
        for k in a:
            assert abs(a[k] - b[k]) < 1e-5 * max(1.0, abs(b[k])), (k, a[k], b[k])
    for a, b in zip(pred_b, pred_s):
        # (the two walks add BatchNorm partial sums in another order; every row agrees to ~1e-5 EXCEPT where a last-bit difference
        #  flips a discrete decision upstream -- a ReLU at a pre-activation within an ulp of zero, an instance column of a voxel on a
        #  box face -- whose effect the following convolutions spread over a neighbourhood: with the chunk lengths of
        #  BnPlan.chunk_rows one such flip shows in window 0 (11 of 6 066 rows up to 4e-3, profiles/r04_bn_chunk_flip_probe.txt;
        #  none with INSMOS_BN_CHUNK=1024, none in the 4D branch, none in windows 1 and 2 of the same launch).  So: the bulk tight,
        #  the flips few and small)
        assert a.shape == b.shape
        d = (a - b).detach().abs().max(1).values
        if chunk is not None:
            assert float(d.max()) < 1e-4, (float(d.max()), int((d > 1e-4).sum()), len(d))
        else:
            assert float(d.max()) < 2e-2 and int((d > 1e-4).sum()) <= max(1, len(d) // 200), (float(d.max()), int((d > 1e-4).sum()), len(d))
    bad = []
    for k, v in tr_b.params.items():
        g, h = v.grad, tr_s.params[k].grad
        assert (g is None) == (h is None), k
        if g is None:
            continue
        # (norm-wise: the two walks round differently in the last bits, which flips a handful of ReLU masks at cells whose
        #  pre-activation is within an ulp of zero -- single elements of a gradient may then differ by their whole value)
        err = float((g - h).norm())
        scale = float(h.norm())
        if err > 2e-2 * scale + 1e-6:
            bad.append((k, err, scale))
    assert not bad, sorted(bad, key=lambda t: -t[1] / (t[2] + 1e-12))[:6]
    for trn_b, trn_s in ((tr_b.motion, tr_s.motion), (tr_b.unet, tr_s.unet)):
        for k, v in trn_b.buffers.items():
            np.testing.assert_allclose(v.cpu().numpy(), trn_s.buffers[k].cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
    # the sequential switch walks the items with the same result
    import os
    os.environ["INSMOS_TRAIN_SEQUENTIAL"] = "1"
    try:
        tr_q = InsMOSTrainer(cfg, sd)
        loss_q, tb_q, _, _ = tr_q.forward(batch, "train")
    finally:
        os.environ.pop("INSMOS_TRAIN_SEQUENTIAL")
    assert abs(float(loss_q.detach()) - float(loss_s.detach())) < 1e-6 * max(1.0, abs(float(loss_s.detach()))) and len(tb_q) == 3

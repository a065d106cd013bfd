"""
oracle/ref_model.py -- CPU restatement of InsMOS_Model.forward(..., 'test') (models/models.py:297-376)
for ONE window, from a reference-layout state_dict.  TEST INFRASTRUCTURE ONLY (see ref_ops.py).

Follows, layer by layer:
  MotionNet.forward            models/backbones_3d/motionnet.py:21-50
  MinkUNetBase.forward         models/MinkowskiEngine/minkunet.py:139-181   (BasicBlock: ME
                               modules/resnet_block.py [dep-knowledge]: conv-bn-relu-conv-bn-(+res)-relu)
  VoxelGenerate / MeanVFE      models/backbones_3d/voxel_generate.py:17-31, backbones_2d/mean_vfe.py:36-55
  UNetV2.forward               models/backbones_3d/spconv_unet.py:267-416
  HeightCompression            models/backbones_2d/height_compression.py:14-33
  BaseBEVBackbone.forward      models/backbones_2d/base_bev_backbone.py:84-115   (torch CPU conv2d)
  CenterHead                   models/backbones_2d/center_head.py:65-98, 251-276
  post_processing              models/post_process.py:112-224
BatchNorm is applied as a separate eval-mode op after each conv, as the reference does (the HIP
path folds it into the taps; the difference is fp32 round-off and is covered by the tolerance).
PARITY UNPINNED for the ME/spconv PRIMITIVES (see ref_ops.py); pinned pieces are marked there.
The WIRING of this file (layer list, indice keys, concatenations, box scaling, instance features, gathers) and every
checkpoint parameter name / shape ARE pinned: tests/golden/wiring.npz holds the output of the reference's own model
classes, imported from the reference as written and run over stand-ins of the two missing libraries that are backed by
ref_ops.py (oracle/shims/README.md); test_restated_wiring_vs_reference_model_code compares this file against it.
"""
import numpy as np

from . import ref_ops as R

ME_P = "model.motion_encoder.MinkUNet."
UN_P = "model.unet."


def _relu(x):
    return np.maximum(x, np.float32(0))


def _bn(sd, stem, x, eps):
    return R.batchnorm_eval(x, sd[stem + ".weight"], sd[stem + ".bias"], sd[stem + ".running_mean"],
                            sd[stem + ".running_var"], eps)


def _me_taps(k):
    k = np.asarray(k, np.float32)
    return k[None] if k.ndim == 2 else k


def _sp_taps(w):
    w = np.asarray(w, np.float32)
    co, kz, ky, kx, ci = w.shape
    return np.ascontiguousarray(w.reshape(co, kz * ky * kx, ci).transpose(1, 2, 0))


# ------------------------------------------------------------------------------------------------
def motionnet_forward(sd, points, dt=0.1, ds=0.1, want_debug=False):
    """points (N,5) fp32 [x,y,z,intensity,t] -> current_point (Ncur,7) [x,y,z,r,m0,m1,m2]."""
    dbg = {}
    pts4 = np.concatenate([points[:, 0:3], points[:, 4:5]], axis=1).astype(np.float32)
    q = np.array([ds, ds, ds, dt], dtype=np.float32)
    coords, keys, inverse = R.me_quantize(pts4, q)
    feat = R.me_feature_average(np.full((len(points), 1), 0.5, np.float32), inverse, len(coords))
    lv = [(coords, keys)]
    parents = []
    for L in (1, 2, 3):
        pc, pk, par = R.me_stride_down(coords, keys, L)
        lv.append((pc, pk))
    # fine->coarse maps are implied by the keys; the k2s2 conv uses explicit neighbour tables instead
    nbr81 = [R.me_nbr(c, k, R.me_kernel_offsets([3, 3, 3, 3], [1 << L, 1 << L, 1 << L, 1])) for L, (c, k) in
             enumerate(lv)]
    nbr125 = R.me_nbr(coords, keys, R.me_kernel_offsets([5, 5, 5, 1], [1, 1, 1, 1]))
    off2 = [R.me_kernel_offsets([2, 2, 2, 1], [1 << L, 1 << L, 1 << L, 1]) for L in range(3)]
    # strided conv L -> L+1: output coarse voxel reads fine children  p + off
    nbr_dn = [R.me_nbr(lv[L + 1][0], lv[L][1], off2[L], +1) for L in range(3)]
    # transposed conv L+1 -> L: fine voxel f reads coarse p = f - off   (same pairs, reversed)
    nbr_up = [R.me_nbr(lv[L][0], lv[L + 1][1], off2[L], -1) for L in range(3)]
    dbg.update(coords=[c for c, _ in lv], nbr81=nbr81, nbr125=nbr125, nbr_dn=nbr_dn, nbr_up=nbr_up, inverse=inverse)

    def conv(x, nbr, name):
        return R.sparse_conv(x, nbr, _me_taps(sd[ME_P + name + ".kernel"]))

    def bn(x, name):
        return _bn(sd, ME_P + name + ".bn", x, 1e-5)

    def block(x, nbr, name):
        out = _relu(bn(conv(x, nbr, name + ".conv1"), name + ".norm1"))
        out = bn(conv(out, nbr, name + ".conv2"), name + ".norm2")
        res = x
        if (ME_P + name + ".downsample.0.kernel") in sd:
            res = bn(conv(x, None, name + ".downsample.0"), name + ".downsample.1")
        return _relu(out + res)

    out_p1 = _relu(bn(conv(feat, nbr125, "conv0p1s1"), "bn0"))
    out = _relu(bn(conv(out_p1, nbr_dn[0], "conv1p1s2"), "bn1"))
    out_b1p2 = block(out, nbr81[1], "block1.0")
    out = _relu(bn(conv(out_b1p2, nbr_dn[1], "conv2p2s2"), "bn2"))
    out_b2p4 = block(out, nbr81[2], "block2.0")
    out = _relu(bn(conv(out_b2p4, nbr_dn[2], "conv3p4s2"), "bn3"))
    out = block(out, nbr81[3], "block3.0")
    dbg["b3"] = out
    out = _relu(bn(conv(out, nbr_up[2], "convtr5p8s2"), "bntr5"))
    out = block(np.concatenate([out, out_b2p4], 1), nbr81[2], "block6.0")
    out = _relu(bn(conv(out, nbr_up[1], "convtr6p4s2"), "bntr6"))
    out = block(np.concatenate([out, out_b1p2], 1), nbr81[1], "block7.0")
    out = _relu(bn(conv(out, nbr_up[0], "convtr7p2s2"), "bntr7"))
    out = block(np.concatenate([out, out_p1], 1), nbr81[0], "block8.0")
    dbg["b8"] = out
    final = R.sparse_conv(out, None, _me_taps(sd[ME_P + "final.kernel"])) + sd[ME_P + "final.bias"].reshape(1, -1)
    dbg["voxel_motion"] = final
    point_feat = final[inverse]  # .slice(tensor_field), motionnet.py:38
    cur = (pts4[:, 3] / q[3]) == 0  # motionnet.py:42
    current_point = np.concatenate([points[cur, :4], point_feat[cur]], axis=1).astype(np.float32)
    return (current_point, dbg) if want_debug else current_point


# ------------------------------------------------------------------------------------------------
def bev_forward(sd, cfg, dense_bev):
    """BaseBEVBackbone + CenterHead convs on torch CPU (fp32).  Returns NHWC cls (H,W,ncls), box (H,W,8)."""
    import torch
    import torch.nn.functional as F
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

    def bn2d(x, stem):
        return F.batch_norm(x, t(sd[stem + ".running_mean"]), t(sd[stem + ".running_var"]), t(sd[stem + ".weight"]),
                            t(sd[stem + ".bias"]), False, 0.0, 1e-3)

    B = UN_P + "bev_backbone."
    with torch.no_grad():
        x = t(dense_bev)
        x = F.pad(x, (1, 1, 1, 1))
        x = F.relu(bn2d(F.conv2d(x, t(sd[B + "blocks.0.1.weight"])), B + "blocks.0.2"))
        for k in range(cfg["MODEL"]["BACKBONE_2D"]["LAYER_NUMS"][0]):
            x = F.relu(bn2d(F.conv2d(x, t(sd[B + f"blocks.0.{4 + 3 * k}.weight"]), padding=1),
                            B + f"blocks.0.{5 + 3 * k}"))
        s = cfg["MODEL"]["BACKBONE_2D"]["UPSAMPLE_STRIDES"][0]
        x = F.relu(bn2d(F.conv_transpose2d(x, t(sd[B + "deblocks.0.0.weight"]), stride=s), B + "deblocks.0.1"))
        H = UN_P + "center_head."
        cls = F.conv2d(x, t(sd[H + "conv_cls.weight"]), t(sd[H + "conv_cls.bias"]))
        box = F.conv2d(x, t(sd[H + "conv_box.weight"]), t(sd[H + "conv_box.bias"]))
        return (x[0].permute(1, 2, 0).contiguous().numpy(), cls[0].permute(1, 2, 0).contiguous().numpy(),
                box[0].permute(1, 2, 0).contiguous().numpy())


def unet_forward(sd, cfg, current_point, max_voxels=100000, max_points=5, quirk=True, want_debug=False):
    """current_point (Ncur,7) -> (point logits (Ncur,3), pred dict).  spconv_unet.py:267-416."""
    dbg = {}
    vs = cfg["DATA"]["VOXEL_SIZE"]
    rng = cfg["DATA"]["POINT_CLOUD_RANGE"]
    ncls = cfg["MODEL"]["DENSE_HEAD"]["NUM_CLASS"]
    grid = np.round((np.array(rng[3:6], np.float64) - np.array(rng[0:3], np.float64)) / np.array(vs)).astype(np.int64)
    shape1 = [int(grid[2]) + 1, int(grid[1]), int(grid[0])]  # sparse_shape = grid[::-1] + [1,0,0]

    voxels, coords1, num, pc_voxel_id = R.voxelize_with_id(current_point, vs, rng, max_voxels, max_points)
    feat = R.mean_vfe(voxels, num)
    k1, p1 = R.sorted_index(R.key3(coords1, shape1))
    # ---- coordinate maps / rulebooks (one per indice_key, spconv_unet.py:120-207)
    S = {1: (coords1, k1, p1, shape1)}
    for lvl in (2, 3, 4):
        ci, ki, pi, si = S[lvl - 1]
        oc, ok, osz = R.spconv_down_coords(ci, si, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        S[lvl] = (oc, ok, None, osz)
    c4, k4, p4, s4 = S[4]
    c5, k5, s5 = R.spconv_down_coords(c4, s4, (3, 1, 1), (2, 1, 1), (0, 0, 0))
    subm = {l: R.spconv_nbr_subm(S[l][0], S[l][1], S[l][2], S[l][3]) for l in (1, 2, 3, 4)}
    down = {l: R.spconv_nbr_down(S[l][0], S[l - 1][1], S[l - 1][2], S[l - 1][3], (3, 3, 3), (2, 2, 2), (1, 1, 1))
            for l in (2, 3, 4)}
    inv = {l: R.spconv_nbr_inverse(S[l - 1][0], S[l][1], S[l][2], S[l][3], (3, 3, 3), (2, 2, 2), (1, 1, 1))
           for l in (2, 3, 4)}
    down5 = R.spconv_nbr_down(c5, k4, p4, s4, (3, 1, 1), (2, 1, 1), (0, 0, 0))
    inv5 = R.spconv_nbr_inverse(c4, k5, None, s5, (3, 1, 1), (2, 1, 1), (0, 0, 0))
    dbg.update(coords={**{l: S[l][0] for l in S}, 5: c5}, subm=subm, down=down, inv=inv, down5=down5, inv5=inv5,
               pc_voxel_id=pc_voxel_id, voxel_features=feat, num_points=num)

    def conv(x, nbr, stem):
        return R.sparse_conv(x, nbr, _sp_taps(sd[UN_P + stem + ".weight"]))

    def cbr(x, nbr, conv_stem, bn_stem):
        return _relu(_bn(sd, UN_P + bn_stem, conv(x, nbr, conv_stem), 1e-3))

    def blk(x, nbr, stem):  # post_act_block: SparseSequential(conv, bn, relu)
        return cbr(x, nbr, stem + ".0", stem + ".1")

    def basic(x, nbr, stem):  # SparseBasicBlock, spconv_unet.py:71-106
        out = cbr(x, nbr, stem + ".conv1", stem + ".bn1")
        out = _bn(sd, UN_P + stem + ".bn2", conv(out, nbr, stem + ".conv2"), 1e-3)
        return _relu(out + x)

    x = cbr(feat, subm[1], "conv_input.0", "conv_input.1")
    x_conv1 = blk(x, subm[1], "conv1.0")
    xs = {1: x_conv1}
    cur = x_conv1
    for l in (2, 3, 4):
        cur = blk(cur, down[l], f"conv{l}.0")
        cur = blk(cur, subm[l], f"conv{l}.1")
        cur = blk(cur, subm[l], f"conv{l}.2")
        xs[l] = cur
    enc = cbr(xs[4], down5, "conv_out.0", "conv_out.1")
    dbg["x_conv"] = xs
    dbg["encoded"] = enc

    # ---- instance detection
    dense = R.sparse_to_dense_bev(enc, c5, s5)
    f2d, cls_hw, box_hw = bev_forward(sd, cfg, dense)
    tcfg = cfg["MODEL"]["DENSE_HEAD"]["TARGET_ASSIGNER_CONFIG"]
    cls_flat, boxes_flat = R.center_decode(cls_hw, box_hw, tcfg["OUT_SIZE_FACTOR"], tcfg["VOXEL_SIZE"], rng[0:2])
    pp = cfg["MODEL"]["POST_PROCESSING"]
    nc = pp["NMS_CONFIG"]
    pred_boxes, pred_scores, pred_labels, sel = R.post_process(
        cls_flat, boxes_flat, pp["SCORE_THRESH"], nc["NMS_THRESH"], nc["NMS_PRE_MAXSIZE"], nc["NMS_POST_MAXSIZE"])
    pred = {"pred_boxes": pred_boxes, "pred_scores": pred_scores, "pred_labels": pred_labels}
    dbg.update(dense=dense, spatial_features_2d=f2d, cls=cls_flat, boxes=boxes_flat, selected=sel)

    # ---- upsample fusion (spconv_unet.py:319-402)
    sparse_inv_bev = conv(enc, inv5, "inv_conv_out")
    stride = 8
    b = pred_boxes.astype(np.float32).copy()
    # fp32 torch ops in the reference (spconv_unet.py:324-329): (x - range) / voxel / stride, left to
    # right.  torch's CUDA div-by-python-scalar multiplies by the fp32 reciprocal of the scalar
    # (ATen BinaryDivMulKernel: b -> 1/b), which is the path the reference runs on; restated so.
    for d in range(3):
        iv = np.float32(1.0) / np.float32(vs[d])
        istr = np.float32(1.0) / np.float32(stride)
        b[:, d] = ((b[:, d] - np.float32(rng[d])) * iv) * istr
        b[:, 3 + d] = (b[:, 3 + d] * iv) * istr
    boxes8 = np.concatenate([b, pred_labels.astype(np.float32).reshape(-1, 1)], 1).astype(np.float32)

    def inst(level_coords_zyx, boxes8):
        return R.boxes_to_onehot(level_coords_zyx[:, [2, 1, 0]], boxes8, ncls, quirk).astype(np.float32)

    def ur_block(x_lateral, x_bottom, lvl, conv_inv):
        x_trans = basic(x_lateral, subm[lvl], f"conv_up_t{lvl}")
        cat = np.concatenate([x_bottom, x_trans], 1)
        x_m = blk(cat, subm[lvl], f"conv_up_m{lvl}")
        C = x_m.shape[1]
        red = cat.reshape(len(cat), C, -1).sum(axis=2)  # channel_reduction, spconv_unet.py:224-238
        return conv_inv(x_m + red)

    onehots = {}
    oh = inst(S[4][0], boxes8)
    onehots[4] = oh
    x_ci = blk(np.concatenate([sparse_inv_bev, oh], 1), subm[4], "conv_up_instance_block")
    x_up4 = ur_block(x_ci, x_ci, 4, lambda x: blk(x, inv[4], "inv_conv4"))
    boxes8[:, 0:6] *= np.float32(2)
    oh = inst(S[3][0], boxes8)
    onehots[3] = oh
    x_i = blk(np.concatenate([x_up4, oh], 1), subm[3], "conv_up_instance_block_up4")
    x_up3 = ur_block(xs[3], x_i, 3, lambda x: blk(x, inv[3], "inv_conv3"))
    boxes8[:, 0:6] *= np.float32(2)
    oh = inst(S[2][0], boxes8)
    onehots[2] = oh
    x_i = blk(np.concatenate([x_up3, oh], 1), subm[2], "conv_up_instance_block_up3")
    x_up2 = ur_block(xs[2], x_i, 2, lambda x: blk(x, inv[2], "inv_conv2"))
    boxes8[:, 0:6] *= np.float32(2)
    oh = inst(S[1][0], boxes8)
    onehots[1] = oh
    x_i = blk(np.concatenate([x_up2, oh], 1), subm[1], "conv_up_instance_block_up2")
    x_up1 = ur_block(xs[1], x_i, 1, lambda x: blk(x, subm[1], "conv_up_out.0"))
    x_fin = blk(np.concatenate([x_up1, oh], 1), subm[1], "conv_up_instance_block_up1")
    dbg.update(onehots=onehots, x_up={4: x_up4, 3: x_up3, 2: x_up2, 1: x_up1}, seg_feature=x_fin)
    vox_logits = x_fin @ sd[UN_P + "mos_seg_layer.weight"].T.astype(np.float32) + sd[UN_P + "mos_seg_layer.bias"]
    dbg["voxel_logits"] = vox_logits
    # gather_features_by_pc_voxel_id: zeros where the point has no voxel [dep-knowledge]
    logits = np.zeros((len(current_point), vox_logits.shape[1]), np.float32)
    ok = pc_voxel_id >= 0
    logits[ok] = vox_logits[pc_voxel_id[ok]]
    return (logits, pred, dbg) if want_debug else (logits, pred)


def forward_window(sd, cfg, points, want_debug=False, quirk=True, max_voxels=100000):
    """InsMOS_Model.forward for one batch item in 'test' mode (models/models.py:313-364)."""
    dt = cfg["MODEL"]["DELTA_T_PREDICTION"]
    ds = cfg["DATA"]["VOXEL_SIZE"][0]
    if want_debug:
        cp, d1 = motionnet_forward(sd, points, dt, ds, True)
        logits, pred, d2 = unet_forward(sd, cfg, cp, max_voxels=max_voxels, quirk=quirk, want_debug=True)
        return logits, pred, {"motion": d1, "unet": d2, "current_point": cp}
    cp = motionnet_forward(sd, points, dt, ds)
    return unet_forward(sd, cfg, cp, max_voxels=max_voxels, quirk=quirk)

"""The 3D branch (voxelise -> UNetV2 -> BEV CenterHead -> instance-fused decoder) in TRAINING mode on MI355X, and the full
training loss of models/models.py:313-345 built from it and MotionNetTrainer.

Layer order: UNetV2.forward (models/backbones_3d/spconv_unet.py:267-416), SparseBasicBlock (:71-106), UR_block_forward /
channel_reduction (:213-238), HeightCompression (models/backbones_2d/height_compression.py:24-31), BaseBEVBackbone
(base_bev_backbone.py:84-115), CenterHead.forward/get_loss (center_head.py:65-98,279-331).  As in the reference's train
mode the instance boxes come from the head's own (train-mode) output through post_processing, detached
(spconv_unet.py:316-331); voxelisation is not differentiable there either (spconv's PointToVoxel), so the 3D branch
sees MotionNet's features as constants and MotionNet learns from its own loss (models/models.py:321-337).

Every convolution / BatchNorm / loss node is a HIP kernel pair (insmos_amd/autograd.py); torch holds the tape and the glue
(concatenation, residual adds, the BEV scatter, the voxel -> point gather).  Kernel maps are the engine's -- the ones
the inference path builds; a strided layer's transposed map is its inverse-conv twin (down <-> inv).  Parameters are kept
in the kernels' tap layout (K, Cin, Cout); export_state_dict() converts back to the reference's checkpoint layout.
`bn_training=False` runs the same graph with running statistics: it must reproduce the inference path's logits
(tests/test_train_unet.py), which pins the wiring to the oracle-checked forward.
"""
import os

import numpy as np
import torch

from . import params as P
from .autograd import BnPlan, batch_norm_train_seg, center_assign_targets, center_head_loss, gather_rows, mos_loss, sparse_conv
from .engine import Engine


def _np(v):
    return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


class UNetV2Trainer:
    def __init__(self, cfg, state_dict, device="cuda:0", bn_eps=1e-3, bn_momentum=0.01, engine=None):
        self.cfg, self.device = cfg, torch.device(device)
        self.engine = engine if engine is not None else Engine(cfg, state_dict, device)
        self.eps, self.momentum = bn_eps, bn_momentum
        self.bn_training = True
        self.ncls = int(cfg["MODEL"]["DENSE_HEAD"]["NUM_CLASS"])
        self.in_ch = len(cfg["MODEL"]["POINT_FEATURE_ENCODING"]["src_feature_list"]) + 3
        self.head_cfg = cfg["MODEL"]["DENSE_HEAD"]
        self.params, self.buffers, self._layout = {}, {}, {}
        U = P.UNET_PREFIX
        for name, (shape, kind) in P.param_spec(cfg).items():
            if not name.startswith(U):
                continue
            stem = name[len(U):]
            v = np.asarray(_np(state_dict[name]), np.float32)
            if kind == "spconv":
                w = P.spconv_weight_to_taps(v)
            elif kind in ("conv2d", "box_w"):
                w = P.conv2d_weight_to_taps(v)
            elif kind == "convT2d":
                w = P.convT2d_weight_to_taps(v)
            elif kind == "linear":
                w = np.ascontiguousarray(v.T[None])
            else:
                w = v.reshape(-1)
            self._layout[stem] = (kind, tuple(v.shape))
            t = torch.from_numpy(np.ascontiguousarray(w)).to(self.device)
            if kind in ("bn_m", "bn_v"):
                self.buffers[stem] = t.clone()
            else:
                self.params[stem] = t.requires_grad_(True)
        self._deconv_tables = None
        self._nbr9_b = None

    # ---------------------------------------------------------------------------------------------
    def to_reference_layout(self, stem, v):
        """A tensor in this class's tap layout (a parameter, its gradient, a buffer) -> the reference checkpoint's layout."""
        kind, shape = self._layout[stem]
        v = _np(v)
        if kind == "spconv":      # (K, ci, co) -> (co, kz, ky, kx, ci)
            v = v.transpose(2, 0, 1)
        elif kind in ("conv2d", "box_w"):  # (kh*kw, ci, co) -> (co, ci, kh, kw)
            v = v.transpose(2, 1, 0)
        elif kind == "convT2d":   # (kh*kw, ci, co) -> (ci, co, kh, kw)
            v = v.transpose(1, 2, 0)
        elif kind == "linear":    # (1, in, out) -> (out, in)
            v = v[0].T
        return np.ascontiguousarray(v).reshape(shape)

    def export_state_dict(self):
        """Parameters and running statistics back in the reference's checkpoint layout (numpy, UNET_PREFIX names)."""
        return {P.UNET_PREFIX + stem: self.to_reference_layout(stem, self.buffers[stem] if stem in self.buffers
                                                              else self.params[stem]).astype(np.float32)
                for stem in self._layout}

    # ---------------------------------------------------------------------------------------------
    def _bn(self, x, stem, relu, plan):
        g, b = self.params[stem + ".weight"], self.params[stem + ".bias"]
        rm, rv = self.buffers[stem + ".running_mean"], self.buffers[stem + ".running_var"]
        if self.bn_training:
            return batch_norm_train_seg(x, g, b, plan, rm, rv, self.momentum, self.eps, relu)
        y = (x - rm) * (g / torch.sqrt(rv + self.eps)) + b
        return torch.relu(y) if relu else y

    def _conv(self, stem, x, nbr, nbr_t=None, bias=None):
        return sparse_conv(x, self.params[stem + ".weight"], bias, nbr, nbr_t)

    def _cbr(self, conv, bn, x, nbr, plan, nbr_t=None):
        return self._bn(self._conv(conv, x, nbr, nbr_t), bn, True, plan)

    def _cbr_cat(self, conv, bn, x, onehot, nbr, plan):
        """conv over cat([x, onehot]) written as two convs on clean channel widths: the one-hot columns carry no
        gradient, and their taps are the trailing rows of the layer's weight (spconv_unet.py:348,365,380,395,401)."""
        w = self.params[conv + ".weight"]
        c = x.shape[1]
        y = sparse_conv(x, w[:, :c, :], None, nbr) + sparse_conv(onehot, w[:, c:, :], None, nbr)
        return self._bn(y, bn, True, plan)

    def _basic_block(self, stem, x, nbr, plan):
        out = self._bn(self._conv(stem + ".conv1", x, nbr), stem + ".bn1", True, plan)
        out = self._bn(self._conv(stem + ".conv2", out, nbr), stem + ".bn2", False, plan)
        return torch.relu(out + x)

    def _ur_block(self, lvl, x_lat, x_bottom, nbr, plan):
        """UR_block_forward up to conv_inv (spconv_unet.py:213-219)."""
        trans = self._basic_block(f"conv_up_t{lvl}", x_lat, nbr, plan)
        cat = torch.cat([x_bottom, trans], 1)
        m = self._cbr(f"conv_up_m{lvl}.0", f"conv_up_m{lvl}.1", cat, nbr, plan)
        return m + cat.view(cat.shape[0], m.shape[1], -1).sum(2)

    def _nbr9(self, B):
        """The dense 3x3 table of B stacked BEV images (a tap never leaves its image)."""
        eng = self.engine
        if B == 1:
            return eng.nbr_bev
        if self._nbr9_b is None or self._nbr9_b[0] != B:
            from . import _lib
            t = torch.empty((9, B * eng.bevH * eng.bevW), dtype=torch.int32, device=self.device)
            _lib.check(eng.lib.insmos_dense_nbr2d_b(eng.bevH, eng.bevW, B, t.data_ptr(), eng._stream()), "insmos_dense_nbr2d_b")
            self._nbr9_b = (B, t)
        return self._nbr9_b[1]

    def _deconv_nbr(self, B=1):
        """ConvTranspose2d(k=2, s=2) as a 4-tap layer whose output rows are the B (2H, 2W) maps in [b][row][col] order:
        out[b, Y, X] = in[b, Y // 2, X // 2] @ W[:, :, Y % 2, X % 2] (base_bev_backbone.py:49-58), and its transposed table."""
        if self._deconv_tables is None or self._deconv_tables[0] != B:
            H, W = self.engine.bevH, self.engine.bevW
            bb, Y, X = torch.meshgrid(torch.arange(B), torch.arange(2 * H), torch.arange(2 * W), indexing="ij")
            site = (bb * (H * W) + (Y // 2) * W + X // 2).reshape(-1)
            k = ((Y % 2) * 2 + X % 2).reshape(-1)
            o = torch.arange(4 * H * W * B)
            nbr = torch.full((4, 4 * H * W * B), -1, dtype=torch.int32)
            nbr[k, o] = site.int()
            nbr_t = torch.full((4, H * W * B), -1, dtype=torch.int32)
            nbr_t[k, site] = o.int()
            self._deconv_tables = (B, nbr.to(self.device), nbr_t.to(self.device))
        return self._deconv_tables[1:]

    # ---------------------------------------------------------------------------------------------
    def forward(self, cur, gt_boxes=None):
        """One window: cur (Ncur, 8) fp32 device = current_point [x, y, z, r, m0, m1, m2, 0] (motionnet.py:48; treated as
        constant) -> dict(point_logits (Ncur, 3), cls_preds (1, 2H, 2W, C), box_preds (1, 2H, 2W, 8), pred boxes of this pass,
        targets when gt_boxes (1, M, 8) is given)."""
        outs = self.forward_windows([cur], None if gt_boxes is None else [gt_boxes])
        o = outs[0]
        o["pred_dicts"] = [o["pred_dicts"]]
        return o

    def forward_windows(self, cur_list, gt_boxes_list=None):
        """The batch items of one training step in ONE set of launches (the reference walks them one by one,
        models/models.py:313-345): window = spconv's batch column, voxel rows window-major, BEV images stacked, and every
        BatchNorm keeps per-window statistics (BnPlan), so each window gets what the item-by-item walk gives it.
        -> per window dict(point_logits, cls_preds (1, 2H, 2W, C), box_preds (1, 2H, 2W, 8), pred_dicts, targets)."""
        eng, p = self.engine, self.params
        B = len(cur_list)
        n_cur = [int(c.shape[0]) for c in cur_list]
        cur = cur_list[0].detach() if B == 1 else torch.cat([c.detach() for c in cur_list], 0).contiguous()
        with torch.no_grad():
            T = eng.unet_tables_windows(cur, n_cur)      # voxelisation, coordinate sets and kernel maps only
        # (NbrTable objects: sparse_conv uses their active-tap masks too)
        subm, down, inv = dict(T["subm"]), dict(T["down"]), dict(T["inv"])
        down5, inv5, coords = T["down5"], T["inv5"], T["coords"]
        nv = {l: int(coords[l].shape[0]) for l in (1, 2, 3, 4, 5)}
        wr = T["win_rows"]
        plan = {l: BnPlan([(int(wr[l][b]), int(wr[l][b + 1]), b) for b in range(B) if wr[l][b + 1] > wr[l][b]], nv[l], B, self.device)
                for l in (1, 2, 3, 4, 5)}
        feat = T["feat"][:, :self.in_ch].detach()

        # ---- encoder (spconv_unet.py:297-306)
        x0 = self._cbr("conv_input.0", "conv_input.1", feat, subm[1], plan[1])
        xc = {1: self._cbr("conv1.0.0", "conv1.0.1", x0, subm[1], plan[1])}
        for l in (2, 3, 4):
            a = self._cbr(f"conv{l}.0.0", f"conv{l}.0.1", xc[l - 1], down[l], plan[l], inv[l])
            b_ = self._cbr(f"conv{l}.1.0", f"conv{l}.1.1", a, subm[l], plan[l])
            xc[l] = self._cbr(f"conv{l}.2.0", f"conv{l}.2.1", b_, subm[l], plan[l])
        enc = self._cbr("conv_out.0", "conv_out.1", xc[4], down5, plan[5], inv5)

        # ---- BEV detection head, NHWC rows, B images stacked (height_compression.py:24-31, base_bev_backbone.py:84-115)
        D, H, W = eng.bevD, eng.bevH, eng.bevW
        HW = H * W
        c5 = coords[5].long()
        bev = torch.zeros((B * HW, enc.shape[1] * D), dtype=torch.float32, device=self.device)
        rows = (c5[:, 0] * HW + c5[:, 2] * W + c5[:, 3])[:, None]
        cols = torch.arange(enc.shape[1], device=self.device)[None, :] * D + c5[:, 1:2]
        bev[rows, cols] = enc
        nbr9 = self._nbr9(B)
        dn, dn_t = self._deconv_nbr(B)
        plan_bev = BnPlan([(b * HW, (b + 1) * HW, b) for b in range(B)], B * HW, B, self.device)
        plan_up = BnPlan([(b * 4 * HW, (b + 1) * 4 * HW, b) for b in range(B)], B * 4 * HW, B, self.device)
        Bn = "bev_backbone."
        f = self._cbr(Bn + "blocks.0.1", Bn + "blocks.0.2", bev, nbr9, plan_bev)
        for k in range(eng.n_bev_layers):
            f = self._cbr(Bn + f"blocks.0.{4 + 3 * k}", Bn + f"blocks.0.{5 + 3 * k}", f, nbr9, plan_bev)
        up = self._cbr(Bn + "deblocks.0.0", Bn + "deblocks.0.1", f, dn, plan_up, dn_t)    # (B * 4HW, 256), rows [b][Y][X]
        cls = self._conv("center_head.conv_cls", up, None, None, p["center_head.conv_cls.bias"])
        box = self._conv("center_head.conv_box", up, None, None, p["center_head.conv_box.bias"])
        cls_maps, box_maps = cls.view(B, 2 * H, 2 * W, self.ncls), box.view(B, 2 * H, 2 * W, 8)

        # ---- post_processing on this pass's head output, detached (spconv_unet.py:314-331)
        with torch.no_grad():
            head = torch.zeros((B * 4 * HW, eng.head_ld), dtype=torch.float32, device=self.device)
            head[:, :self.ncls] = cls
            head[:, self.ncls:self.ncls + 8] = box
            pb, psc, pl, cnt_k, _ = eng.detect_windows(head, 1, B)
            scratch = torch.empty((int(eng.lib.insmos_boxes_to_onehot_scratch_ints_b(eng.post_max, B, max(nv.values()))),),
                                  dtype=torch.int32, device=self.device)

            def onehot(level, mult):
                oh = torch.zeros((nv[level], 16), dtype=torch.float32, device=self.device)  # the kernel pads to 16 columns
                eng.instance_onehot_windows(pb, pl, cnt_k, B, coords[level], nv[level], mult, oh, 16, 0, scratch)
                return oh[:, :self.ncls]

            oh = {4: onehot(4, 1.0), 3: onehot(3, 2.0), 2: onehot(2, 4.0), 1: onehot(1, 8.0)}
            K = cnt_k[:, 0].cpu().numpy()

        # ---- upsample fusion (spconv_unet.py:319-402)
        x = self._conv("inv_conv_out", enc, inv5, down5)
        x = self._cbr_cat("conv_up_instance_block.0", "conv_up_instance_block.1", x, oh[4], subm[4], plan[4])
        m = self._ur_block(4, x, x, subm[4], plan[4])
        x = self._cbr("inv_conv4.0", "inv_conv4.1", m, inv[4], plan[3], down[4])
        x = self._cbr_cat("conv_up_instance_block_up4.0", "conv_up_instance_block_up4.1", x, oh[3], subm[3], plan[3])
        m = self._ur_block(3, xc[3], x, subm[3], plan[3])
        x = self._cbr("inv_conv3.0", "inv_conv3.1", m, inv[3], plan[2], down[3])
        x = self._cbr_cat("conv_up_instance_block_up3.0", "conv_up_instance_block_up3.1", x, oh[2], subm[2], plan[2])
        m = self._ur_block(2, xc[2], x, subm[2], plan[2])
        x = self._cbr("inv_conv2.0", "inv_conv2.1", m, inv[2], plan[1], down[2])
        x = self._cbr_cat("conv_up_instance_block_up2.0", "conv_up_instance_block_up2.1", x, oh[1], subm[1], plan[1])
        m = self._ur_block(1, xc[1], x, subm[1], plan[1])
        x = self._cbr("conv_up_out.0.0", "conv_up_out.0.1", m, subm[1], plan[1])
        seg = self._cbr_cat("conv_up_instance_block_up1.0", "conv_up_instance_block_up1.1", x, oh[1], subm[1], plan[1])
        vox = sparse_conv(seg, p["mos_seg_layer.weight"], p["mos_seg_layer.bias"], None)      # Linear(16 -> 3)
        pcid = T["pcid"][:cur.shape[0]]
        point_logits = gather_rows(vox, pcid.clamp(min=0)) * (pcid >= 0)[:, None].float()  # dropped points get zeros (:410)
        pl_split = torch.split(point_logits, n_cur, 0)
        outs = []
        for b in range(B):
            o = {"cls_preds": cls_maps[b:b + 1], "box_preds": box_maps[b:b + 1], "point_logits": pl_split[b],
                 "pred_dicts": {"pred_boxes": pb[b, :int(K[b])], "pred_scores": psc[b, :int(K[b])], "pred_labels": pl[b, :int(K[b])]}}
            if gt_boxes_list is not None:
                # grid / range exactly as models/models.py:277-280 builds them (an integer range list stays integer: the
                # dtype decides float32 vs float64 cell arithmetic in the reference, see insmos_center_assign_targets)
                pcr = np.array(self.cfg["DATA"]["POINT_CLOUD_RANGE"])
                grid = np.round((pcr[3:6] - pcr[0:3]) / np.array(self.cfg["DATA"]["VOXEL_SIZE"])).astype(np.int64)
                o["targets"] = center_assign_targets(gt_boxes_list[b], self.head_cfg, grid, pcr, self.ncls)
            outs.append(o)
        self._last_tables = T
        return outs

    def loss(self, cur, gt_boxes, gt_labels_cur):
        """(loss_rpn + loss_mos, tb_dict) of models/models.py:328-343 for one batch item."""
        out = self.forward(cur, gt_boxes)
        loss_rpn, tb = center_head_loss(out["cls_preds"], out["box_preds"], out["targets"], self.head_cfg)
        loss_mos = mos_loss(out["point_logits"], gt_labels_cur, 3, (0,))
        tb = dict(tb)
        tb["loss_mos"] = float(loss_mos.detach())
        return loss_rpn + loss_mos, tb, out


class InsMOSTrainer:
    """InsMOS_Model.forward(list, 'train') (models/models.py:297-345, 365-367): MotionNet and the 3D branch in train mode,
    loss = mean over the batch items of loss_rpn + loss_mos (+ loss_motion_encoder when MODEL.USE_MOTION_LOSS)."""

    def __init__(self, cfg, state_dict, device="cuda:0", bf16_convs=None):
        """bf16_convs (default: INSMOS_TRAIN_BF16=1 in the environment, else off): the training convolutions' forward and d/dx
        round their operands to bf16 and accumulate in fp32 (autograd.train_conv_precision, per instance); everything else, and the
        inference path always, stays fp32.  The reference trains in fp32 only (config/config.yaml has no precision key)."""
        from .train_motionnet import MotionNetTrainer
        from . import autograd
        if bf16_convs is None:
            bf16_convs = os.environ.get("INSMOS_TRAIN_BF16", "0") == "1"
        self.bf16_convs = bool(bf16_convs)   # per trainer instance: forward() wraps its nodes in train_conv_precision(...)
        self.cfg, self.device = cfg, torch.device(device)
        self.use_motion_loss = bool(cfg["MODEL"].get("USE_MOTION_LOSS", False))
        self.motion = MotionNetTrainer(cfg, state_dict, device)
        self.unet = UNetV2Trainer(cfg, state_dict, device, engine=self.motion.engine)  # one set of kernel-map builders
        self.dt = float(cfg["MODEL"]["DELTA_T_PREDICTION"])

    @property
    def params(self):
        d = {"motion." + k: v for k, v in self.motion.params.items()}
        d.update({"unet." + k: v for k, v in self.unet.params.items()})
        return d

    def forward(self, list_batch_dict, Model_mode="train"):
        if Model_mode != "train":
            raise ValueError("InsMOSTrainer serves Model_mode == 'train'; use InsMOS_Model for 'test' / 'eval'")
        from . import autograd
        with autograd.train_conv_precision(1 if self.bf16_convs else 0):
            return self._forward_train(list_batch_dict)

    def _forward_train(self, list_batch_dict):
        """All batch items in one set of launches per branch (windows_per_step = len(list): the reference trains with 4-6 per GPU,
        README.md:195); per-window BatchNorm statistics and per-window losses keep the item-by-item semantics of
        models/models.py:313-345.  INSMOS_TRAIN_SEQUENTIAL=1 walks the items one by one (the same numbers, B x the launches)."""
        if os.environ.get("INSMOS_TRAIN_SEQUENTIAL", "0") == "1" and len(list_batch_dict) > 1:
            parts = [self._forward_train([b]) for b in list_batch_dict]
            loss = sum(pt[0] for pt in parts) / len(parts)
            return loss, [pt[1][0] for pt in parts], [pt[2][0] for pt in parts], [pt[3][0] for pt in parts]
        B = len(list_batch_dict)
        pts_list = [b["past_point_clouds"] for b in list_batch_dict]
        gts = [b["past_labels"][-1] for b in list_batch_dict]
        motions = self.motion.forward_windows(pts_list)                          # B x (Ncur_b, 3), differentiable
        curs = []
        for pts, motion in zip(pts_list, motions):
            cur_rows = torch.nonzero((pts[:, 4] / self.dt) == 0).flatten()
            cur = torch.zeros((cur_rows.shape[0], 8), dtype=torch.float32, device=self.device)
            cur[:, :4] = pts[cur_rows, :4]
            cur[:, 4:7] = motion.detach()                                       # voxelisation cuts the tape (see module doc)
            curs.append(cur)
        outs = self.unet.forward_windows(curs, [b["gt_boxes"] for b in list_batch_dict])
        loss = torch.zeros(1, device=self.device)
        reported, gt_list, pred_list = [], [], []
        for b in range(B):
            loss_motion = mos_loss(motions[b], gts[b], 3, (0,))
            loss_rpn, rp = center_head_loss(outs[b]["cls_preds"], outs[b]["box_preds"], outs[b]["targets"], self.unet.head_cfg,
                                            as_tensors=True)
            loss_mos = mos_loss(outs[b]["point_logits"], gts[b], 3, (0,))
            loss = loss + loss_rpn + loss_mos + (loss_motion if self.use_motion_loss else 0.0)
            reported.append(torch.cat([rp, loss_mos.detach().reshape(1), loss_motion.detach().reshape(1)]))
            gt_list.append(gts[b])
            pred_list.append(outs[b]["point_logits"])
        host = torch.stack(reported).cpu().numpy()                             # ONE read-back for everything that is reported
        train_loss_dict = [{"rpn_loss_cls": float(h[0]), "rpn_loss_loc": float(h[1]), "rpn_loss": float(h[2]), "loss_mos": float(h[3]),
                            "loss_motion_encoder": float(h[4])} for h in host]
        self.last_pred_dicts = [o["pred_dicts"] for o in outs]
        return loss / B, train_loss_dict, gt_list, pred_list

    def make_reducer(self, bucket_bytes=8 << 20, overlap=True, force_collective=False):
        from .ddp import BucketedGradReducer
        # overlap: buckets are all-reduced while backward is still producing the earlier layers' gradients (the parameter
        # dict is in forward order: MotionNet first, the 3D branch's decoder last -- the reverse is the arrival order)
        if getattr(self, "_reducer", None) is not None:
            self._reducer.close()   # one live reducer per parameter set: its backward hooks would launch stray collectives
        self._reducer = BucketedGradReducer(self.params, bucket_bytes, overlap=overlap, force_collective=force_collective)
        return self._reducer

    def sgd_step(self, lr):
        with torch.no_grad():
            for v in self.params.values():
                if v.grad is not None:
                    v -= lr * v.grad
                    v.grad = None

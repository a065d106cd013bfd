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
from .autograd import batch_norm_train, center_assign_targets, center_head_loss, mos_loss, sparse_conv
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
    def _bn(self, x, stem, relu):
        g, b = self.params[stem + ".weight"], self.params[stem + ".bias"]
        rm, rv = self.buffers[stem + ".running_mean"], self.buffers[stem + ".running_var"]
        if self.bn_training:
            return batch_norm_train(x, g, b, rm, rv, self.momentum, self.eps, relu)
        y = (x - rm) * (g / torch.sqrt(rv + self.eps)) + b
        return torch.relu(y) if relu else y

    def _conv(self, stem, x, nbr, nbr_t=None, bias=None):
        return sparse_conv(x, self.params[stem + ".weight"], bias, nbr, nbr_t)

    def _cbr(self, conv, bn, x, nbr, nbr_t=None):
        return self._bn(self._conv(conv, x, nbr, nbr_t), bn, True)

    def _cbr_cat(self, conv, bn, x, onehot, nbr):
        """conv over cat([x, onehot]) written as two convs on clean channel widths: the one-hot columns carry no
        gradient, and their taps are the trailing rows of the layer's weight (spconv_unet.py:348,365,380,395,401)."""
        w = self.params[conv + ".weight"]
        c = x.shape[1]
        y = sparse_conv(x, w[:, :c, :], None, nbr) + sparse_conv(onehot, w[:, c:, :], None, nbr)
        return self._bn(y, bn, True)

    def _basic_block(self, stem, x, nbr):
        out = self._bn(self._conv(stem + ".conv1", x, nbr), stem + ".bn1", True)
        out = self._bn(self._conv(stem + ".conv2", out, nbr), stem + ".bn2", False)
        return torch.relu(out + x)

    def _ur_block(self, lvl, x_lat, x_bottom, nbr):
        """UR_block_forward up to conv_inv (spconv_unet.py:213-219)."""
        trans = self._basic_block(f"conv_up_t{lvl}", x_lat, nbr)
        cat = torch.cat([x_bottom, trans], 1)
        m = self._cbr(f"conv_up_m{lvl}.0", f"conv_up_m{lvl}.1", cat, nbr)
        return m + cat.view(cat.shape[0], m.shape[1], -1).sum(2)

    def _deconv_nbr(self):
        """ConvTranspose2d(k=2, s=2) as a 4-tap layer whose output rows are the (2H, 2W) map in [row][col] order:
        out[Y, X] = in[Y // 2, X // 2] @ W[:, :, Y % 2, X % 2] (base_bev_backbone.py:49-58)."""
        if self._deconv_tables is None:
            H, W = self.engine.bevH, self.engine.bevW
            Y, X = torch.meshgrid(torch.arange(2 * H), torch.arange(2 * W), indexing="ij")
            site = ((Y // 2) * W + X // 2).reshape(-1)
            k = ((Y % 2) * 2 + X % 2).reshape(-1)
            o = torch.arange(4 * H * W)
            nbr = torch.full((4, 4 * H * W), -1, dtype=torch.int32)
            nbr[k, o] = site.int()
            nbr_t = torch.full((4, H * W), -1, dtype=torch.int32)
            nbr_t[k, site] = o.int()
            self._deconv_tables = (nbr.to(self.device), nbr_t.to(self.device))
        return self._deconv_tables

    # ---------------------------------------------------------------------------------------------
    def forward(self, cur, gt_boxes=None):
        """cur (Ncur, 8) fp32 device = current_point [x, y, z, r, m0, m1, m2, 0] (motionnet.py:48; treated as constant)
        -> dict(point_logits (Ncur, 3), cls_preds (1, 2H, 2W, C), box_preds (1, 2H, 2W, 8), pred boxes of this pass,
        targets when gt_boxes (1, M, 8) is given)."""
        eng, p = self.engine, self.params
        with torch.no_grad():
            eng.tables_only = True   # voxelisation, coordinate sets and kernel maps only
            try:
                eng.unet(cur.detach())
            finally:
                eng.tables_only = False
        T = eng._un_tables
        # (NbrTable objects: sparse_conv uses their active-tap masks too)
        subm, down, inv = dict(T["subm"]), dict(T["down"]), dict(T["inv"])
        down5, inv5, coords = T["down5"], T["inv5"], T["coords"]
        nv = {l: int(coords[l].shape[0]) for l in (1, 2, 3, 4, 5)}
        feat = T["feat"][:, :self.in_ch].detach()

        # ---- encoder (spconv_unet.py:297-306)
        x0 = self._cbr("conv_input.0", "conv_input.1", feat, subm[1])
        xc = {1: self._cbr("conv1.0.0", "conv1.0.1", x0, subm[1])}
        for l in (2, 3, 4):
            a = self._cbr(f"conv{l}.0.0", f"conv{l}.0.1", xc[l - 1], down[l], inv[l])
            b = self._cbr(f"conv{l}.1.0", f"conv{l}.1.1", a, subm[l])
            xc[l] = self._cbr(f"conv{l}.2.0", f"conv{l}.2.1", b, subm[l])
        enc = self._cbr("conv_out.0", "conv_out.1", xc[4], down5, inv5)

        # ---- BEV detection head, NHWC rows (height_compression.py:24-31, base_bev_backbone.py:84-115)
        D, H, W = eng.bevD, eng.bevH, eng.bevW
        c5 = coords[5].long()
        bev = torch.zeros((H * W, enc.shape[1] * D), dtype=torch.float32, device=self.device)
        rows = (c5[:, 2] * W + c5[:, 3])[:, None]
        cols = torch.arange(enc.shape[1], device=self.device)[None, :] * D + c5[:, 1:2]
        bev[rows, cols] = enc
        B = "bev_backbone."
        f = self._cbr(B + "blocks.0.1", B + "blocks.0.2", bev, eng.nbr_bev)
        for k in range(eng.n_bev_layers):
            f = self._cbr(B + f"blocks.0.{4 + 3 * k}", B + f"blocks.0.{5 + 3 * k}", f, eng.nbr_bev)
        dn, dn_t = self._deconv_nbr()
        up = self._cbr(B + "deblocks.0.0", B + "deblocks.0.1", f, dn, dn_t)           # (4HW, 256), rows [Y][X]
        cls = self._conv("center_head.conv_cls", up, None, None, p["center_head.conv_cls.bias"])
        box = self._conv("center_head.conv_box", up, None, None, p["center_head.conv_box.bias"])
        out = {"cls_preds": cls.view(1, 2 * H, 2 * W, self.ncls), "box_preds": box.view(1, 2 * H, 2 * W, 8)}

        # ---- post_processing on this pass's head output, detached (spconv_unet.py:314-331)
        with torch.no_grad():
            head = torch.zeros((4 * H * W, eng.head_ld), dtype=torch.float32, device=self.device)
            head[:, :self.ncls] = cls
            head[:, self.ncls:self.ncls + 8] = box
            pb, psc, pl, cnt_k = eng.detect(head, up=1)[:4]
            scratch = torch.empty((int(eng.lib.insmos_boxes_to_onehot_scratch_ints(eng.post_max, max(nv.values()))),),
                                  dtype=torch.int32, device=self.device)

            def onehot(level, mult):
                oh = torch.zeros((nv[level], 16), dtype=torch.float32, device=self.device)  # the kernel pads to 16 columns
                eng.instance_onehot(pb, pl, cnt_k, coords[level], nv[level], mult, oh, 16, 0, scratch)
                return oh[:, :self.ncls]

            oh = {4: onehot(4, 1.0), 3: onehot(3, 2.0), 2: onehot(2, 4.0), 1: onehot(1, 8.0)}
            K = int(cnt_k[0].item())
        out["pred_dicts"] = [{"pred_boxes": pb[:K], "pred_scores": psc[:K], "pred_labels": pl[:K]}]

        # ---- upsample fusion (spconv_unet.py:319-402)
        x = self._conv("inv_conv_out", enc, inv5, down5)
        x = self._cbr_cat("conv_up_instance_block.0", "conv_up_instance_block.1", x, oh[4], subm[4])
        m = self._ur_block(4, x, x, subm[4])
        x = self._cbr("inv_conv4.0", "inv_conv4.1", m, inv[4], down[4])
        x = self._cbr_cat("conv_up_instance_block_up4.0", "conv_up_instance_block_up4.1", x, oh[3], subm[3])
        m = self._ur_block(3, xc[3], x, subm[3])
        x = self._cbr("inv_conv3.0", "inv_conv3.1", m, inv[3], down[3])
        x = self._cbr_cat("conv_up_instance_block_up3.0", "conv_up_instance_block_up3.1", x, oh[2], subm[2])
        m = self._ur_block(2, xc[2], x, subm[2])
        x = self._cbr("inv_conv2.0", "inv_conv2.1", m, inv[2], down[2])
        x = self._cbr_cat("conv_up_instance_block_up2.0", "conv_up_instance_block_up2.1", x, oh[1], subm[1])
        m = self._ur_block(1, xc[1], x, subm[1])
        x = self._cbr("conv_up_out.0.0", "conv_up_out.0.1", m, subm[1])
        seg = self._cbr_cat("conv_up_instance_block_up1.0", "conv_up_instance_block_up1.1", x, oh[1], subm[1])
        vox = sparse_conv(seg, p["mos_seg_layer.weight"], p["mos_seg_layer.bias"], None)      # Linear(16 -> 3)
        pcid = T["pcid"][:cur.shape[0]]
        out["point_logits"] = vox[pcid.clamp(min=0)] * (pcid >= 0)[:, None].float()  # dropped points get zeros (:410)
        if gt_boxes is not None:
            # grid / range exactly as models/models.py:277-280 builds them (an integer range list stays integer: the
            # dtype decides float32 vs float64 cell arithmetic in the reference, see insmos_center_assign_targets)
            pcr = np.array(self.cfg["DATA"]["POINT_CLOUD_RANGE"])
            grid = np.round((pcr[3:6] - pcr[0:3]) / np.array(self.cfg["DATA"]["VOXEL_SIZE"])).astype(np.int64)
            out["targets"] = center_assign_targets(gt_boxes, self.head_cfg, grid, pcr, self.ncls)
        return out

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
        loss = torch.zeros(1, device=self.device)
        train_loss_dict, gt_list, pred_list = [], [], []
        for b in list_batch_dict:
            pts = b["past_point_clouds"]
            gt = b["past_labels"][-1]
            motion = self.motion.forward(pts)                                   # (Ncur, 3), differentiable
            loss_motion = mos_loss(motion, gt, 3, (0,))
            cur_rows = torch.nonzero((pts[:, 4] / self.dt) == 0).flatten()
            cur = torch.zeros((cur_rows.shape[0], 8), dtype=torch.float32, device=self.device)
            cur[:, :4] = pts[cur_rows, :4]
            cur[:, 4:7] = motion.detach()                                       # voxelisation cuts the tape (see module doc)
            l3d, tb, out = self.unet.loss(cur, b["gt_boxes"], gt)
            loss = loss + l3d + (loss_motion if self.use_motion_loss else 0.0)
            tb["loss_motion_encoder"] = float(loss_motion.detach())
            train_loss_dict.append(tb)
            gt_list.append(gt)
            pred_list.append(out["point_logits"])
        return loss / len(list_batch_dict), train_loss_dict, gt_list, pred_list

    def make_reducer(self, bucket_bytes=8 << 20, overlap=True):
        from .ddp import BucketedGradReducer
        # overlap: buckets are all-reduced while backward is still producing the earlier layers' gradients (the parameter
        # dict is in forward order: MotionNet first, the 3D branch's decoder last -- the reverse is the arrival order)
        if getattr(self, "_reducer", None) is not None:
            self._reducer.close()   # one live reducer per parameter set: its backward hooks would launch stray collectives
        self._reducer = BucketedGradReducer(self.params, bucket_bytes, overlap=overlap)
        return self._reducer

    def sgd_step(self, lr):
        with torch.no_grad():
            for v in self.params.values():
                if v.grad is not None:
                    v -= lr * v.grad
                    v.grad = None

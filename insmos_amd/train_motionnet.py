"""MotionNet in TRAINING mode on MI355X: forward with batch-statistics BatchNorm, the motion-encoder loss
(models/models.py:321-324: MOSLoss on the current points' motion logits) and its backward through all 25 sparse
convolutions -- the `backbones_3d + loss.py backward` part of BASELINE.json configs[4] for the 4D branch.

Layer order: CustomMinkUNet.forward (models/MinkowskiEngine/minkunet.py:139-181), BasicBlock (:63-124), the slice back
to points and the current-scan selection of motionnet.py:36-48.  Every convolution / BatchNorm / loss node is a HIP
kernel pair (insmos_amd/autograd.py); torch holds the tape, the ReLU-free glue (residual adds, channel concatenation,
the point gather) and the parameters.  The kernel maps are the engine's (the same tables the inference path uses; the
transposed map of a strided layer is its transposed-conv twin, dn <-> up).  What is NOT here: the 3D branch
(voxelise -> UNetV2 -> CenterHead) in train mode, target assignment and the detection losses, the optimiser, DDP.
"""
import torch

from . import params as P
from .autograd import BnPlan, batch_norm_train_seg, gather_rows, mos_loss, sparse_conv
from .engine import Engine


class MotionNetTrainer:
    def __init__(self, cfg, state_dict, device="cuda:0", bn_eps=1e-5, bn_momentum=0.1):
        self.cfg, self.device = cfg, torch.device(device)
        self.engine = Engine(cfg, state_dict, device)
        self.engine.const_input = False      # the generic first layer: its 125-tap table is needed for d/dW
        self.engine.prune_dead_rows = False  # batch statistics see every voxel
        self.eps, self.momentum = bn_eps, bn_momentum
        self.dt = float(cfg["MODEL"]["DELTA_T_PREDICTION"])
        M = P.ME_PREFIX
        self.params, self.buffers = {}, {}

        def t(name, shape=None):
            v = state_dict[M + name]
            v = v.detach().cpu().numpy() if torch.is_tensor(v) else v
            w = torch.as_tensor(P.me_kernel_to_taps(v) if name.endswith(".kernel") else v, dtype=torch.float32)
            return w.reshape(shape) if shape is not None else w

        def conv(name):
            self.params[name + ".kernel"] = t(name + ".kernel").to(self.device).requires_grad_(True)

        def bn(name, c):
            self.params[name + ".weight"] = t(name + ".bn.weight").to(self.device).requires_grad_(True)
            self.params[name + ".bias"] = t(name + ".bn.bias").to(self.device).requires_grad_(True)
            self.buffers[name + ".running_mean"] = t(name + ".bn.running_mean").to(self.device).clone()
            self.buffers[name + ".running_var"] = t(name + ".bn.running_var").to(self.device).clone()

        for cname, bname, kv, ci, co in P.ME_CONVS:
            conv(cname)
            bn(bname, co)
        for name, ci, co in P.ME_BLOCKS:
            conv(name + ".conv1")
            bn(name + ".norm1", co)
            conv(name + ".conv2")
            bn(name + ".norm2", co)
            if ci != co:
                conv(name + ".downsample.0")
                bn(name + ".downsample.1", co)
        conv("final")
        self.params["final.bias"] = t("final.bias", (-1,)).to(self.device).requires_grad_(True)

    # ---------------------------------------------------------------------------------------------
    def _bn(self, x, name, relu, plan):
        return batch_norm_train_seg(x, self.params[name + ".weight"], self.params[name + ".bias"], plan,
                                    self.buffers[name + ".running_mean"], self.buffers[name + ".running_var"], self.momentum,
                                    self.eps, relu)

    def _block(self, name, x, nbr, plan):
        """BasicBlock (minkunet.py:63-124): conv-bn-relu, conv-bn, + (downsample(x) | x), relu."""
        p = self.params
        out = self._bn(sparse_conv(x, p[name + ".conv1.kernel"], None, nbr), name + ".norm1", True, plan)
        out = self._bn(sparse_conv(out, p[name + ".conv2.kernel"], None, nbr), name + ".norm2", False, plan)
        if (name + ".downsample.0.kernel") in p:
            res = self._bn(sparse_conv(x, p[name + ".downsample.0.kernel"], None, None), name + ".downsample.1", False, plan)
        else:
            res = x
        return torch.relu(out + res)

    def forward(self, pts):
        """pts (N, 5) fp32 device [x, y, z, intensity, t] -> current-point motion logits (Ncur, 3) (motionnet.py:46)."""
        return self.forward_windows([pts])[0]

    def forward_windows(self, pts_list):
        """The batch items of one training step in ONE set of launches (the reference walks them one by one, models/models.py:313):
        the windows share every table and every convolution launch (window index folded into the time coordinate), and
        every BatchNorm keeps per-window statistics (BnPlan: a window's rows are the runs t' = t * B + b with the same b), so each
        window's output -- and the gradient of the summed loss -- is what the item-by-item walk gives.
        -> list of current-point motion logits (Ncur_b, 3)."""
        eng, p = self.engine, self.params
        B = len(pts_list)
        if B == 1:
            pts = pts_list[0]
            win_sizes = None
        else:
            pts = torch.cat([q[:, :5] for q in pts_list], 0).contiguous()
            win_sizes = [int(q.shape[0]) for q in pts_list]
        eng.tables_only = True   # coordinate sets and kernel maps only: the inference convolutions are not needed here
        try:
            eng.motionnet(pts, win_sizes)
        finally:
            eng.tables_only = False
        T = eng._me_tables
        nbr125, n81 = T["nbr125"], list(T["nbr81"])     # NbrTable objects: sparse_conv uses their active-tap masks too
        dn, up = list(T["dn"]), list(T["up"])
        n0 = n81[0].nbr.shape[1]
        if B == 1:
            plans = [BnPlan.whole(int(T["coords"][l].shape[0]), self.device) for l in range(4)]
        else:   # the window of a row: its time coordinate t' = t * B + b modulo B
            plans = [BnPlan.from_segment_ids(torch.remainder(T["coords"][l][:, 3], B), B) for l in range(4)]
        x = torch.full((n0, 1), 0.5, dtype=torch.float32, device=self.device)  # motionnet.py:29-32
        out_p1 = self._bn(sparse_conv(x, p["conv0p1s1.kernel"], None, nbr125), "bn0", True, plans[0])
        out = self._bn(sparse_conv(out_p1, p["conv1p1s2.kernel"], None, dn[0], up[0]), "bn1", True, plans[1])
        out_b1p2 = self._block("block1.0", out, n81[1], plans[1])
        out = self._bn(sparse_conv(out_b1p2, p["conv2p2s2.kernel"], None, dn[1], up[1]), "bn2", True, plans[2])
        out_b2p4 = self._block("block2.0", out, n81[2], plans[2])
        out = self._bn(sparse_conv(out_b2p4, p["conv3p4s2.kernel"], None, dn[2], up[2]), "bn3", True, plans[3])
        out = self._block("block3.0", out, n81[3], plans[3])
        out = self._bn(sparse_conv(out, p["convtr5p8s2.kernel"], None, up[2], dn[2]), "bntr5", True, plans[2])
        out = self._block("block6.0", torch.cat([out, out_b2p4], 1), n81[2], plans[2])
        out = self._bn(sparse_conv(out, p["convtr6p4s2.kernel"], None, up[1], dn[1]), "bntr6", True, plans[1])
        out = self._block("block7.0", torch.cat([out, out_b1p2], 1), n81[1], plans[1])
        out = self._bn(sparse_conv(out, p["convtr7p2s2.kernel"], None, up[0], dn[0]), "bntr7", True, plans[0])
        out = self._block("block8.0", torch.cat([out, out_p1], 1), n81[0], plans[0])
        motion = sparse_conv(out, p["final.kernel"], p["final.bias"], None)  # (n0, 3)
        cur = torch.nonzero((pts[:, 4] / self.dt) == 0).flatten()  # motionnet.py:42-46 (window-major, input order inside a window)
        m = gather_rows(motion, T["inverse"].long()[cur])
        if B == 1:
            return [m]
        ncur = [int(((q[:, 4] / self.dt) == 0).sum()) for q in pts_list]
        return list(torch.split(m, ncur, 0))

    def loss(self, pts, gt_labels_cur):
        """loss_motion_encoder of models/models.py:324."""
        return mos_loss(self.forward(pts), gt_labels_cur)

    def make_reducer(self, bucket_bytes=8 << 20, overlap=True):
        """Data-parallel gradient exchange (insmos_amd/ddp.py): call .reduce() between backward() and the update."""
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

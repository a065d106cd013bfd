"""MI355X execution engine for one InsMOS window: torch is used only for device memory, streams and
(in bench/metrics) torch.distributed; all compute goes through libinsmos_hip.so (include/insmos_hip.h).

Layer order, channel widths and parameter names follow the reference modules:
  MotionNet / CustomMinkUNet   models/backbones_3d/motionnet.py:21-50, models/MinkowskiEngine/minkunet.py:139-181
  VoxelGenerate + MeanVFE      models/backbones_3d/voxel_generate.py:17-31, models/backbones_2d/mean_vfe.py:36-55
  UNetV2                       models/backbones_3d/spconv_unet.py:267-416
  HeightCompression            models/backbones_2d/height_compression.py:14-33
  BaseBEVBackbone, CenterHead  models/backbones_2d/base_bev_backbone.py:84-115, center_head.py:65-98,251-276
  post_processing              models/post_process.py:112-224
Eval-mode BatchNorm is folded into the conv taps; channel concatenations (ME.cat, torch.cat of the
UR blocks and the instance one-hots) are free: producers write into column slices of shared rows.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from . import params as P


def _np_i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _hp(a):  # host pointer of a numpy array
    return a.ctypes.data_as(ctypes.c_void_p)


def me_kernel_offsets(kernel_size, tensor_stride):
    """ME kernel-region tap order: x fastest; odd sizes centred, even sizes {0,1}; scaled by tensor stride."""
    ks, ts = list(kernel_size), list(tensor_stride)
    offs = []
    for it in range(ks[3]):
        for iz in range(ks[2]):
            for iy in range(ks[1]):
                for ix in range(ks[0]):
                    idx = (ix, iy, iz, it)
                    offs.append([((idx[d] - (ks[d] - 1) // 2) if ks[d] % 2 == 1 else idx[d]) * ts[d] for d in range(4)])
    return _np_i32(offs)


def spconv_tap_offsets(ksize):
    return [(kz, ky, kx) for kz in range(ksize[0]) for ky in range(ksize[1]) for kx in range(ksize[2])]


class ConvLayer:
    """Device-resident packed taps + folded bias of one convolution."""

    def __init__(self, lib, taps, bias, cin_pad, cout_store, device):
        K, cin_real, cout_real = taps.shape
        self.K, self.cin, self.cout = K, cin_pad, cout_store
        self.cout_real = cout_real
        nfl = lib.insmos_packed_weight_floats(K, cin_pad, cout_store)
        packed = np.empty(nfl, dtype=np.float32)
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        _lib.check(lib.insmos_pack_weights_host(_hp(taps), K, cin_real, cout_real, cin_pad, cout_store, _hp(packed)),
                   "insmos_pack_weights_host")
        ntile = (cout_store + 15) // 16
        b = np.zeros(ntile * 16, dtype=np.float32)
        if bias is not None:
            b[:cout_real] = np.asarray(bias, np.float32).reshape(-1)
        self.w = torch.from_numpy(packed).to(device)
        self.b = torch.from_numpy(b).to(device)
        self.flops_per_pair = 2 * cin_real * cout_real


def _pad4(c):
    return (c + 3) // 4 * 4


def _padc(c):
    """Channel widths insmos_sparse_conv accepts: 4, 8, or a multiple of 16 (zero-padded columns)."""
    return 4 if c <= 4 else 8 if c <= 8 else (c + 15) // 16 * 16


class _NativeCtx:
    """Owner of one insmos_ctx_create handle (destroyed with the last reference)."""

    def __init__(self, lib, handle, keepalive):
        self.lib, self.handle, self.keepalive = lib, handle, keepalive

    def __del__(self):
        try:
            if self.handle is not None and self.lib is not None:
                self.lib.insmos_ctx_destroy(self.handle)
        except Exception:  # interpreter shutdown: the library may already be gone
            pass
        self.handle = None


class NbrTable:
    """Output-stationary neighbour table + its per-16-row-group active-tap bitmasks."""

    def __init__(self, nbr, mask16):
        self.nbr, self.mask16 = nbr, mask16
        self.shape = nbr.shape

    def data_ptr(self):
        return self.nbr.data_ptr()

    def cpu(self):
        return self.nbr.cpu()

    def __ge__(self, other):
        return self.nbr >= other


MAX_WINDOWS_PER_LAUNCH = 16  # INSMOS_MAX_BATCH of csrc/common.h


class Engine:
    def __init__(self, cfg, state_dict, device="cuda:0", quirk_exact=True, max_voxels=100000, max_points=5, native=False):
        self.lib = _lib.load()
        self.native = native      # default path of forward_window (see there)
        self.prune_dead_rows = True  # MotionNet decoder layers skip rows nothing consumes
        self.tables_only = False  # motionnet() / unet() stop after the coordinate sets and kernel maps (the trainers' use)
        self.fuse_deconv_head = True  # BEV deblock + heads in one kernel (the step path can run them separately)
        self.dense_bev_kernel = os.environ.get("INSMOS_BEV_KERNEL", "1") != "0"  # LDS-tiled 3x3 kernel for the BEV backbone
        self.keep_current_points = False  # 'eval' mode: keep current_point (Ncur, 8) of the last window (motion loss)
        self.last_current_points = None
        self._ctx_box = [None]    # native context, shared with clones
        self._arena = None
        self.cfg = cfg
        self.device = torch.device(device)
        self.quirk_exact = quirk_exact
        self.max_voxels, self.max_points = max_voxels, max_points
        d, m = cfg["DATA"], cfg["MODEL"]
        self.vs = [float(v) for v in d["VOXEL_SIZE"]]
        self.range = [float(v) for v in d["POINT_CLOUD_RANGE"]]
        self.dt = float(m["DELTA_T_PREDICTION"])
        self.ncls = int(m["DENSE_HEAD"]["NUM_CLASS"])
        self.in_ch = len(m["POINT_FEATURE_ENCODING"]["src_feature_list"]) + 3
        grid = np.round((np.array(self.range[3:6]) - np.array(self.range[0:3])) / np.array(self.vs)).astype(np.int64)
        self.grid = [int(g) for g in grid]  # [nx, ny, nz]
        self.shape = {1: [self.grid[2] + 1, self.grid[1], self.grid[0]]}  # spconv_unet.py:114
        for l in (2, 3, 4):
            self.shape[l] = [(s + 2 - 3) // 2 + 1 for s in self.shape[l - 1]]
        s4 = self.shape[4]
        self.shape[5] = [(s4[0] - 3) // 2 + 1, s4[1], s4[2]]
        self.bevD, self.bevH, self.bevW = self.shape[5]
        self.nbev = int(m["MAP_TO_BEV"]["NUM_BEV_FEATURES"])
        if self.nbev != 128 * self.bevD:
            raise ValueError(f"NUM_BEV_FEATURES={self.nbev} does not match 128*{self.bevD} (height_compression.py:28)")
        pp = m["POST_PROCESSING"]
        self.score_thresh = float(pp["SCORE_THRESH"])
        self.nms_thresh = float(pp["NMS_CONFIG"]["NMS_THRESH"])
        self.pre_max = int(pp["NMS_CONFIG"]["NMS_PRE_MAXSIZE"])
        self.post_max = int(pp["NMS_CONFIG"]["NMS_POST_MAXSIZE"])
        if pp["NMS_CONFIG"]["MULTI_CLASSES_NMS"] or pp["NMS_CONFIG"]["NMS_TYPE"] != "nms_gpu" or pp["OUTPUT_RAW_SCORE"]:
            raise NotImplementedError("only the class-agnostic nms_gpu branch of post_processing is on the hot path")
        tcfg = m["DENSE_HEAD"]["TARGET_ASSIGNER_CONFIG"]
        self.out_factor = float(tcfg["OUT_SIZE_FACTOR"])
        self.tvs = [float(v) for v in tcfg["VOXEL_SIZE"]]
        self.up = int(m["BACKBONE_2D"]["UPSAMPLE_STRIDES"][0])
        if self.up != 2:
            raise NotImplementedError("BEV deconv stride 2 only (config.yaml:118)")
        self._ws = None
        self._conv_log = []
        self._conv_nin = []
        self.const_input = True  # MotionNet input features are the constant 0.5 (motionnet.py:29-32)
        self.layer_timing = None  # set to [] to record per-conv (name, K, cin, cout, n_out, ev0, ev1)
        self._load_weights(state_dict)
        self._static_tables()
        self.last_counts = {}

    # ------------------------------------------------------------------------------------------------
    def _sd(self, name):
        v = self.sd[name]
        return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)

    def _bn(self, stem):
        return (self._sd(stem + ".weight"), self._sd(stem + ".bias"), self._sd(stem + ".running_mean"),
                self._sd(stem + ".running_var"))

    def _drop_split_weights(self):
        """Unregister this engine's split-bf16 weight copies (the library keys them by the packed weights' device address,
        which the caching allocator may hand to another layer once these tensors are gone)."""
        reg = getattr(self, "_split_reg", None)
        if reg:
            for wptr in reg:
                self.lib.insmos_register_split_weights(wptr, None)
        self._split_reg, self._split_w = [], None

    def __del__(self):
        try:
            if getattr(self, "_split_owner", False):
                self._drop_split_weights()
        except Exception:
            pass

    def _load_weights(self, sd):
        if getattr(self, "_split_owner", False):
            self._drop_split_weights()   # new packed weights: the old split copies must not be found under a reused address
            self._split_owner = False
            if getattr(self, "conv_precision", 0) == 3:
                self.lib.insmos_conv_precision(0)
                self.conv_precision = 0
        self.sd = sd
        self._ctx_box = [None]  # a fresh box: clones made earlier keep (and may still run on) the old context
        L, dev, lib = {}, self.device, self.lib
        M = P.ME_PREFIX

        def me(conv, bn, cin_pad, cout_store, bias=None):
            taps = P.me_kernel_to_taps(self._sd(M + conv + ".kernel"))
            b = None
            if bn is not None:
                taps, b = P.fold_bn(taps, *self._bn(M + bn + ".bn"), 1e-5)
            elif bias is not None:
                b = bias
            return ConvLayer(lib, taps, b, cin_pad, cout_store, dev)

        for conv, bn, kv, ci, co in P.ME_CONVS:
            L[conv] = me(conv, bn, _padc(ci), co)
        for name, ci, co in P.ME_BLOCKS:
            L[name + ".conv1"] = me(name + ".conv1", name + ".norm1", _padc(ci), co)
            L[name + ".conv2"] = me(name + ".conv2", name + ".norm2", _padc(co), co)
            if ci != co:
                L[name + ".ds"] = me(name + ".downsample.0", name + ".downsample.1", _padc(ci), co)
        L["final"] = me("final", None, 8, 3, bias=self._sd(M + "final.bias"))
        # constant-input form of conv0: 0.5 * BN-folded taps (125, 8) and the folded shift
        t0, b0 = P.fold_bn(P.me_kernel_to_taps(self._sd(M + "conv0p1s1.kernel")), *self._bn(M + "bn0.bn"), 1e-5)
        self.w0_const = torch.from_numpy(np.ascontiguousarray(np.float32(0.5) * t0[:, 0, :])).to(dev)
        self.b0_const = torch.from_numpy(np.ascontiguousarray(b0)).to(dev)

        U = P.UNET_PREFIX
        for conv, bn, ks, ci, co in P.unet_convs(self.in_ch, self.ncls):
            taps = P.spconv_weight_to_taps(self._sd(U + conv + ".weight"))
            b = None
            if bn:
                taps, b = P.fold_bn(taps, *self._bn(U + bn), 1e-3)
            L[conv] = ConvLayer(lib, taps, b, _padc(ci), co, dev)
        B = U + "bev_backbone."
        taps, b = P.fold_bn(P.conv2d_weight_to_taps(self._sd(B + "blocks.0.1.weight")), *self._bn(B + "blocks.0.2"), 1e-3)
        L["bev0"] = ConvLayer(lib, taps, b, taps.shape[1], taps.shape[2], dev)
        self.n_bev_layers = int(self.cfg["MODEL"]["BACKBONE_2D"]["LAYER_NUMS"][0])
        for k in range(self.n_bev_layers):
            taps, b = P.fold_bn(P.conv2d_weight_to_taps(self._sd(B + f"blocks.0.{4 + 3 * k}.weight")),
                                *self._bn(B + f"blocks.0.{5 + 3 * k}"), 1e-3)
            L[f"bev{k + 1}"] = ConvLayer(lib, taps, b, taps.shape[1], taps.shape[2], dev)
        # ConvTranspose2d(k=2,s=2) == a 1x1 conv with 4*Cout output channels laid out [ky][kx][co]
        tt = P.convT2d_weight_to_taps(self._sd(B + "deblocks.0.0.weight"))  # (4, Cin, Cout)
        tt, b = P.fold_bn(tt, *self._bn(B + "deblocks.0.1"), 1e-3)
        K4, ci, co = tt.shape
        wide = np.ascontiguousarray(tt.transpose(1, 0, 2).reshape(1, ci, K4 * co))
        L["deconv"] = ConvLayer(lib, wide, np.tile(b, K4), ci, K4 * co, dev)
        self.up_ch = co
        Hd = U + "center_head."
        wc = self._sd(Hd + "conv_cls.weight").reshape(self.ncls, -1)
        wb = self._sd(Hd + "conv_box.weight").reshape(8, -1)
        taps = np.concatenate([wc, wb], 0).T[None].astype(np.float32)  # (1, 256, ncls+8)
        bias = np.concatenate([self._sd(Hd + "conv_cls.bias"), self._sd(Hd + "conv_box.bias")])
        self.head_ld = _pad4(self.ncls + 8)
        L["head"] = ConvLayer(lib, taps, bias, co, self.head_ld, dev)
        wl = self._sd(U + "mos_seg_layer.weight")  # (3,16)
        L["mos_seg"] = ConvLayer(lib, np.ascontiguousarray(wl.T[None]), self._sd(U + "mos_seg_layer.bias"), 16, 3, dev)
        for k, v in L.items():
            v.name = k
        self.L = L

    def _static_tables(self):
        dev = self.device
        self.nbr_bev = torch.empty((9, self.bevH * self.bevW), dtype=torch.int32, device=dev)
        _lib.check(self.lib.insmos_dense_nbr2d(self.bevH, self.bevW, self.nbr_bev.data_ptr(), self._stream()),
                   "insmos_dense_nbr2d")
        self.off125 = me_kernel_offsets([5, 5, 5, 1], [1, 1, 1, 1])
        self.off81 = [me_kernel_offsets([3, 3, 3, 3], [1 << l, 1 << l, 1 << l, 1]) for l in range(4)]
        self.off8 = [me_kernel_offsets([2, 2, 2, 1], [1 << l, 1 << l, 1 << l, 1]) for l in range(3)]
        self.ones4 = _np_i32([1, 1, 1, 1])

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=self.device)
        return self._ws

    def _empty(self, shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------------------------------------
    def conv(self, layer, x, ld_in, nbr, n_out, out, ld_out, col_out=0, col_in=0, res=None, ld_res=0, col_res=0,
             res_mode=0, relu_pre=0, relu_post=0, n_in=None, row0=0):
        """out[row0:, col_out:col_out+cout] = epilogue(conv(x[:, col_in:col_in+cin])) (rows below row0 are not computed)."""
        if n_out == 0:
            return
        K = layer.K
        mask = None
        if nbr is not None:
            assert nbr.shape[0] == K and nbr.shape[1] == n_out, (nbr.shape, K, n_out)
            if isinstance(nbr, NbrTable):
                mask = nbr.mask16
        if self.layer_timing is not None:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record(torch.cuda.current_stream(self.device))
        row0 = int(row0) & ~15
        rc = self.lib.insmos_sparse_conv_rows(
            x.data_ptr() + 4 * col_in, x.shape[0] if n_in is None else n_in, ld_in, layer.cin, nbr.data_ptr() if nbr is not None else None,
            mask.data_ptr() if mask is not None else None, K, n_out, row0,
            layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr() + 4 * col_out, ld_out, layer.cout,
            (res.data_ptr() + 4 * col_res) if res is not None else None, ld_res, res_mode, relu_pre, relu_post,
            self._stream())
        _lib.check(rc, "insmos_sparse_conv_rows")
        self._conv_log.append((nbr, n_out, layer, row0))
        self._conv_nin.append(int(x.shape[0] if n_in is None else n_in))
        if self.layer_timing is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.device))
            self.layer_timing.append((layer.name, K, layer.cin, layer.cout, n_out, self._t0, e1))

    def set_conv_precision(self, mode):
        """EXPERIMENT / training opt-in, process-wide, never the default (include/insmos_hip.h: insmos_conv_precision).
        0 = exact fp32; 1 = bf16 operands, fp32 accumulate; 3 = split-bf16 x 3 on every layer with Cin % 16 == 0 and a
        neighbour table (their weights are split once, here, and registered)."""
        mode = int(mode)
        if mode == 3 and not getattr(self, "_split_w", None):
            self._split_w, self._split_reg, self._split_owner = {}, [], True
            for name, l in self.L.items():
                if l.cin % 16 == 0 and l.K > 1:
                    buf = torch.empty_like(l.w)
                    _lib.check(self.lib.insmos_split_weights_bf16(l.w.data_ptr(), l.w.numel(), buf.data_ptr(), self._stream()),
                               "insmos_split_weights_bf16")
                    _lib.check(self.lib.insmos_register_split_weights(l.w.data_ptr(), buf.data_ptr()), "insmos_register_split_weights")
                    self._split_w[name] = buf
                    self._split_reg.append(l.w.data_ptr())
            torch.cuda.synchronize(self.device)
        _lib.check(self.lib.insmos_conv_precision(mode), "insmos_conv_precision")
        self.conv_precision = mode

    def bev_conv(self, layer, x, ld_in, out, ld_out):
        """One 3x3 layer of the dense BEV backbone (+ folded BN + ReLU): the LDS-tiled kernel (csrc/bev.hip) when the layer
        has a shape it is built for, else the generic kernel over the dense 9-tap table."""
        nsite = self.bevH * self.bevW
        if self.dense_bev_kernel and layer.cin % 16 == 0 and layer.cout in (64, 128):
            _lib.check(self.lib.insmos_bev_conv3x3(x.data_ptr(), 1, self.bevH, self.bevW, ld_in, layer.cin, layer.w.data_ptr(),
                                                   layer.b.data_ptr(), out.data_ptr(), ld_out, layer.cout, 1, self._stream()),
                       "insmos_bev_conv3x3")
            self._conv_log.append((self.nbr_bev, nsite, layer, 0))
            self._conv_nin.append(nsite)
            if getattr(self, "_bev_dist", None) is not None and int(layer.name[3:]) < int(os.environ.get("INSMOS_BEV_SKIP_LAYERS", "99")):
                # accounting only (bench.py): the pairs the native runner's constant-region skipping executes for this layer
                cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
                _lib.check(self.lib.insmos_bev_skip_executed_pairs(self._bev_dist.data_ptr(), 1, self.bevH, self.bevW, int(layer.name[3:]),
                                                                   cnt.data_ptr(), self._stream()), "insmos_bev_skip_executed_pairs")
                self._bev_exec_pairs[layer.name] = int(cnt.item())
        else:
            self.conv(layer, x, ld_in, self.nbr_bev, nsite, out, ld_out, relu_post=1)

    def build_nbr(self, out_coords, n_out, in_keys, in_perm, n_in, mode, shape, delta, mul=None, div=None):
        K = len(delta)
        nbr = self._empty((K, n_out), torch.int32)
        mask = self._empty(((n_out + 15) // 16, 4), torch.int32)
        if n_out == 0:
            return NbrTable(nbr, mask)
        delta = _np_i32(delta)
        mul = _np_i32(mul if mul is not None else [1, 1, 1, 1])
        div = _np_i32(div if div is not None else [1, 1, 1, 1])
        shp = _np_i32(shape) if shape is not None else None
        rc = self.lib.insmos_build_nbr(out_coords.data_ptr(), n_out, in_keys.data_ptr(),
                                       in_perm.data_ptr() if in_perm is not None else None, n_in, mode,
                                       _hp(shp) if shp is not None else None, _hp(delta), K, _hp(mul), _hp(div),
                                       nbr.data_ptr(), mask.data_ptr(), self._stream())
        _lib.check(rc, "insmos_build_nbr")
        return NbrTable(nbr, mask)

    # ------------------------------------------------------------------------------------------------
    def motionnet(self, pts, win_sizes=None):
        """pts (N, ld>=5) fp32 device [x,y,z,r,t] -> current_point (Ncur, 8) [x,y,z,r,m0,m1,m2,0].
        win_sizes: pts holds B = len(win_sizes) windows back to back; the window index is
        folded into the time coordinate (t' = t * B + b), the searched table is built on time offsets scaled by B,
        everything else runs unchanged on B x the rows -> current points of all windows in input order (window-major).
        (The product path for batches is the native runner, insmos_forward_windows; this is its inspectable 4D half.)"""
        lib, st = self.lib, self._stream()
        B = len(win_sizes) if win_sizes is not None else 1
        batched = B > 1
        N, ld = pts.shape[0], pts.stride(0)
        if batched:
            assert sum(win_sizes) == N
            starts = np.concatenate([[0], np.cumsum(win_sizes)[:-1]]).astype(np.int64)
            win_ptr = (ctypes.c_void_p * B)(*[pts.data_ptr() + 4 * ld * int(o) for o in starts])
            win_n = (ctypes.c_int64 * B)(*[int(v) for v in win_sizes])
        ws = self._workspace(lib.insmos_quantize4d_ws_bytes(N))
        keys0 = self._empty((N,), torch.int64)
        coords0 = self._empty((N, 4), torch.int32)
        inverse = self._empty((N,), torch.int32)
        cur_index = self._empty((N,), torch.int32)
        counts = self._empty((8 + B,), torch.int32)
        quant = np.array([self.vs[0], self.vs[0], self.vs[0], self.dt], dtype=np.float32)
        for compact in (1, 0):  # 40-bit sort keys first; the full-width sort only for windows wider than +-2048 voxels
            if batched:
                _lib.check(lib.insmos_quantize4d_windows(win_ptr, win_n, B, ld, _hp(quant), keys0.data_ptr(),
                                                         coords0.data_ptr(), inverse.data_ptr(), cur_index.data_ptr(),
                                                         counts.data_ptr(), ws.data_ptr(), ws.numel(), compact, st),
                           "insmos_quantize4d_windows")
            else:
                _lib.check(lib.insmos_quantize4d_ex(pts.data_ptr(), N, ld, _hp(quant), keys0.data_ptr(), coords0.data_ptr(),
                                                    inverse.data_ptr(), cur_index.data_ptr(), counts.data_ptr(), ws.data_ptr(),
                                                    ws.numel(), compact, st), "insmos_quantize4d_ex")
            c = counts.cpu().numpy()
            if int(c[3]) == 0:
                break
        n0, ncur = int(c[0]), int(c[1])
        if int(c[2]) != 0:
            raise ValueError(f"{int(c[2])} points fall outside the +-32768-voxel key window")
        keys = [keys0[:n0]]
        coords = [coords0[:n0]]
        n = [n0]
        parent, child_start, child_mask = [], [], []  # index l: maps level l -> level l+1
        for l in (1, 2, 3):
            kk = self._empty((n[-1],), torch.int64)
            cc = self._empty((n[-1], 4), torch.int32)
            par = self._empty((n[-1],), torch.int32)
            cst = self._empty((n[-1],), torch.int32)
            cmk = self._empty((n[-1],), torch.int32)
            ws = self._workspace(lib.insmos_level_down4d_ws_bytes(n[-1]))
            # level l from level l-1 (its keys already have the lower bits cleared)
            _lib.check(lib.insmos_level_down4d(keys[-1].data_ptr(), n[-1], l, kk.data_ptr(), cc.data_ptr(),
                                               par.data_ptr(), cst.data_ptr(), cmk.data_ptr(), counts.data_ptr(),
                                               ws.data_ptr(), ws.numel(), st), "insmos_level_down4d")
            nl = int(counts[0].item())
            keys.append(kk[:nl])
            coords.append(cc[:nl])
            n.append(nl)
            parent.append(par)
            child_start.append(cst)
            child_mask.append(cmk)
        self.last_counts["me_voxels"] = list(n)
        self.last_counts["n_cur"] = ncur
        # Dead-row elimination: only the current scan's voxels of the final feature map are read
        # (motionnet.py:38-48), rows are ordered by time first, and a 3^4 convolution widens the needed time range by
        # one scan per layer.  starts[l][d] = first level-l row with t >= t_last - d; row_from(l, d) = where a layer whose
        # output is needed `d` scans back starts computing.
        if self.prune_dead_rows:
            starts_d = self._empty((4, 16), torch.int32)
            for l in range(4):
                if batched:
                    _lib.check(lib.insmos_tslice_starts_batched(keys[l].data_ptr(), n[l], 16, B, starts_d[l].data_ptr(), st),
                               "insmos_tslice_starts_batched")
                else:
                    _lib.check(lib.insmos_tslice_starts(keys[l].data_ptr(), n[l], 16, starts_d[l].data_ptr(), st),
                               "insmos_tslice_starts")
            starts = starts_d.cpu().numpy()
        else:
            starts = np.zeros((4, 16), np.int32)

        def row_from(l, d):
            return int(starts[l][d]) if d < 16 else 0
        self.last_counts["me_row_starts"] = [[row_from(l, d) for d in range(10)] for l in range(4)]

        def table(K, n_out):
            return (self._empty((K, n_out), torch.int32), self._empty(((n_out + 15) // 16, 4), torch.int32))

        def from_coarse(l, offs, coarse):
            """level-l table for taps `offs` from the level-(l+1) 81-tap table: no key search."""
            nb, mk = table(len(offs), n[l])
            _lib.check(lib.insmos_nbr_from_coarse(coords[l].data_ptr(), n[l], parent[l].data_ptr(), l,
                                                  coarse.nbr.data_ptr(), n[l + 1], child_start[l].data_ptr(),
                                                  child_mask[l].data_ptr(), _hp(offs), len(offs), nb.data_ptr(),
                                                  mk.data_ptr(), st), "insmos_nbr_from_coarse")
            return NbrTable(nb, mk)

        # only the coarsest level is searched; every finer table is derived through the Morton hierarchy
        off_c = self.off81[3]
        if batched:  # the only arithmetic on t in the whole branch: time offsets of the searched table move by B
            off_c = off_c.copy()
            off_c[:, 3] *= B
        nbr81 = [None, None, None, self.build_nbr(coords[3], n[3], keys[3], None, n[3], 0, None, off_c)]
        for l in (2, 1, 0):
            nb, mk = table(81, n[l])
            # the level-0 table is read by block8 only: rows of the last two scans (dead-row elimination, see above)
            _lib.check(lib.insmos_nbr81_from_coarse_rows(coords[l].data_ptr(), n[l], row_from(0, 1) if l == 0 else 0,
                                                         parent[l].data_ptr(), l, nbr81[l + 1].nbr.data_ptr(), n[l + 1],
                                                         child_start[l].data_ptr(), child_mask[l].data_ptr(), nb.data_ptr(),
                                                         mk.data_ptr(), st), "insmos_nbr81_from_coarse_rows")
            nbr81[l] = NbrTable(nb, mk)
        # the 125-tap table is only materialised when the first layer's input is not constant (generic path)
        nbr125 = from_coarse(0, self.off125, nbr81[1]) if not self.const_input else None
        dn, up = [], []
        for l in range(3):
            d_nb, d_mk = table(8, n[l + 1])
            u_nb, u_mk = table(8, n[l])
            _lib.check(lib.insmos_nbr_down_up(coords[l].data_ptr(), n[l], parent[l].data_ptr(), l, n[l + 1],
                                              child_start[l].data_ptr(), child_mask[l].data_ptr(), d_nb.data_ptr(),
                                              d_mk.data_ptr(), u_nb.data_ptr(), u_mk.data_ptr(), st),
                       "insmos_nbr_down_up")
            dn.append(NbrTable(d_nb, d_mk))
            up.append(NbrTable(u_nb, u_mk))
        self._me_tables = dict(nbr125=nbr125, nbr81=nbr81, dn=dn, up=up, coords=coords, keys=keys, inverse=inverse,
                               parent=parent, child_start=child_start, child_mask=child_mask)
        if self.tables_only:
            return None

        L, E = self.L, self._empty
        x_in = None
        if not self.const_input:
            x_in = E((n[0], 4))
            x_in.zero_()
            _lib.check(lib.insmos_fill_cols(x_in.data_ptr(), n[0], 4, 0, 1, 0.5, st), "insmos_fill_cols")
        cat8 = E((n[0], 16))  # [convtr7 (8) | out_p1 (8)]
        cat7 = E((n[1], 32))  # [convtr6 (16) | out_b1p2 (8) | zero pad (8)]
        _lib.check(lib.insmos_fill_cols(cat7.data_ptr(), n[1], 32, 24, 8, 0.0, st), "insmos_fill_cols")
        cat6 = E((n[2], 48))  # [convtr5 (32) | out_b2p4 (16)]
        if self.const_input:
            # motionnet.py:29-32: every point carries the feature 0.5 -> conv0 needs no table and no gathers
            cubes = torch.empty(max(n[1], 1) * 12, dtype=torch.int32, device=self.device)   # occupancy cubes, 48 B per coarse voxel
            _lib.check(lib.insmos_const_conv125_cubes(coords[0].data_ptr(), n[0], parent[0].data_ptr(), 0,
                                                      nbr81[1].nbr.data_ptr(), None, n[1], child_start[0].data_ptr(),
                                                      child_mask[0].data_ptr(), self.w0_const.data_ptr(),
                                                      self.b0_const.data_ptr(), cat8.data_ptr() + 4 * 8, 16, 1,
                                                      cubes.data_ptr(), st),
                       "insmos_const_conv125_cubes")
            self._conv_log.append((None, n[0], L["conv0p1s1"], 0))
            self._conv_nin.append(0)   # constant input: nothing is read
        else:
            self.conv(L["conv0p1s1"], x_in, 4, nbr125, n[0], cat8, 16, col_out=8, relu_post=1)
        x1 = E((n[1], 8))
        self.conv(L["conv1p1s2"], cat8, 16, dn[0], n[1], x1, 8, col_in=8, relu_post=1)

        def block(name, x, ld_x, col_x, nb, nn, cout, out, ld_out, col_out, lvl=0, depth=99):
            """BasicBlock whose output is needed `depth` scans back: conv2 / downsample run on those rows, conv1 one
            scan further back (conv2 reads it through a 3^4 window)."""
            r2, r1 = row_from(lvl, depth), row_from(lvl, depth + 1)
            t = E((nn, cout))
            self.conv(L[name + ".conv1"], x, ld_x, nb, nn, t, cout, col_in=col_x, relu_post=1, row0=r1)
            if (name + ".ds") in L:
                r = E((nn, cout))
                self.conv(L[name + ".ds"], x, ld_x, None, nn, r, cout, col_in=col_x, row0=r2)
                self.conv(L[name + ".conv2"], t, cout, nb, nn, out, ld_out, col_out=col_out, res=r, ld_res=cout,
                          res_mode=1, relu_post=1, row0=r2)
            else:
                self.conv(L[name + ".conv2"], t, cout, nb, nn, out, ld_out, col_out=col_out, res=x, ld_res=ld_x,
                          col_res=col_x, res_mode=1, relu_post=1, row0=r2)

        block("block1.0", x1, 8, 0, nbr81[1], n[1], 8, cat7, 32, 16)
        x2 = E((n[2], 8))
        self.conv(L["conv2p2s2"], cat7, 32, dn[1], n[2], x2, 8, col_in=16, relu_post=1)
        block("block2.0", x2, 8, 0, nbr81[2], n[2], 16, cat6, 48, 32)
        x3 = E((n[3], 16))
        self.conv(L["conv3p4s2"], cat6, 48, dn[2], n[3], x3, 16, col_in=32, relu_post=1)
        b3 = E((n[3], 32))
        # decoder side: needed time depth per layer = 0 at `final`, +1 per 3^4 conv on the way back (block = 2 convs)
        block("block3.0", x3, 16, 0, nbr81[3], n[3], 32, b3, 32, 0, lvl=3, depth=6)
        self.conv(L["convtr5p8s2"], b3, 32, up[2], n[2], cat6, 48, col_out=0, relu_post=1, row0=row_from(2, 6))
        b6 = E((n[2], 32))
        block("block6.0", cat6, 48, 0, nbr81[2], n[2], 32, b6, 32, 0, lvl=2, depth=4)
        self.conv(L["convtr6p4s2"], b6, 32, up[1], n[1], cat7, 32, col_out=0, relu_post=1, row0=row_from(1, 4))
        b7 = E((n[1], 16))
        block("block7.0", cat7, 32, 0, nbr81[1], n[1], 16, b7, 16, 0, lvl=1, depth=2)
        self.conv(L["convtr7p2s2"], b7, 16, up[0], n[0], cat8, 16, col_out=0, relu_post=1, row0=row_from(0, 2))
        b8 = E((n[0], 8))
        block("block8.0", cat8, 16, 0, nbr81[0], n[0], 8, b8, 8, 0, lvl=0, depth=0)
        motion = E((n[0], 4))
        self.conv(L["final"], b8, 8, None, n[0], motion, 4, row0=row_from(0, 0))
        cur = E((ncur, 8))
        _lib.check(lib.insmos_build_current_points(pts.data_ptr(), ld, motion.data_ptr(), 4, inverse.data_ptr(),
                                                   cur_index.data_ptr(), ncur, cur.data_ptr(), 8, st),
                   "insmos_build_current_points")
        self._me_debug = dict(b3=b3, b8=b8, motion=motion, cat8=cat8)
        if self.keep_current_points:
            self.last_current_points = cur
        return cur

    # ------------------------------------------------------------------------------------------------
    def motionnet_windows(self, pts_list):
        """The 4D half of a launch set on the step path: MotionNet of several windows in ONE set of launches ->
        list of current_point tensors, each bit-identical to motionnet() of that window alone."""
        B = len(pts_list)
        if B == 1:
            return [self.motionnet(pts_list[0])]
        pts = torch.cat([p[:, :5] for p in pts_list], 0).contiguous()
        cur = self.motionnet(pts, [int(p.shape[0]) for p in pts_list])
        ncur = [int((p[:, 4] == 0).sum()) for p in pts_list]
        assert sum(ncur) == int(cur.shape[0]), (ncur, cur.shape)
        return list(torch.split(cur, ncur, 0))

    # ------------------------------------------------------------------------------------------------
    def detect(self, head, up):
        """CenterHead decode + post_processing (center_head.py:251-276, post_process.py:112-224) on a head map
        (n_cells, head_ld) = [cls | box] rows: up=2 -> the deconv's sub-site row order, up=1 -> plain [row][col] order.
        -> (pred_boxes (post_max, 7), scores, labels int64, count (device), candidate count (device), candidates...)."""
        lib, st, E = self.lib, self._stream(), self._empty
        H2, W2 = 2 * self.bevH, 2 * self.bevW
        ncell = H2 * W2
        cb, cs = E((self.pre_max, 7)), E((self.pre_max,))
        cl, cc = E((self.pre_max,), torch.int32), E((self.pre_max,), torch.int32)
        cnt_c = E((4,), torch.int32)
        w = self._workspace(lib.insmos_center_decode_select_ws_bytes(ncell))
        _lib.check(lib.insmos_center_decode_select(head.data_ptr(), self.head_ld, self.ncls, H2, W2, up, self.out_factor,
                                                   self.tvs[0], self.tvs[1], self.range[0], self.range[1],
                                                   self.score_thresh, self.pre_max, cb.data_ptr(), cs.data_ptr(),
                                                   cl.data_ptr(), cc.data_ptr(), cnt_c.data_ptr(), w.data_ptr(), w.numel(),
                                                   st), "insmos_center_decode_select")
        keep = E((self.post_max,), torch.int32)
        cnt_k = E((4,), torch.int32)
        w = self._workspace(lib.insmos_nms_ws_bytes(self.pre_max))
        _lib.check(lib.insmos_nms_rotated_bev(cb.data_ptr(), cnt_c.data_ptr(), self.pre_max, self.nms_thresh,
                                              self.post_max, keep.data_ptr(), cnt_k.data_ptr(), w.data_ptr(), w.numel(),
                                              st), "insmos_nms_rotated_bev")
        pb, psc = E((self.post_max, 7)), E((self.post_max,))
        pl = E((self.post_max,), torch.int64)
        _lib.check(lib.insmos_gather_preds(cb.data_ptr(), cs.data_ptr(), cl.data_ptr(), keep.data_ptr(),
                                           cnt_k.data_ptr(), self.post_max, pb.data_ptr(), psc.data_ptr(), pl.data_ptr(),
                                           st), "insmos_gather_preds")
        return pb, psc, pl, cnt_k, cnt_c, cb, cs, cl, cc, keep

    def instance_onehot(self, pb, pl, cnt_k, level_coords, n, mult, out, ld, col, scratch):
        """Array_Index.find_features_by_bbox_with_yaw at one decoder level (spconv_unet.py:322-345): boxes scaled to that
        level's voxel units (stride 8 / mult), one-hot class columns written at out[:, col:col+ncls]."""
        lo = np.array(self.range[0:3], dtype=np.float32)
        vsz = np.array(self.vs, dtype=np.float32)
        _lib.check(self.lib.insmos_boxes_to_onehot(pb.data_ptr(), pl.data_ptr(), cnt_k.data_ptr(), self.post_max, _hp(lo),
                                                   _hp(vsz), 8.0, float(mult), level_coords.data_ptr(), n, self.ncls,
                                                   16, 1 if self.quirk_exact else 0, out.data_ptr() + 4 * col, ld,
                                                   scratch.data_ptr(), self._stream()), "insmos_boxes_to_onehot")

    # ------------------------------------------------------------------------------------------------
    def unet(self, cur):
        """cur (Ncur, 8) -> (point logits (Ncur,3), pred dict of device tensors)."""
        lib, st, E, L = self.lib, self._stream(), self._empty, self.L
        ncur = cur.shape[0]
        ncls = self.ncls
        counts = E((4,), torch.int32)
        V_cap = self.max_voxels
        feat = E((V_cap, 8))
        coords1 = E((V_cap, 4), torch.int32)
        num_points = E((V_cap,), torch.int32)
        pcid = E((max(ncur, 1),), torch.int64)
        ukeys = E((max(ncur, 1),), torch.int64)
        uperm = E((max(ncur, 1),), torch.int32)
        if ncur == 0:
            raise ValueError("window has no current-scan points (t == 0)")
        ws = self._workspace(lib.insmos_voxelize_mean_ws_bytes(ncur))
        rng = np.array(self.range, dtype=np.float32)
        vsz = np.array(self.vs, dtype=np.float32)
        _lib.check(lib.insmos_voxelize_mean(cur.data_ptr(), ncur, 8, self.in_ch, _hp(rng), _hp(vsz), self.max_voxels,
                                            self.max_points, feat.data_ptr(), 8, coords1.data_ptr(),
                                            num_points.data_ptr(), pcid.data_ptr(), ukeys.data_ptr(), uperm.data_ptr(),
                                            counts.data_ptr(), ws.data_ptr(), ws.numel(), st), "insmos_voxelize_mean")
        c = counts.cpu().numpy()
        V, S = int(c[0]), int(c[1])
        nv = {1: V}
        coords = {1: coords1[:V]}
        keys = {1: ukeys[:S]}
        perm = {1: uperm[:S]}
        nkeys = {1: S}
        k333, s222, p111 = _np_i32([3, 3, 3]), _np_i32([2, 2, 2]), _np_i32([1, 1, 1])

        def down_coords(lvl_in, ks, stv, pd, oshape):
            n_in = nv[lvl_in]
            K = int(np.prod(ks))
            cells = int(np.prod(oshape))
            cap = max(min(n_in * K, cells), 1)
            ok = E((cap,), torch.int64)
            oc = E((cap, 4), torch.int32)
            if n_in == 0:
                return ok[:0], oc[:0], 0
            w = self._workspace(lib.insmos_down_coords3d_ws_bytes(_hp(_np_i32(oshape))))
            _lib.check(lib.insmos_down_coords3d(coords[lvl_in].data_ptr(), n_in, _hp(ks), _hp(stv), _hp(pd),
                                                _hp(_np_i32(oshape)), ok.data_ptr(), oc.data_ptr(), counts.data_ptr(),
                                                w.data_ptr(), w.numel(), st), "insmos_down_coords3d")
            no = int(counts[0].item())
            return ok[:no], oc[:no], no

        for l in (2, 3, 4):
            keys[l], coords[l], nv[l] = down_coords(l - 1, k333, s222, p111, self.shape[l])
            perm[l], nkeys[l] = None, nv[l]
        k311, s211, p000 = _np_i32([3, 1, 1]), _np_i32([2, 1, 1]), _np_i32([0, 0, 0])
        keys[5], coords[5], nv[5] = down_coords(4, k311, s211, p000, self.shape[5])
        perm[5], nkeys[5] = None, nv[5]
        self.last_counts["unet_voxels"] = [nv[l] for l in (1, 2, 3, 4, 5)]

        t27 = spconv_tap_offsets((3, 3, 3))
        t3 = spconv_tap_offsets((3, 1, 1))
        d_subm = [[0, kz - 1, ky - 1, kx - 1] for kz, ky, kx in t27]
        d_down = [[0, kz - 1, ky - 1, kx - 1] for kz, ky, kx in t27]  # i = 2*o - 1 + k
        d_inv = [[0, 1 - kz, 1 - ky, 1 - kx] for kz, ky, kx in t27]  # o = (i + 1 - k) / 2
        d_down5 = [[0, kz, 0, 0] for kz, ky, kx in t3]  # i = (2*oz + kz, oy, ox)
        d_inv5 = [[0, -kz, 0, 0] for kz, ky, kx in t3]
        subm = {l: self.build_nbr(coords[l], nv[l], keys[l], perm[l], nkeys[l], 1, self.shape[l], d_subm)
                for l in (1, 2, 3, 4)}
        down = {l: self.build_nbr(coords[l], nv[l], keys[l - 1], perm[l - 1], nkeys[l - 1], 1, self.shape[l - 1], d_down,
                                  mul=[1, 2, 2, 2]) for l in (2, 3, 4)}
        inv = {l: self.build_nbr(coords[l - 1], nv[l - 1], keys[l], perm[l], nkeys[l], 1, self.shape[l], d_inv,
                                 div=[1, 2, 2, 2]) for l in (2, 3, 4)}
        down5 = self.build_nbr(coords[5], nv[5], keys[4], None, nkeys[4], 1, self.shape[4], d_down5, mul=[1, 2, 1, 1])
        inv5 = self.build_nbr(coords[4], nv[4], keys[5], None, nkeys[5], 1, self.shape[5], d_inv5, div=[1, 2, 1, 1])
        self._un_tables = dict(subm=subm, down=down, inv=inv, down5=down5, inv5=inv5, coords=coords, pcid=pcid,
                               feat=feat[:V], num_points=num_points[:V])
        if self.tables_only:
            return None

        # ---- encoder (spconv_unet.py:297-306)
        x0 = E((V, 16))
        self.conv(L["conv_input.0"], feat, 8, subm[1], V, x0, 16, relu_post=1)
        xc = {1: E((V, 16))}
        self.conv(L["conv1.0.0"], x0, 16, subm[1], V, xc[1], 16, relu_post=1)
        for l, C in ((2, 32), (3, 64), (4, 128)):
            a, b = E((nv[l], C)), E((nv[l], C))
            xc[l] = E((nv[l], C))
            self.conv(L[f"conv{l}.0.0"], xc[l - 1], C // 2, down[l], nv[l], a, C, relu_post=1)
            self.conv(L[f"conv{l}.1.0"], a, C, subm[l], nv[l], b, C, relu_post=1)
            self.conv(L[f"conv{l}.2.0"], b, C, subm[l], nv[l], xc[l], C, relu_post=1)
        enc = E((nv[5], 128))
        self.conv(L["conv_out.0"], xc[4], 128, down5, nv[5], enc, 128, relu_post=1)

        # ---- BEV detection head in NHWC (height_compression.py:24-31, base_bev_backbone.py:84-115)
        nsite = self.bevH * self.bevW
        bev = E((nsite, self.nbev))
        _lib.check(lib.insmos_sparse_to_bev(enc.data_ptr(), 128, 128, coords[5].data_ptr(), nv[5], self.bevD, self.bevH,
                                            self.bevW, bev.data_ptr(), st), "insmos_sparse_to_bev")
        nf = L["bev0"].cout
        fa, fb = E((nsite, nf)), E((nsite, nf))
        # (the step path computes every site -- it is the cross-check of the native runner's constant-region skipping, csrc/bev.hip;
        #  with bev_skip_accounting it also COUNTS what the runner executes, for the roofline numerator of bench.py)
        self._bev_dist, self._bev_exec_pairs = None, {}
        if getattr(self, "bev_skip_accounting", False):
            self._bev_dist = torch.empty(nsite, dtype=torch.uint8, device=self.device)
            wsd = torch.empty(int(lib.insmos_bev_distance_map_ws_bytes(1, self.bevH, self.bevW)), dtype=torch.uint8, device=self.device)
            _lib.check(lib.insmos_bev_distance_map(coords[5].data_ptr(), nv[5], 1, self.bevH, self.bevW, self.n_bev_layers + 1,
                                                   self._bev_dist.data_ptr(), wsd.data_ptr(), wsd.numel(), st), "insmos_bev_distance_map")
        self.bev_conv(L["bev0"], bev, self.nbev, fa, nf)
        for k in range(self.n_bev_layers):
            self.bev_conv(L[f"bev{k + 1}"], fa, nf, fb, nf)
            fa, fb = fb, fa
        upc = self.up_ch
        ncell = 4 * nsite
        head = E((ncell, self.head_ld))
        upf = None
        if self.fuse_deconv_head and upc == 256 and nf % 16 == 0 and self.head_ld <= 16:
            # the 2x2 deconv output (rows [y][x], columns [ky][kx][co] == (4*nsite, upc) sub-site rows) is read by the
            # heads only: one kernel keeps it in the MFMA accumulators (bit-identical to the two launches below)
            _lib.check(lib.insmos_deconv_head(fa.data_ptr(), nsite, nf, nf, L["deconv"].w.data_ptr(), L["deconv"].b.data_ptr(),
                                              upc, L["head"].w.data_ptr(), L["head"].b.data_ptr(), self.head_ld,
                                              head.data_ptr(), self.head_ld, st), "insmos_deconv_head")
            self._conv_log.append((None, nsite, L["deconv"], 0))
            self._conv_log.append((None, ncell, L["head"], 0))
            self._conv_nin += [nsite, 0]  # fused: the deconv output feeds the heads from the accumulators
            if (self._bev_dist is not None and os.environ.get("INSMOS_DECONV_SKIP", "1") != "0"
                    and int(os.environ.get("INSMOS_BEV_SKIP_LAYERS", "99")) > self.n_bev_layers):
                # accounting only (bench.py): the sites the native runner's insmos_deconv_head_skip computes behind the skipping stack
                cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
                _lib.check(lib.insmos_deconv_head_skip_active_sites(self._bev_dist.data_ptr(), nsite, self.bevH, self.bevW,
                                                                    self.n_bev_layers + 1, cnt.data_ptr(), st),
                           "insmos_deconv_head_skip_active_sites")
                self._bev_exec_pairs["deconv"] = int(cnt.item())
                self._bev_exec_pairs["head"] = 4 * int(cnt.item())
        else:
            upf = E((nsite, 4 * upc))
            self.conv(L["deconv"], fa, nf, None, nsite, upf, 4 * upc, relu_post=1)
            self.conv(L["head"], upf, upc, None, ncell, head, self.head_ld, n_in=ncell)  # upf viewed as (4*nsite, upc)
        pb, psc, pl, cnt_k, cnt_c, cb, cs, cl, cc, keep = self.detect(head, up=2)
        self._head_debug = dict(head=head, cand_boxes=cb, cand_scores=cs, cand_labels=cl, cand_cell=cc, n_cand=cnt_c,
                                keep=keep, spatial_features_2d=upf, bev=bev)

        # ---- upsample fusion (spconv_unet.py:319-402)
        scratch = E((int(lib.insmos_boxes_to_onehot_scratch_ints(self.post_max, max(nv.values()))),), torch.int32)

        def onehot(level, mult, out, ld, col):
            self.instance_onehot(pb, pl, cnt_k, coords[level], nv[level], mult, out, ld, col, scratch)

        def ur_block(lvl, C, x_lat, ld_lat, catm):
            """UR_block_forward up to (not including) conv_inv; catm[:, 0:C] already holds x_bottom."""
            t = E((nv[lvl], C))
            self.conv(L[f"conv_up_t{lvl}.conv1"], x_lat, ld_lat, subm[lvl], nv[lvl], t, C, relu_post=1)
            self.conv(L[f"conv_up_t{lvl}.conv2"], t, C, subm[lvl], nv[lvl], catm, 2 * C, col_out=C, res=x_lat,
                      ld_res=ld_lat, res_mode=1, relu_post=1)
            m = E((nv[lvl], C))
            self.conv(L[f"conv_up_m{lvl}.0"], catm, 2 * C, subm[lvl], nv[lvl], m, C, res=catm, ld_res=2 * C,
                      res_mode=2, relu_pre=1)
            return m

        ci4 = E((nv[4], 144))
        self.conv(L["inv_conv_out"], enc, 128, inv5, nv[4], ci4, 144)
        onehot(4, 1.0, ci4, 144, 128)
        catm4 = E((nv[4], 256))
        self.conv(L["conv_up_instance_block.0"], ci4, 144, subm[4], nv[4], catm4, 256, relu_post=1)
        m4 = ur_block(4, 128, catm4, 256, catm4)
        ci3 = E((nv[3], 80))
        self.conv(L["inv_conv4.0"], m4, 128, inv[4], nv[3], ci3, 80, relu_post=1)
        onehot(3, 2.0, ci3, 80, 64)
        catm3 = E((nv[3], 128))
        self.conv(L["conv_up_instance_block_up4.0"], ci3, 80, subm[3], nv[3], catm3, 128, relu_post=1)
        m3 = ur_block(3, 64, xc[3], 64, catm3)
        ci2 = E((nv[2], 48))
        self.conv(L["inv_conv3.0"], m3, 64, inv[3], nv[2], ci2, 48, relu_post=1)
        onehot(2, 4.0, ci2, 48, 32)
        catm2 = E((nv[2], 64))
        self.conv(L["conv_up_instance_block_up3.0"], ci2, 48, subm[2], nv[2], catm2, 64, relu_post=1)
        m2 = ur_block(2, 32, xc[2], 32, catm2)
        ci1 = E((V, 32))
        self.conv(L["inv_conv2.0"], m2, 32, inv[2], V, ci1, 32, relu_post=1)
        onehot(1, 8.0, ci1, 32, 16)
        catm1 = E((V, 32))
        self.conv(L["conv_up_instance_block_up2.0"], ci1, 32, subm[1], V, catm1, 32, relu_post=1)
        m1 = ur_block(1, 16, xc[1], 16, catm1)
        ci0 = E((V, 32))
        self.conv(L["conv_up_out.0.0"], m1, 16, subm[1], V, ci0, 32, relu_post=1)
        # spconv_unet.py:401 re-uses the stride-1 instance features: same one-hots, copied instead of recomputed
        _lib.check(lib.insmos_copy_cols(ci1.data_ptr() + 4 * 16, 32, ci0.data_ptr() + 4 * 16, 32, V, 16, st), "insmos_copy_cols")
        seg = E((V, 16))
        self.conv(L["conv_up_instance_block_up1.0"], ci0, 32, subm[1], V, seg, 16, relu_post=1)
        vox_logits = E((V, 4))
        self.conv(L["mos_seg"], seg, 16, None, V, vox_logits, 4)
        logits = E((ncur, 3))
        _lib.check(lib.insmos_gather_rows(vox_logits.data_ptr(), 4, 3, pcid.data_ptr(), ncur, logits.data_ptr(), 3, st),
                   "insmos_gather_rows")
        K = int(cnt_k[0].item())  # the one unavoidable read-back: output tensors are sized by it
        self.last_counts["n_boxes"] = K
        self.last_counts["n_candidates"] = int(cnt_c[0].item())
        self._un_debug = dict(xc=xc, enc=enc, ci4=ci4, ci3=ci3, ci2=ci2, ci1=ci1, seg=seg, vox_logits=vox_logits,
                              m=[m1, m2, m3, m4])
        pred = {"pred_boxes": pb[:K], "pred_scores": psc[:K], "pred_labels": pl[:K]}
        return logits, pred

    # ------------------------------------------------------------------------------------------------
    # The 3D branch's coordinate side for a BATCH of windows on the step path (the training step, insmos_amd/train_unet.py):
    # the same C entry points the native runner drives for a launch set (csrc/forward.hip), window = spconv's batch column.
    def unet_tables_windows(self, cur, n_cur_list):
        """cur (sum Ncur_b, 8) fp32: the windows' current points back to back; n_cur_list their counts -> the dict unet() leaves
        in _un_tables (coordinate sets and kernel maps over window-major voxel rows) + 'B', 'win_rows' {level: (B + 1,) row
        starts per window}.  Every window is voxelised on its own (first-seen order, its own max_voxels cap)."""
        lib, st, E = self.lib, self._stream(), self._empty
        B = len(n_cur_list)
        ncur = int(cur.shape[0])
        assert sum(int(v) for v in n_cur_list) == ncur and 1 <= B <= MAX_WINDOWS_PER_LAUNCH
        counts = E((8 + B,), torch.int32)
        Vcap = self.max_voxels * B
        feat = E((Vcap, 8))
        coords1 = E((Vcap, 4), torch.int32)
        num_points = E((Vcap,), torch.int32)
        pcid = E((max(ncur, 1),), torch.int64)
        ukeys = E((max(ncur, 1),), torch.int64)
        uperm = E((max(ncur, 1),), torch.int32)
        starts = np.concatenate([[0], np.cumsum([int(v) for v in n_cur_list])]).astype(np.int32)
        win_start = torch.from_numpy(starts).to(self.device)
        ws = self._workspace(lib.insmos_voxelize_mean_ws_bytes(ncur))
        rng = np.array(self.range, dtype=np.float32)
        vsz = np.array(self.vs, dtype=np.float32)
        key_cells = int(np.prod(self.shape[1]))
        _lib.check(lib.insmos_voxelize_mean_windows(cur.data_ptr(), ncur, 8, self.in_ch, win_start.data_ptr(), B, key_cells, _hp(rng),
                                                    _hp(vsz), self.max_voxels, self.max_points, feat.data_ptr(), 8, coords1.data_ptr(),
                                                    num_points.data_ptr(), pcid.data_ptr(), ukeys.data_ptr(), uperm.data_ptr(),
                                                    counts.data_ptr(), ws.data_ptr(), ws.numel(), st), "insmos_voxelize_mean_windows")
        c = counts.cpu().numpy()
        V, S = int(c[0]), int(c[1])
        nv, coords, keys, perm, nkeys = {1: V}, {1: coords1[:V]}, {1: ukeys[:S]}, {1: uperm[:S]}, {1: S}
        win_rows = {1: np.asarray(c[4:4 + B + 1], np.int64)}
        k333, s222, p111 = _np_i32([3, 3, 3]), _np_i32([2, 2, 2]), _np_i32([1, 1, 1])
        k311, s211, p000 = _np_i32([3, 1, 1]), _np_i32([2, 1, 1]), _np_i32([0, 0, 0])

        def down_coords(lvl_in, ks, stv, pd, oshape):
            n_in = nv[lvl_in]
            cap = max(min(n_in * int(np.prod(ks)), int(np.prod(oshape)) * B), 1)
            ok, oc = E((cap,), torch.int64), E((cap, 4), torch.int32)
            if n_in == 0:
                return ok[:0], oc[:0], 0
            osh = _np_i32(oshape)
            w = self._workspace(lib.insmos_down_coords3d_ws_bytes_b(_hp(osh), B))
            _lib.check(lib.insmos_down_coords3d_b(coords[lvl_in].data_ptr(), n_in, _hp(ks), _hp(stv), _hp(pd), _hp(osh), B, ok.data_ptr(),
                                                  oc.data_ptr(), counts.data_ptr(), w.data_ptr(), w.numel(), st), "insmos_down_coords3d_b")
            no = int(counts[0].item())
            return ok[:no], oc[:no], no

        for l in (2, 3, 4):
            keys[l], coords[l], nv[l] = down_coords(l - 1, k333, s222, p111, self.shape[l])
            perm[l], nkeys[l] = None, nv[l]
        keys[5], coords[5], nv[5] = down_coords(4, k311, s211, p000, self.shape[5])
        perm[5], nkeys[5] = None, nv[5]
        for l in (2, 3, 4, 5):   # rows of the generated levels are in ascending (b, z, y, x) order: a window's rows are one range
            bcol = coords[l][:, 0].contiguous()
            win_rows[l] = torch.searchsorted(bcol, torch.arange(B + 1, device=self.device, dtype=torch.int32)).cpu().numpy().astype(np.int64) \
                if nv[l] else np.zeros(B + 1, np.int64)
        self.last_counts["unet_voxels"] = [nv[l] for l in (1, 2, 3, 4, 5)]
        t27, t3 = spconv_tap_offsets((3, 3, 3)), spconv_tap_offsets((3, 1, 1))
        d_subm = [[0, kz - 1, ky - 1, kx - 1] for kz, ky, kx in t27]
        d_inv = [[0, 1 - kz, 1 - ky, 1 - kx] for kz, ky, kx in t27]
        d_down5 = [[0, kz, 0, 0] for kz, ky, kx in t3]
        d_inv5 = [[0, -kz, 0, 0] for kz, ky, kx in t3]
        subm = {l: self.build_nbr(coords[l], nv[l], keys[l], perm[l], nkeys[l], 1, self.shape[l], d_subm) for l in (1, 2, 3, 4)}
        down = {l: self.build_nbr(coords[l], nv[l], keys[l - 1], perm[l - 1], nkeys[l - 1], 1, self.shape[l - 1], d_subm,
                                  mul=[1, 2, 2, 2]) for l in (2, 3, 4)}
        inv = {l: self.build_nbr(coords[l - 1], nv[l - 1], keys[l], perm[l], nkeys[l], 1, self.shape[l], d_inv,
                                 div=[1, 2, 2, 2]) for l in (2, 3, 4)}
        down5 = self.build_nbr(coords[5], nv[5], keys[4], None, nkeys[4], 1, self.shape[4], d_down5, mul=[1, 2, 1, 1])
        inv5 = self.build_nbr(coords[4], nv[4], keys[5], None, nkeys[5], 1, self.shape[5], d_inv5, div=[1, 2, 1, 1])
        self._un_tables = dict(subm=subm, down=down, inv=inv, down5=down5, inv5=inv5, coords=coords, pcid=pcid, feat=feat[:V],
                               num_points=num_points[:V], B=B, win_rows=win_rows)
        return self._un_tables

    def detect_windows(self, head, up, B):
        """detect() for B stacked head maps (rows [b][row][col]): -> pred_boxes (B, post_max, 7), scores (B, post_max), labels
        (B, post_max) int64, kept counts (B, 4) int32 (slot 0), candidate counts (B, 4)."""
        lib, st, E = self.lib, self._stream(), self._empty
        H2, W2 = 2 * self.bevH, 2 * self.bevW
        ncell = H2 * W2 * B
        cb, cs = E((B, self.pre_max, 7)), E((B, self.pre_max))
        cl, cc = E((B, self.pre_max), torch.int32), E((B, self.pre_max), torch.int32)
        cnt_c, cnt_k = E((B, 4), torch.int32), E((B, 4), torch.int32)
        w = self._workspace(lib.insmos_center_decode_select_ws_bytes(ncell))
        _lib.check(lib.insmos_center_decode_select_b(head.data_ptr(), self.head_ld, self.ncls, H2, W2, up, B, self.out_factor, self.tvs[0],
                                                     self.tvs[1], self.range[0], self.range[1], self.score_thresh, self.pre_max,
                                                     cb.data_ptr(), cs.data_ptr(), cl.data_ptr(), cc.data_ptr(), cnt_c.data_ptr(),
                                                     w.data_ptr(), w.numel(), st), "insmos_center_decode_select_b")
        keep = E((B, self.post_max), torch.int32)
        w = self._workspace(lib.insmos_nms_ws_bytes_b(self.pre_max, B))
        _lib.check(lib.insmos_nms_rotated_bev_b(cb.data_ptr(), cnt_c.data_ptr(), self.pre_max, self.nms_thresh, self.post_max, B,
                                                keep.data_ptr(), cnt_k.data_ptr(), w.data_ptr(), w.numel(), st), "insmos_nms_rotated_bev_b")
        pb, psc = E((B, self.post_max, 7)), E((B, self.post_max))
        pl = E((B, self.post_max), torch.int64)
        _lib.check(lib.insmos_gather_preds_b(cb.data_ptr(), cs.data_ptr(), cl.data_ptr(), keep.data_ptr(), cnt_k.data_ptr(), self.pre_max,
                                             self.post_max, B, pb.data_ptr(), psc.data_ptr(), pl.data_ptr(), st), "insmos_gather_preds_b")
        return pb, psc, pl, cnt_k, cnt_c

    def instance_onehot_windows(self, pb, pl, cnt_k, B, level_coords, n, mult, out, ld, col, scratch):
        """instance_onehot() over window-major voxel rows: a voxel meets the boxes of ITS window (coords[:, 0]) only."""
        lo = np.array(self.range[0:3], dtype=np.float32)
        vsz = np.array(self.vs, dtype=np.float32)
        _lib.check(self.lib.insmos_boxes_to_onehot_b(pb.data_ptr(), pl.data_ptr(), cnt_k.data_ptr(), self.post_max, B, _hp(lo), _hp(vsz), 8.0,
                                                     float(mult), level_coords.data_ptr(), n, self.ncls, 16, 1 if self.quirk_exact else 0,
                                                     out.data_ptr() + 4 * col, ld, scratch.data_ptr(), self._stream()),
                   "insmos_boxes_to_onehot_b")

    # ------------------------------------------------------------------------------------------------
    def forward_window(self, pts, native=None):
        """One batch item of InsMOS_Model.forward(..., 'test') (models/models.py:313-364).

        native=True runs the window through insmos_forward_window (csrc/forward.hip: the same operator sequence
        driven from C++, one foreign call, no interpreter work between launches); native=False issues the operators
        step by step from Python and keeps every intermediate for inspection (tests, profiling hooks).  Both give
        identical bits (tests/test_gpu_model.py).  Default: the engine's `native` attribute."""
        pts = self._check_points(pts)
        if self.native if native is None else native:
            return self._forward_native(pts)
        self._conv_log = []
        self._conv_nin = []
        cur = self.motionnet(pts)
        return self.unet(cur)

    @staticmethod
    def _check_points(pts):
        if pts.dtype != torch.float32 or pts.device.type != "cuda" or pts.dim() != 2 or pts.shape[1] < 5:
            raise ValueError("past_point_clouds must be a float32 CUDA tensor of shape (N, 5) [x,y,z,intensity,t]")
        return pts if pts.stride(1) == 1 else pts.contiguous()

    # ------------------------------------------------------------------------------------------------
    def _native_ctx(self):
        """The immutable C++ context (layer table + config) -- built once, shared by clones."""
        if self._ctx_box[0] is None:
            cfgs = _lib.NetCfg()
            cfgs.w0_const, cfgs.b0_const = self.w0_const.data_ptr(), self.b0_const.data_ptr()
            cfgs.nbr_bev = self.nbr_bev.data_ptr()
            cfgs.vs[:] = self.vs
            cfgs.dt = self.dt
            cfgs.range[:] = self.range
            cfgs.score_thresh, cfgs.nms_thresh, cfgs.out_factor = self.score_thresh, self.nms_thresh, self.out_factor
            cfgs.tvs[:] = self.tvs[:2]
            cfgs.in_ch, cfgs.ncls, cfgs.max_voxels, cfgs.max_points = self.in_ch, self.ncls, self.max_voxels, self.max_points
            for l in (1, 2, 3, 4, 5):
                cfgs.shape[l][:] = self.shape[l]
            cfgs.bevD, cfgs.bevH, cfgs.bevW, cfgs.nbev = self.bevD, self.bevH, self.bevW, self.nbev
            cfgs.n_bev_layers, cfgs.up_ch, cfgs.head_ld = self.n_bev_layers, self.up_ch, self.head_ld
            cfgs.pre_max, cfgs.post_max, cfgs.quirk_exact = self.pre_max, self.post_max, 1 if self.quirk_exact else 0
            names = sorted(self.L)
            arr_n = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
            arr_l = (_lib.ConvW * len(names))()
            for i, n in enumerate(names):
                l = self.L[n]
                arr_l[i].w, arr_l[i].b, arr_l[i].K, arr_l[i].cin, arr_l[i].cout = l.w.data_ptr(), l.b.data_ptr(), l.K, l.cin, l.cout
            ctx = ctypes.c_void_p()
            _lib.check(self.lib.insmos_ctx_create(ctypes.byref(cfgs), arr_n, arr_l, len(names), ctypes.byref(ctx)),
                       "insmos_ctx_create")
            # the holder frees the C++ object when the last engine (or clone) that can still run on it lets go of the box;
            # it also keeps the device tensors the context points at alive for as long as the context exists
            self._ctx_box[0] = _NativeCtx(self.lib, ctx, (self.L, self.w0_const, self.b0_const, self.nbr_bev))
        return self._ctx_box[0].handle

    def _forward_native(self, pts):
        return self.forward_windows([pts])[0][:2]

    def forward_windows(self, pts_list):
        """The batch list of InsMOS_Model.forward(..., 'test') in ONE set of launches (insmos_forward_windows,
        csrc/forward.hip): where the reference walks the list window by window (models/models.py:313), the B windows share
        every launch -> [(logits (Ncur_b, 3), pred dict, current_point or None)] per window, each with the bits the window
        gets alone (tests/test_gpu_batched.py)."""
        if not self.const_input:
            raise NotImplementedError("the native runner implements the constant-input first layer only")
        B = len(pts_list)
        if B < 1 or B > MAX_WINDOWS_PER_LAUNCH:
            raise ValueError(f"1 .. {MAX_WINDOWS_PER_LAUNCH} windows per launch set, got {B}")
        pts_list = [self._check_points(p) for p in pts_list]
        ld = pts_list[0].stride(0)
        if any(p.stride(0) != ld for p in pts_list):
            pts_list = [p[:, :5].contiguous() for p in pts_list]
            ld = 5
        ctx = self._native_ctx()
        N = sum(int(p.shape[0]) for p in pts_list)
        if self._arena is None:
            self._arena = torch.empty(900 * N + (256 << 20), dtype=torch.uint8, device=self.device)   # (generous: a retry repeats launches)
        win_ptr = (ctypes.c_void_p * B)(*[p.data_ptr() for p in pts_list])
        win_n = (ctypes.c_int64 * B)(*[int(p.shape[0]) for p in pts_list])
        outs = (_lib.ForwardOut * B)()
        for _ in range(12):
            rc = self.lib.insmos_forward_windows(ctx, win_ptr, win_n, B, ld, self._arena.data_ptr(), self._arena.numel(),
                                                 self._stream(), outs)
            if rc != -3:  # INSMOS_EWORKSPACE: grow the arena and retry
                break
            self._arena = None
            self._arena = torch.empty(max(int(outs[0].arena_needed * 1.5), 1 << 20), dtype=torch.uint8, device=self.device)
        if rc == -4:  # INSMOS_EBATCH: the set's finest table would pass 2 GiB, or a window's time range does not fit beside the
            # others (the batch narrows the time field by B) -> two smaller launch sets, same bits per window
            h = B // 2
            return self.forward_windows(pts_list[:h]) + self.forward_windows(pts_list[h:])
        if rc == -1 and outs[0].n_out_of_window:
            raise ValueError(f"{int(outs[0].n_out_of_window)} points fall outside the +-32768-voxel key window")
        if rc == -1 and outs[0].me_voxels[0] > 0 and any(int(o.n_cur) == 0 for o in outs):
            raise ValueError("window has no current-scan points (t == 0)")
        _lib.check(rc, "insmos_forward_windows")
        a = self._arena

        def view(off, count, dtype, shape):
            nb = count * torch.empty((), dtype=dtype).element_size()
            return a[off:off + nb].view(dtype).reshape(shape).clone()  # the arena is rewritten by the next batch

        # the arena is rewritten by the next launch set: ONE copy per output array and set (the windows' slices of it are
        # views), not four per window
        pm = self.post_max
        n_all = sum(int(o.n_cur) for o in outs)
        logits_all = view(outs[0].logits_off, n_all * 3, torch.float32, (n_all, 3))
        boxes_all = view(outs[0].boxes_off, B * pm * 7, torch.float32, (B, pm, 7))
        scores_all = view(outs[0].scores_off, B * pm, torch.float32, (B, pm))
        labels_all = view(outs[0].labels_off, B * pm, torch.int64, (B, pm))
        cur_all = view(outs[0].cur_points_off, n_all * 8, torch.float32, (n_all, 8)) if self.keep_current_points else None
        results, c0 = [], 0
        for b, out in enumerate(outs):
            ncur, K = int(out.n_cur), int(out.n_boxes)
            pred = {"pred_boxes": boxes_all[b, :K], "pred_scores": scores_all[b, :K], "pred_labels": labels_all[b, :K]}
            results.append((logits_all[c0:c0 + ncur], pred, cur_all[c0:c0 + ncur] if cur_all is not None else None))
            c0 += ncur
        o0 = outs[0]
        self.last_counts = {"me_voxels": [int(v) for v in o0.me_voxels], "n_cur": int(o0.n_cur),
                            "unet_voxels": [int(v) for v in o0.unet_voxels], "n_boxes": int(o0.n_boxes),
                            "n_candidates": int(o0.n_candidates), "batch": B,
                            "per_window": [{"n_cur": int(o.n_cur), "voxels": int(o.unet_voxels[0]), "n_boxes": int(o.n_boxes),
                                            "n_candidates": int(o.n_candidates)} for o in outs]}
        if self.keep_current_points:
            self.last_current_points = results[0][2]
        return results

    def clone_shared(self):
        """A second runner over the SAME device weights, tables and native context, with its own arena/workspace --
        one per window in flight (InsMOS_Model.forward with several batch items)."""
        import copy
        e = copy.copy(self)
        e._ws, e._arena, e._conv_log, e._conv_nin, e.last_counts, e.layer_timing = None, None, [], [], {}, None
        e._split_owner = False   # the clone shares the weights (and their split copies); only the original unregisters them
        return e

    def algorithmic_work(self):
        """Algorithmic work of the LAST forward_window over all sparse_conv launches:
        flops = sum 2*pairs*Cin*Cout, gather bytes = sum 4*pairs*(Cin+Cout) + 8*pairs.  pairs = valid
        entries of the layer's neighbour table (n_out for 1x1 layers).  Costs a device reduction per
        distinct table: bench/profiling only."""
        cache = {}
        flops = gather = pairs_total = compulsory = 0
        for (nbr, n_out, layer, row0), n_in in zip(self._conv_log, self._conv_nin):
            if nbr is None:
                pairs = n_out - row0
            else:
                key = (nbr.data_ptr(), row0)
                if key not in cache:
                    tab = nbr.nbr if isinstance(nbr, NbrTable) else nbr
                    cache[key] = int((tab[:, row0:] >= 0).sum().item())
                pairs = cache[key]
            cin = layer.flops_per_pair // (2 * layer.cout_real)
            if (nbr is self.nbr_bev or nbr is None) and layer.name in getattr(self, "_bev_exec_pairs", {}):
                pairs = self._bev_exec_pairs[layer.name]   # executed by the runner's constant-region skipping (bev_skip_accounting)
            flops += pairs * layer.flops_per_pair
            gather += 4 * pairs * (cin + layer.cout_real) + 8 * pairs
            pairs_total += pairs
            # compulsory HBM bytes: input rows and output rows once, plus the table rows read
            rows = n_out - row0
            compulsory += 4 * (n_in * cin + rows * layer.cout_real) + (4 * layer.K * rows if nbr is not None else 0)
        return {"flops": flops, "gather_bytes": gather, "pairs": pairs_total, "launches": len(self._conv_log),
                "compulsory_bytes": compulsory}

"""Stand-in for the spconv.pytorch API subset the reference touches (spconv_unet.py:14-18,76-83,120-207; voxel_generate.py;
height_compression.py:26), backed by oracle/ref_ops.py -- see oracle/shims/README.md (pins wiring, not primitive semantics)."""
import types

import numpy as np
import torch
import torch.nn as nn

from oracle import ref_ops as R

from . import utils  # noqa: F401


def _t3(v):
    return tuple(int(x) for x in (v if isinstance(v, (list, tuple)) else (v, v, v)))


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, indice_dict=None, _keys=None):
        self.features = features
        self.indices = indices                       # (V, 4) int32 [b, z, y, x]
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = batch_size
        self.indice_dict = indice_dict if indice_dict is not None else {}
        self._keys = _keys                           # (sorted keys, perm) of this coordinate set, built lazily

    def replace_feature(self, feature):
        return SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.indice_dict, self._keys)

    def coords_zyx(self):
        return self.indices[:, 1:].detach().cpu().numpy().astype(np.int32)

    def keys(self):
        if self._keys is None:
            self._keys = R.sorted_index(R.key3(self.coords_zyx(), self.spatial_shape))
        return self._keys

    def dense(self):
        """(N, C, D, H, W) -- height_compression.py:26."""
        D, H, W = self.spatial_shape
        C = self.features.shape[1]
        out = torch.zeros((self.batch_size, C, D, H, W), dtype=self.features.dtype)
        idx = self.indices.long()
        out[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = self.features
        return out


class SparseModule(nn.Module):
    pass


class SparseConvolution(SparseModule):
    kind = "subm"

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None):
        super().__init__()
        assert dilation == 1 and groups == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _t3(kernel_size), _t3(stride), _t3(padding)
        self.indice_key = indice_key
        # spconv 2.3.6 weight layout: (Cout, kz, ky, kx, Cin)
        self.weight = nn.Parameter(torch.zeros((out_channels,) + self.kernel_size + (in_channels,)))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def _taps(self):
        w = self.weight.detach().cpu().numpy().astype(np.float32)
        co, kz, ky, kx, ci = w.shape
        return np.ascontiguousarray(w.reshape(co, kz * ky * kx, ci).transpose(1, 2, 0))

    def forward(self, x):
        feat = x.features.detach().cpu().numpy().astype(np.float32)
        if self.kind == "subm":
            # output set == input set; the rulebook is cached under indice_key and shared by every layer naming it
            cache = x.indice_dict.get(self.indice_key)
            if cache is None or cache.get("kind") != "subm":
                ks, perm = x.keys()
                cache = {"kind": "subm", "nbr": R.spconv_nbr_subm(x.coords_zyx(), ks, perm, x.spatial_shape, self.kernel_size)}
                if self.indice_key is not None:
                    x.indice_dict[self.indice_key] = cache
            out = SparseConvTensor(None, x.indices, x.spatial_shape, x.batch_size, x.indice_dict, x._keys)
            nbr = cache["nbr"]
        elif self.kind == "spconv":
            ks, perm = x.keys()
            oc, ok, oshape = R.spconv_down_coords(x.coords_zyx(), x.spatial_shape, self.kernel_size, self.stride, self.padding)
            nbr = R.spconv_nbr_down(oc, ks, perm, x.spatial_shape, self.kernel_size, self.stride, self.padding)
            ind = torch.from_numpy(np.concatenate([np.zeros((len(oc), 1), np.int32), oc], 1))
            out = SparseConvTensor(None, ind, oshape, x.batch_size, x.indice_dict, (ok, None))
            x.indice_dict[self.indice_key] = {"kind": "spconv", "in_indices": x.indices, "in_shape": x.spatial_shape,
                                              "in_keys": x._keys, "out_keys": (ok, None), "out_shape": oshape,
                                              "ksize": self.kernel_size, "stride": self.stride, "padding": self.padding}
        else:  # inverse: the pairs of the forward layer with the same indice_key, reversed
            c = x.indice_dict[self.indice_key]
            assert c["kind"] == "spconv" and list(c["out_shape"]) == list(x.spatial_shape)
            fine = c["in_indices"][:, 1:].detach().cpu().numpy().astype(np.int32)
            ks, perm = x.keys()
            nbr = R.spconv_nbr_inverse(fine, ks, perm, x.spatial_shape, c["ksize"], c["stride"], c["padding"])
            out = SparseConvTensor(None, c["in_indices"], c["in_shape"], x.batch_size, x.indice_dict, c["in_keys"])
        if torch.is_grad_enabled():  # training-wiring golden: differentiable torch index ops, weight in its own layout
            from MinkowskiEngine import conv_torch
            co, kz, ky, kx, ci = self.weight.shape
            y = conv_torch(x.features, nbr, self.weight.reshape(co, kz * ky * kx, ci).permute(1, 2, 0))
            if self.bias is not None:
                y = y + self.bias
        else:
            y = torch.from_numpy(np.asarray(R.sparse_conv(feat, nbr, self._taps()), np.float32))
            if self.bias is not None:
                y = y + self.bias.detach()
        out.features = y
        return out


class SubMConv3d(SparseConvolution):
    kind = "subm"


class SparseConv3d(SparseConvolution):
    kind = "spconv"


class SparseInverseConv3d(SparseConvolution):
    kind = "inverse"

    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, algo=None):
        super().__init__(in_channels, out_channels, kernel_size, bias=bias, indice_key=indice_key)


class SparseSequential(SparseModule):
    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                x = x.replace_feature(m(x.features))   # dense layers (BatchNorm1d, ReLU) act on the feature matrix
            else:
                x = m(x)
        return x


conv = types.SimpleNamespace(SparseConvolution=SparseConvolution)  # spconv_unet.py:37 isinstance check

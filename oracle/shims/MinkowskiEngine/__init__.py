"""Stand-in for the MinkowskiEngine API subset the reference touches -- see oracle/shims/README.md (test infrastructure,
backed by oracle/ref_ops.py; pins wiring, not primitive semantics)."""
import math

import numpy as np
import torch
import torch.nn as nn

from oracle import ref_ops as R

from . import utils  # noqa: F401  (ME.utils.sparse_collate / kaiming_normal_)


def _as_list(v, d):
    return [int(v)] * d if np.isscalar(v) else [int(x) for x in v]


def conv_torch(x, nbr, taps):
    """y[o] = sum_k x[nbr[k][o]] @ taps[k] with torch index ops (autograd-capable); nbr None = 1x1."""
    if nbr is None:
        return x @ taps[0]
    y = torch.zeros((nbr.shape[1], taps.shape[2]), dtype=x.dtype)
    for k in range(nbr.shape[0]):
        o = np.nonzero(nbr[k] >= 0)[0]
        if len(o):
            y = y.index_add(0, torch.from_numpy(o), x[torch.from_numpy(nbr[k][o].astype(np.int64))] @ taps[k])
    return y


class CoordinateManager:
    """Coordinate maps per tensor stride; coarser maps derive from the finest one (floor(c / s) * s, unique)."""

    def __init__(self, coords, keys):
        self.maps = {(1, 1, 1, 1): (coords, keys)}

    def get(self, stride):
        stride = tuple(int(s) for s in stride)
        if stride not in self.maps:
            assert stride[0] == stride[1] == stride[2] and stride[3] == 1, stride
            level = int(round(math.log2(stride[0])))
            c0, k0 = self.maps[(1, 1, 1, 1)]
            pc, pk, _ = R.me_stride_down(c0, k0, level)
            self.maps[stride] = (pc, pk)
        return self.maps[stride]


class SparseTensor:
    def __init__(self, features, coordinate_manager=None, tensor_stride=(1, 1, 1, 1), coordinates=None, device=None):
        self.F = features
        self.manager = coordinate_manager
        self.tensor_stride = tuple(tensor_stride)

    @property
    def features(self):
        return self.F

    @property
    def C(self):
        c, _ = self.manager.get(self.tensor_stride)
        return torch.from_numpy(np.concatenate([np.zeros((len(c), 1), np.int32), c], 1))

    coordinates = C

    def _like(self, features, tensor_stride=None):
        return SparseTensor(features, self.manager, self.tensor_stride if tensor_stride is None else tensor_stride)

    def slice(self, field):
        """Voxel feature to every source point of the field (motionnet.py:38)."""
        out = TensorField(self.F[torch.from_numpy(field.inverse.astype(np.int64))], field.coordinates.clone())
        out.inverse = field.inverse
        return out


class TensorField:
    def __init__(self, features, coordinates):
        self.features = features
        self.coordinates = coordinates  # (N, 1 + 4) float: [batch, x, y, z, t] already divided by the quantisation
        self.inverse = None

    F = property(lambda self: self.features)
    C = property(lambda self: self.coordinates)

    def sparse(self):
        """floor -> unique -> unweighted average of the features (motionnet.py:34-36)."""
        pts4 = self.coordinates[:, 1:].detach().cpu().numpy().astype(np.float32)
        coords, keys, inverse = R.me_quantize(pts4, np.ones(4, np.float32))
        self.inverse = inverse
        feat = R.me_feature_average(self.features.detach().cpu().numpy(), inverse, len(coords))
        return SparseTensor(torch.from_numpy(feat), CoordinateManager(coords, keys))


class _ConvBase(nn.Module):
    transpose = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, kernel_generator=None,
                 dimension=None, **kw):
        super().__init__()
        assert dimension == 4 and dilation == 1 and kernel_generator is None
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _as_list(kernel_size, dimension)
        self.stride = _as_list(stride, dimension)
        vol = int(np.prod(self.kernel_size))
        shape = (vol, in_channels, out_channels) if vol > 1 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.zeros(shape))
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None

    def forward(self, x):
        ts_in = x.tensor_stride
        taps = self.kernel.detach().cpu().numpy().astype(np.float32)
        taps = taps[None] if taps.ndim == 2 else taps
        in_coords, in_keys = x.manager.get(ts_in)
        if int(np.prod(self.kernel_size)) == 1 and all(s == 1 for s in self.stride):
            nbr, ts_out = None, ts_in
        elif not self.transpose:
            ts_out = tuple(t * s for t, s in zip(ts_in, self.stride))
            out_coords, _ = x.manager.get(ts_out)
            offs = R.me_kernel_offsets(self.kernel_size, ts_in)
            nbr = R.me_nbr(out_coords, in_keys, offs, +1)
        else:
            ts_out = tuple(t // s for t, s in zip(ts_in, self.stride))
            out_coords, _ = x.manager.get(ts_out)  # the cached finer map (needed for ME.cat, minkunet.py:164,171,178)
            offs = R.me_kernel_offsets(self.kernel_size, ts_out)
            nbr = R.me_nbr(out_coords, in_keys, offs, -1)
        if torch.is_grad_enabled():  # training-wiring golden: the same contraction as differentiable torch index ops
            kt = self.kernel[None] if self.kernel.dim() == 2 else self.kernel
            y = conv_torch(x.F, nbr, kt)
            if self.bias is not None:
                y = y + self.bias
        else:
            y = R.sparse_conv(x.F.detach().cpu().numpy().astype(np.float32), nbr, taps)
            y = torch.from_numpy(np.asarray(y, np.float32))
            if self.bias is not None:
                y = y + self.bias.detach()
        return x._like(y, ts_out)


class MinkowskiConvolution(_ConvBase):
    transpose = False


class MinkowskiConvolutionTranspose(_ConvBase):
    transpose = True


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        return x._like(self.bn(x.F))


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return x._like(torch.relu(x.F))


def cat(*tensors):
    t0 = tensors[0]
    assert all(t.tensor_stride == t0.tensor_stride and t.manager is t0.manager for t in tensors)
    return t0._like(torch.cat([t.F for t in tensors], 1))


class _Absent(nn.Module):
    """Names the reference mentions in classes this path never instantiates (resnet.py:58-86,165-186)."""

    def __init__(self, *a, **k):
        raise NotImplementedError(type(self).__name__ + " is not part of the stand-in (unused on the InsMOS path)")


class MinkowskiInstanceNorm(_Absent): pass          # noqa: E701
class MinkowskiMaxPooling(_Absent): pass            # noqa: E701
class MinkowskiDropout(_Absent): pass               # noqa: E701
class MinkowskiGELU(_Absent): pass                  # noqa: E701
class MinkowskiGlobalMaxPooling(_Absent): pass      # noqa: E701
class MinkowskiLinear(_Absent): pass                # noqa: E701
class MinkowskiSinusoidal(_Absent): pass            # noqa: E701
class MinkowskiToSparseTensor(_Absent): pass        # noqa: E701

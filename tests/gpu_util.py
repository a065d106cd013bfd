"""Helpers for the -m gpu parity tests: everything goes through the C ABI (ctypes) with torch tensors
as device memory."""
import ctypes

import numpy as np
import torch

from insmos_amd import _lib

DEV = "cuda:0"


def lib():
    return _lib.load()


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def hp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def ws(nbytes):
    return torch.empty(int(nbytes) + 4096, dtype=torch.uint8, device=DEV)


def u64(t):
    """int64 device tensor holding uint64 bits -> numpy uint64."""
    return t.cpu().numpy().view(np.uint64)


def pack_layer(taps, bias, cin_pad, cout_store):
    from insmos_amd.engine import ConvLayer
    return ConvLayer(lib(), np.ascontiguousarray(taps, np.float32), bias, cin_pad, cout_store, torch.device(DEV))


def tap_masks(nbr_np):
    """Reference construction of the per-16-row-group active-tap bitmasks (what insmos_build_nbr emits)."""
    K, n = nbr_np.shape
    ng = (n + 15) // 16
    v = np.zeros((K, ng * 16), bool)
    v[:, :n] = nbr_np >= 0
    any16 = v.reshape(K, ng, 16).any(2)  # (K, ng)
    m = np.zeros((ng, 4), np.uint32)
    for k in range(K):
        m[:, k >> 5] |= (any16[k].astype(np.uint32) << np.uint32(k & 31))
    return m


def run_conv(layer, x, nbr, n_out, ld_out=None, col_out=0, res=None, res_mode=0, relu_pre=0, relu_post=0, out=None,
             mask=None):
    ld_out = ld_out or layer.cout
    if out is None:
        out = torch.zeros((n_out, ld_out), dtype=torch.float32, device=DEV)
    rc = lib().insmos_sparse_conv(x.data_ptr(), x.shape[0], x.stride(0), layer.cin,
                                  nbr.data_ptr() if nbr is not None else None,
                                  mask.data_ptr() if mask is not None else None,
                                  layer.K, n_out, layer.w.data_ptr(), layer.b.data_ptr(), out.data_ptr() + 4 * col_out,
                                  ld_out, layer.cout, res.data_ptr() if res is not None else None,
                                  res.stride(0) if res is not None else 0, res_mode, relu_pre, relu_post, stream())
    _lib.check(rc, "insmos_sparse_conv")
    torch.cuda.synchronize()
    return out

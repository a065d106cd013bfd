"""
oracle/ref_ops.py -- CPU restatement (numpy + oracle/coracle.c) of the operators on the InsMOS
inference hot path.  TEST INFRASTRUCTURE ONLY (see oracle/coracle.c header): only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY STATUS: functions marked [pinned] are checked bit-/tolerance-exactly against golden vectors
generated from the reference's own importable / compilable code (tests/golden/make_golden.py).
Functions marked [dep-knowledge, PARITY UNPINNED] restate MinkowskiEngine / spconv 2.3.6
semantics at the reference's call sites; neither dependency is available, so they are anchored
only on dense-conv cross-checks and the survey's known voxel/pair counts.

Canonical orders (documented in DESIGN.md; the reference's CUDA path is itself nondeterministic):
  * 4D (MotionNet) voxels: ascending key4 = (t+32768)<<48 | morton3(x+32768, y+32768, z+32768).
  * 3D stride-1 voxels: first-seen point order (spconv CPU point-to-voxel semantics).
  * 3D generated levels (strided SparseConv3d outputs): ascending linear index (z*H + y)*W + x.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_coracle(force=False):
    """gcc-compile oracle/coracle.c -> oracle/_build/libcoracle.so (idempotent)."""
    src = os.path.join(_HERE, "coracle.c")
    outdir = os.path.join(_HERE, "_build")
    out = os.path.join(outdir, "libcoracle.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(outdir, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-std=c11", src, "-o", out, "-lm"])
    return out


def _lib():
    global _LIB
    if _LIB is None:
        path = build_coracle()
        L = ctypes.CDLL(path)
        vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        L.co_nbr_lookup.argtypes = [vp, i64, vp, vp, i64, vp]
        L.co_sparse_conv.argtypes = [vp, i64, i32, vp, i32, i64, vp, i32, vp, i64]
        L.co_iou_bev_matrix.argtypes = [vp, i32, vp, i32, vp]
        L.co_overlap_bev_matrix.argtypes = [vp, i32, vp, i32, vp]
        L.co_nms_bev.argtypes = [vp, i32, f32, vp]
        L.co_nms_bev.restype = i32
        L.co_boxes_to_onehot.argtypes = [vp, i64, vp, i32, vp, i32, i32]
        L.co_points_in_instance_boxes.argtypes = [vp, i64, i32, vp, i32, vp, i32, f32, i32]
        L.co_num_threads.restype = i32
        _LIB = L
    return _LIB


def num_threads():
    return int(_lib().co_num_threads())


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ------------------------------------------------------------------------------------------------
# keys
# ------------------------------------------------------------------------------------------------
def _spread3(v):
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


KEY4_BIAS = 32768
INVALID_KEY = np.uint64(0xFFFFFFFFFFFFFFFF)


def key4(coords):
    """coords (n,4) int [x,y,z,t] -> uint64 Morton-in-space, t-major key; INVALID_KEY if out of the
    16-bit window."""
    c = coords.astype(np.int64) + KEY4_BIAS
    ok = np.all((c >= 0) & (c < 65536), axis=1)
    cu = np.where(ok[:, None], c, 0).astype(np.uint64)
    k = (cu[:, 3] << np.uint64(48)) | _spread3(cu[:, 0]) | (_spread3(cu[:, 1]) << np.uint64(1)) | (
        _spread3(cu[:, 2]) << np.uint64(2))
    return np.where(ok, k, INVALID_KEY)


def key3(coords_zyx, shape):
    """coords (n,3) int [z,y,x] inside shape [D,H,W] -> linear uint64 key; INVALID_KEY outside."""
    c = coords_zyx.astype(np.int64)
    D, H, W = [int(s) for s in shape]
    ok = (c[:, 0] >= 0) & (c[:, 0] < D) & (c[:, 1] >= 0) & (c[:, 1] < H) & (c[:, 2] >= 0) & (c[:, 2] < W)
    k = ((c[:, 0] * H + c[:, 1]) * W + c[:, 2]).astype(np.uint64)
    return np.where(ok, k, INVALID_KEY)


def nbr_lookup(query_keys, in_keys_sorted, in_perm=None):
    """query_keys (K, n_out) uint64 -> (K, n_out) int32 input row ids (-1 = absent)."""
    q = np.ascontiguousarray(query_keys, dtype=np.uint64)
    ik = np.ascontiguousarray(in_keys_sorted, dtype=np.uint64)
    out = np.empty(q.shape, dtype=np.int32)
    perm = None if in_perm is None else np.ascontiguousarray(in_perm, dtype=np.int32)
    _lib().co_nbr_lookup(_p(q), q.size, _p(ik), None if perm is None else _p(perm), ik.size, _p(out))
    return out


def nbr_lookup_numpy(query_keys, in_keys_sorted, in_perm=None):
    """Pure-numpy twin of nbr_lookup (cross-check of the C helper)."""
    q = np.asarray(query_keys, dtype=np.uint64)
    pos = np.searchsorted(in_keys_sorted, q.ravel())
    pos_c = np.minimum(pos, max(len(in_keys_sorted) - 1, 0))
    hit = (pos < len(in_keys_sorted)) & (in_keys_sorted[pos_c] == q.ravel()) & (q.ravel() != INVALID_KEY)
    ids = pos_c if in_perm is None else np.asarray(in_perm)[pos_c]
    return np.where(hit, ids, -1).astype(np.int32).reshape(q.shape)


def sparse_conv(feat_in, nbr, W, n_out=None):
    """out[o] = sum_k feat_in[nbr[k,o]] @ W[k]   (W: (K,Cin,Cout) fp32).  nbr None -> 1x1."""
    x = np.ascontiguousarray(feat_in, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    K, cin, cout = W.shape
    assert x.shape[1] == cin, (x.shape, W.shape)
    if nbr is None:
        assert K == 1
        n_out = x.shape[0]
    else:
        nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        assert nbr.shape[0] == K
        n_out = nbr.shape[1]
    out = np.empty((n_out, cout), dtype=np.float32)
    _lib().co_sparse_conv(_p(x), x.shape[1], cin, None if nbr is None else _p(nbr), K, n_out, _p(W), cout, _p(out),
                          cout)
    return out


def sparse_conv_numpy(feat_in, nbr, W):
    """Pure-numpy twin of sparse_conv (per-offset gather -> matmul -> scatter-add)."""
    K, cin, cout = W.shape
    if nbr is None:
        return (feat_in.astype(np.float32) @ W[0].astype(np.float32)).astype(np.float32)
    out = np.zeros((nbr.shape[1], cout), dtype=np.float32)
    for k in range(K):
        o = np.nonzero(nbr[k] >= 0)[0]
        if len(o):
            out[o] += feat_in[nbr[k, o]].astype(np.float32) @ W[k].astype(np.float32)
    return out


def batchnorm_eval(x, weight, bias, mean, var, eps):
    """nn.BatchNorm1d/2d in eval mode on (n, C) rows, fp32."""
    x = x.astype(np.float32)
    inv = (1.0 / np.sqrt(var.astype(np.float32) + np.float32(eps))).astype(np.float32)
    return ((x - mean.astype(np.float32)) * inv * weight.astype(np.float32) + bias.astype(np.float32)).astype(
        np.float32)


# ------------------------------------------------------------------------------------------------
# MinkowskiEngine semantics  [dep-knowledge, PARITY UNPINNED]
# ------------------------------------------------------------------------------------------------
def me_quantize(points4, quantization):
    """motionnet.py:22-36: coords = floor(fp32 points / fp32 quantization) -> unique voxel set.
    points4 (N,4) float32 [x,y,z,t]; returns coords (V,4) int32 in canonical key4 order,
    keys (V,) uint64, inverse (N,) int32 (point -> voxel row)."""
    q = np.asarray(quantization, dtype=np.float32)
    c = np.floor(points4.astype(np.float32) / q).astype(np.int32)
    k = key4(c)
    if np.any(k == INVALID_KEY):
        raise ValueError("coordinate outside the +-32768-voxel key window")
    uk, first, inverse = np.unique(k, return_index=True, return_inverse=True)
    return c[first], uk, inverse.astype(np.int32)


def me_feature_average(features, inverse, n_vox):
    """TensorField.sparse() UNWEIGHTED_AVERAGE of point features per voxel (motionnet.py:34-36)."""
    s = np.zeros((n_vox, features.shape[1]), dtype=np.float32)
    np.add.at(s, inverse, features.astype(np.float32))
    cnt = np.bincount(inverse, minlength=n_vox).astype(np.float32)
    return s / cnt[:, None]


def me_stride_down(coords, keys, level_shift):
    """Coordinate map of tensor_stride 2^level_shift in x,y,z (time untouched, minkunet.py:43-44):
    floor(c / s) * s, unique.  Returns coarse coords, keys and fine->coarse parent index."""
    s = 1 << level_shift
    pc = coords.copy()
    pc[:, :3] = (coords[:, :3] >> level_shift) << level_shift  # arithmetic shift == floor division
    mask = ~np.uint64((1 << (3 * level_shift)) - 1)
    pk = keys & mask
    uk, first, parent = np.unique(pk, return_index=True, return_inverse=True)
    assert s >= 1
    return pc[first], uk, parent.astype(np.int32)


def me_kernel_offsets(kernel_size, tensor_stride):
    """ME kernel-region enumeration: first (x) dimension fastest; odd sizes centred, even sizes
    start at 0; offsets scaled by the tensor stride of the *input* map (minkunet.py:55-124)."""
    ks = list(kernel_size)
    ts = list(tensor_stride)
    offs = []
    for it in range(ks[3]):
        for iz in range(ks[2]):
            for iy in range(ks[1]):
                for ix in range(ks[0]):
                    idx = [ix, iy, iz, it]
                    o = []
                    for d in range(4):
                        if ks[d] % 2 == 1:
                            o.append((idx[d] - (ks[d] - 1) // 2) * ts[d])
                        else:
                            o.append(idx[d] * ts[d])
                    offs.append(o)
    return np.array(offs, dtype=np.int32)


def me_nbr(out_coords, in_keys, offsets, sign=1):
    """nbr[k,o] = row of in-coordinate (out + sign*offset_k).  sign=-1 gives the transposed map:
    out fine voxel f reads coarse parent p with f = p + offset_k  <=>  p = f - offset_k."""
    K = len(offsets)
    q = np.empty((K, len(out_coords)), dtype=np.uint64)
    for k in range(K):
        q[k] = key4(out_coords.astype(np.int64) + sign * offsets[k].astype(np.int64))
    return nbr_lookup(q, in_keys)


# ------------------------------------------------------------------------------------------------
# spconv semantics  [dep-knowledge, PARITY UNPINNED]
# ------------------------------------------------------------------------------------------------
def voxelize_with_id(points, voxel_size, pc_range, max_voxels, max_points):
    """spconv PointToVoxel.generate_voxel_with_id as called at voxel_generate.py:19-28 (CPU
    semantics: first-come voxel ids, first max_points points kept per voxel).
    points (N,C) fp32; returns voxels (V,max_points,C), coords (V,3) int32 [z,y,x], num_points (V,)
    int32 (clamped to max_points), pc_voxel_id (N,) int64 (-1 = out of range or over the cap)."""
    pts = np.asarray(points, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    lo = np.asarray(pc_range[:3], dtype=np.float32)
    hi = np.asarray(pc_range[3:], dtype=np.float32)
    grid = np.round((hi - lo) / vs).astype(np.int64)  # [nx, ny, nz]
    c = np.floor((pts[:, :3] - lo) / vs)  # fp32 subtract, fp32 divide, floor
    inr = np.all((c >= 0) & (c < grid.astype(np.float32)), axis=1)
    ci = c.astype(np.int64)
    lin = (ci[:, 2] * grid[1] + ci[:, 1]) * grid[0] + ci[:, 0]
    idx = np.nonzero(inr)[0]
    ul, first, inv = np.unique(lin[idx], return_index=True, return_inverse=True)
    # first-seen order: rank voxels by the index of their first point
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    vid = rank[inv]
    pc_voxel_id = np.full(len(pts), -1, dtype=np.int64)
    pc_voxel_id[idx] = vid
    pc_voxel_id[pc_voxel_id >= max_voxels] = -1
    V = min(len(ul), max_voxels)
    voxels = np.zeros((V, max_points, pts.shape[1]), dtype=np.float32)
    num = np.zeros(V, dtype=np.int32)
    coords = np.zeros((V, 3), dtype=np.int32)
    fp = idx[first[order]][:V]
    coords[:, 0] = ci[fp, 2]
    coords[:, 1] = ci[fp, 1]
    coords[:, 2] = ci[fp, 0]
    for i in np.nonzero(pc_voxel_id >= 0)[0]:  # point order == first-come order
        v = pc_voxel_id[i]
        if num[v] < max_points:
            voxels[v, num[v]] = pts[i]
            num[v] += 1
    return voxels, coords, num, pc_voxel_id


def mean_vfe(voxels, num_points):
    """models/backbones_2d/mean_vfe.py:47-52  [pinned]."""
    s = voxels[:, 0, :].astype(np.float32).copy()
    for p in range(1, voxels.shape[1]):
        s = s + voxels[:, p, :]
    norm = np.maximum(num_points.astype(np.float32), np.float32(1.0))[:, None]
    return (s / norm).astype(np.float32)


def spconv_out_shape(in_shape, ksize, stride, pad):
    return [(int(in_shape[d]) + 2 * pad[d] - ksize[d]) // stride[d] + 1 for d in range(3)]


def spconv_kernel_offsets(ksize):
    """k = (kz*KH + ky)*KW + kx -- the order of spconv's weight[Cout,kz,ky,kx,Cin] taps."""
    return np.array([[kz, ky, kx] for kz in range(ksize[0]) for ky in range(ksize[1]) for kx in range(ksize[2])],
                    dtype=np.int32)


def spconv_down_coords(in_coords, in_shape, ksize, stride, pad):
    """SparseConv3d output coordinate set: o is active iff some tap k reads an active input,
    i = o*stride - pad + k.  Canonical order: ascending linear index."""
    out_shape = spconv_out_shape(in_shape, ksize, stride, pad)
    offs = spconv_kernel_offsets(ksize)
    st = np.array(stride)
    pd = np.array(pad)
    cands = []
    for k in offs:
        num = in_coords.astype(np.int64) + pd - k
        ok = np.all(num % st == 0, axis=1)
        o = num // st
        ok &= np.all((o >= 0) & (o < np.array(out_shape)), axis=1)
        cands.append(key3(o[ok], out_shape))
    uk = np.unique(np.concatenate(cands))
    H, W = out_shape[1], out_shape[2]
    ku = uk.astype(np.int64)
    coords = np.stack([ku // (H * W), (ku // W) % H, ku % W], 1).astype(np.int32)
    return coords, uk, out_shape


def spconv_nbr_subm(coords, keys_sorted, perm, shape, ksize=(3, 3, 3)):
    """SubMConv3d: out set == in set, tap k reads o + k - (ksize-1)/2."""
    offs = spconv_kernel_offsets(ksize)
    ctr = (np.array(ksize) - 1) // 2
    q = np.stack([key3(coords.astype(np.int64) + k - ctr, shape) for k in offs])
    return nbr_lookup(q, keys_sorted, perm)


def spconv_nbr_down(out_coords, in_keys_sorted, in_perm, in_shape, ksize, stride, pad):
    """SparseConv3d: tap k of output o reads input o*stride - pad + k."""
    offs = spconv_kernel_offsets(ksize)
    q = np.stack([key3(out_coords.astype(np.int64) * np.array(stride) - np.array(pad) + k, in_shape) for k in offs])
    return nbr_lookup(q, in_keys_sorted, in_perm)


def spconv_nbr_inverse(fine_coords, coarse_keys_sorted, coarse_perm, coarse_shape, ksize, stride, pad):
    """SparseInverseConv3d re-uses the forward pairs (i, o, k) reversed: fine voxel i receives
    coarse o through tap k iff i == o*stride - pad + k."""
    offs = spconv_kernel_offsets(ksize)
    st = np.array(stride)
    pd = np.array(pad)
    qs = []
    for k in offs:
        num = fine_coords.astype(np.int64) + pd - k
        ok = np.all(num % st == 0, axis=1)
        kk = key3(num // st, coarse_shape)
        qs.append(np.where(ok, kk, INVALID_KEY))
    return nbr_lookup(np.stack(qs), coarse_keys_sorted, coarse_perm)


def sorted_index(keys):
    """(sorted unique keys, perm) with perm[pos] = original row."""
    order = np.argsort(keys, kind="stable")
    return keys[order], order.astype(np.int32)


def sparse_to_dense_bev(feat, coords_zyx, shape):
    """HeightCompression (height_compression.py:24-31): .dense() -> (1, C*D, H, W), channel = c*D + d."""
    D, H, W = shape
    C = feat.shape[1]
    dense = np.zeros((C, D, H, W), dtype=np.float32)
    dense[:, coords_zyx[:, 0], coords_zyx[:, 1], coords_zyx[:, 2]] = feat.T
    return dense.reshape(1, C * D, H, W)


# ------------------------------------------------------------------------------------------------
# CenterHead decode / post-processing / instance features / metrics   [pinned]
# ------------------------------------------------------------------------------------------------
def center_decode(cls_hw, box_hw, out_size_factor, voxel_size_xy, pc_range_xy):
    """center_head.py:251-276 on NHWC maps cls (H,W,ncls), box (H,W,8) -> (H*W,ncls), (H*W,7).
    Uses torch CPU elementwise ops (exp / atan2) as the reference does."""
    import torch
    H, W, _ = box_hw.shape
    b = torch.from_numpy(np.ascontiguousarray(box_hw, dtype=np.float32)).reshape(H * W, 8)
    ys, xs = torch.meshgrid([torch.arange(0, H), torch.arange(0, W)], indexing="ij")
    xs = xs.reshape(-1, 1) + b[:, 0:1]
    ys = ys.reshape(-1, 1) + b[:, 1:2]
    xs = xs * out_size_factor * voxel_size_xy[0] + pc_range_xy[0]
    ys = ys * out_size_factor * voxel_size_xy[1] + pc_range_xy[1]
    rot = torch.atan2(b[:, 6:7], b[:, 7:8])
    boxes = torch.cat([xs, ys, b[:, 2:3], torch.exp(b[:, 3:6]), rot], dim=1)
    return np.ascontiguousarray(cls_hw, dtype=np.float32).reshape(H * W, -1), boxes.numpy().astype(np.float32)


def iou_bev_matrix(a, b):
    a = np.ascontiguousarray(a[:, :7], dtype=np.float32)
    b = np.ascontiguousarray(b[:, :7], dtype=np.float32)
    out = np.empty((len(a), len(b)), dtype=np.float32)
    _lib().co_iou_bev_matrix(_p(a), len(a), _p(b), len(b), _p(out))
    return out


def sparse_conv_backward(x, nbr, taps, dy):
    """Gradients of y[o] = sum_k x[nbr[k][o]] @ taps[k] (float64 accumulation): (dx, dtaps, dbias)."""
    x = np.asarray(x, np.float64)
    dy = np.asarray(dy, np.float64)
    taps = np.asarray(taps, np.float64)
    K = taps.shape[0]
    dx = np.zeros_like(x)
    dw = np.zeros_like(taps)
    for k in range(K):
        idx = np.arange(len(dy)) if nbr is None else np.asarray(nbr[k])
        o = np.nonzero(idx >= 0)[0]
        i = idx[o]
        np.add.at(dx, i, dy[o] @ taps[k].T)
        dw[k] = x[i].T @ dy[o]
    return dx, dw, dy.sum(0)


def mos_loss(logits, gt, n_classes=3, ignore_index=(0,)):
    """models/loss.py:20-34 in float64 numpy: (loss, d loss / d logits)."""
    z = np.asarray(logits, np.float64).copy()
    g = np.asarray(gt, np.int64)
    w = np.array([0.0 if c in ignore_index else 1.0 for c in range(n_classes)])
    w = w / w.sum()
    live = np.array([c not in ignore_index for c in range(n_classes)])
    zl = np.where(live[None, :], z, -np.inf)
    m = zl.max(1, keepdims=True)
    e = np.where(live[None, :], np.exp(zl - m), 0.0)
    p = e / e.sum(1, keepdims=True)
    pg = p[np.arange(len(g)), g]
    wg = w[g]
    loss = float((-wg * np.log(np.maximum(pg, 1e-8))).sum() / wg.sum())
    onehot = np.zeros_like(p)
    onehot[np.arange(len(g)), g] = 1.0
    grad = -(wg * (pg >= 1e-8))[:, None] * (onehot - p) * live[None, :] / wg.sum()
    return loss, grad


def gaussian_radius_f32(height, width, min_overlap):
    """center_head.py:395-424 on fp32 scalars, operation for operation (the reference evaluates it on 0-dim float32
    tensors; Python scalars are cast to float32 before each multiplication)."""
    f = np.float32
    h, w, ov = f(height), f(width), float(min_overlap)
    b1 = h + w
    c1 = w * h * f(1 - ov) / f(1 + ov)
    sq1 = np.sqrt(b1 * b1 - f(4) * c1)
    r1 = (b1 + sq1) / f(2)
    b2 = f(2) * (h + w)
    c2 = f(1 - ov) * w * h
    sq2 = np.sqrt(b2 * b2 - f(16) * c2)
    r2 = (b2 + sq2) / f(2)
    a3 = 4 * ov
    b3 = f(-2 * ov) * (h + w)
    c3 = f(ov - 1) * w * h
    sq3 = np.sqrt(b3 * b3 - f(4 * a3) * c3)
    r3 = (b3 + sq3) / f(2)
    return min(r1, r2, r3)


def center_assign_targets(gt_boxes8, grid_size, pc_range, voxel_size, out_size_factor, num_class, max_objs,
                          gaussian_overlap, min_radius):
    """CenterHead.get_targets_single (center_head.py:170-249) for one batch item, gt_boxes8 (M, 8) = box(7) + label:
    -> heatmap (C, H, W) fp32, anno_box (max_objs, 8) fp32, ind (max_objs) int64, mask (max_objs) uint8.
    Box sizes and the radius are float32 as in the reference (0-dim float32 tensors).  The cell coordinate
    (x - pc_range[0]) / voxel / factor is float32 when pc_range is an INTEGER array (the shipped config,
    config/config.yaml:6 -> np.array int64 -> torch int64) and float64 (then rounded to float32) when it holds floats:
    torch promotes a 0-dim float32 minus a 0-dim float64 to float64.  The gaussian is float64, then cast (:346-362)."""
    f = np.float32
    range_f64 = np.asarray(pc_range).dtype.kind == "f"
    gt = np.asarray(gt_boxes8, np.float32)
    Wf, Hf = int(grid_size[0]) // int(out_size_factor), int(grid_size[1]) // int(out_size_factor)
    vs = np.asarray(voxel_size, np.float32)
    pr = np.asarray(pc_range, np.float32)
    fac = f(out_size_factor)
    heat = np.zeros((num_class, Hf, Wf), np.float32)
    anno = np.zeros((max_objs, 8), np.float32)
    ind = np.zeros((max_objs,), np.int64)
    mask = np.zeros((max_objs,), np.uint8)
    with np.errstate(all="ignore"):
        for k in range(min(len(gt), max_objs)):
            cls_id = int(np.trunc(gt[k, 7] - f(1)))
            width = gt[k, 3] / vs[0] / fac
            length = gt[k, 4] / vs[1] / fac
            if not (width > 0 and length > 0 and cls_id > -1):
                continue
            radius = max(int(min_radius), int(gaussian_radius_f32(length, width, gaussian_overlap)))
            if range_f64:
                pr64 = np.asarray(pc_range, np.float64)
                cx = f((np.float64(gt[k, 0]) - pr64[0]) / np.float64(vs[0]) / np.float64(out_size_factor))
                cy = f((np.float64(gt[k, 1]) - pr64[1]) / np.float64(vs[1]) / np.float64(out_size_factor))
            else:
                cx = (gt[k, 0] - pr[0]) / vs[0] / fac
                cy = (gt[k, 1] - pr[1]) / vs[1] / fac
            x, y = int(np.trunc(cx)), int(np.trunc(cy))   # .to(torch.int32): truncation, so (-1, 0) lands in cell 0
            if not (0 <= x < Wf and 0 <= y < Hf):
                continue
            d = 2 * radius + 1
            sigma = d / 6
            yy, xx = np.ogrid[-radius:radius + 1, -radius:radius + 1]
            g = np.exp(-(xx * xx + yy * yy) / (2 * sigma * sigma))
            g[g < np.finfo(g.dtype).eps * g.max()] = 0
            left, right = min(x, radius), min(Wf - x, radius + 1)
            top, bottom = min(y, radius), min(Hf - y, radius + 1)
            hm = heat[cls_id, y - top:y + bottom, x - left:x + right]
            np.maximum(hm, g[radius - top:radius + bottom, radius - left:radius + right].astype(np.float32), out=hm)
            ind[k] = y * Wf + x
            mask[k] = 1
            anno[k] = [cx - f(x), cy - f(y), gt[k, 2], np.log(gt[k, 3]), np.log(gt[k, 4]), np.log(gt[k, 5]),
                       np.sin(gt[k, 6]), np.cos(gt[k, 6])]
    return heat, anno, ind, mask


def center_head_loss(cls_preds, box_preds, heatmap, anno_box, ind, mask, cls_weight=1.0, loc_weight=2.0,
                     code_weights=(1.0,) * 8):
    """CenterHead.get_loss (center_head.py:279-331) for one batch item in float64: cls_preds (H, W, C) raw logits,
    box_preds (H, W, 8) -> (loss_cls, loss_loc, d total / d cls_preds, d total / d box_preds).
    clip_sigmoid :333-344 (clamp 1e-4 .. 1 - 1e-4, zero gradient where clamped), gaussian_focal_loss :597-616
    (alpha 2, gamma 4, eps 1e-12, summed / max(#cells equal to 1, 1)), l1_loss :618-631 (sum / (#masked + 1e-4))."""
    z = np.asarray(cls_preds, np.float64)
    H, Wd, C = z.shape
    t = np.asarray(heatmap, np.float64).transpose(1, 2, 0)  # (H, W, C)
    sg = 1.0 / (1.0 + np.exp(-z))
    sg32 = sg.astype(np.float32).astype(np.float64)         # the clamp DECISION is taken on the fp32 sigmoid ...
    lo, hi = float(np.float32(1e-4)), float(np.float32(1 - 1e-4))
    inside = (sg32 >= lo) & (sg32 <= hi)
    p = np.where(inside, sg, np.clip(sg32, lo, hi))         # ... the value stays float64 where it is not clamped
    eps = 1e-12
    pos = (t == 1.0)
    negw = (1.0 - t) ** 4
    with np.errstate(all="ignore"):
        pos_loss = -np.log(p + eps) * (1 - p) ** 2 * pos
        neg_loss = -np.log(1 - p + eps) * p ** 2 * negw
        avg = max(float(pos.sum()), 1.0)
        loss_cls = float((pos_loss + neg_loss).sum() / avg) * cls_weight
        dpos = (-(1 - p) ** 2 / (p + eps) + 2 * (1 - p) * np.log(p + eps)) * pos
        dneg = (p ** 2 / (1 - p + eps) - 2 * p * np.log(1 - p + eps)) * negw
    g_cls = (dpos + dneg) * (p * (1 - p)) * inside * (cls_weight / avg)
    b = np.asarray(box_preds, np.float64).reshape(H * Wd, 8)
    tb = np.asarray(anno_box, np.float64)
    m = np.asarray(mask, np.float64)
    num = m.sum()
    wgt = m[:, None] * (~np.isnan(tb)) * np.asarray(code_weights, np.float64)[None, :]
    diff = b[np.asarray(ind, np.int64)] - tb
    loss_loc = float((np.abs(diff) * wgt).sum() / (num + 1e-4)) * loc_weight
    g_box = np.zeros_like(b)
    np.add.at(g_box, np.asarray(ind, np.int64), np.sign(diff) * wgt * (loc_weight / (num + 1e-4)))
    return loss_cls, loss_loc, g_cls, g_box.reshape(H, Wd, 8)


def overlap_bev_matrix(a, b):
    a = np.ascontiguousarray(a[:, :7], dtype=np.float32)
    b = np.ascontiguousarray(b[:, :7], dtype=np.float32)
    out = np.empty((len(a), len(b)), dtype=np.float32)
    _lib().co_overlap_bev_matrix(_p(a), len(a), _p(b), len(b), _p(out))
    return out


def iou3d_matrix(a, b):
    """iou3d_nms_utils.boxes_iou3d_gpu (models/bbox_post_process/iou3d_nms_utils.py:28-61), fp32 op for op."""
    a = np.ascontiguousarray(a[:, :7], dtype=np.float32)
    b = np.ascontiguousarray(b[:, :7], dtype=np.float32)
    two = np.float32(2)
    a_max, a_min = (a[:, 2] + a[:, 5] / two)[:, None], (a[:, 2] - a[:, 5] / two)[:, None]
    b_max, b_min = (b[:, 2] + b[:, 5] / two)[None, :], (b[:, 2] - b[:, 5] / two)[None, :]
    ov_h = np.maximum(np.minimum(a_max, b_max) - np.maximum(a_min, b_min), np.float32(0))
    ov3 = overlap_bev_matrix(a, b) * ov_h
    vol_a = (a[:, 3] * a[:, 4] * a[:, 5])[:, None]
    vol_b = (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return (ov3 / np.maximum(vol_a + vol_b - ov3, np.float32(1e-6))).astype(np.float32)


def generate_recall_record(box_preds, recall_dict, gt_boxes, thresh_list, rois=None):
    """models/post_process.py:67-110 for one batch item (gt_boxes: (G, >=7), trailing all-zero rows are padding)."""
    recall_dict = dict(recall_dict)
    if len(recall_dict) == 0:
        recall_dict = {"gt": 0}
        for t in thresh_list:
            recall_dict["roi_%s" % str(t)] = 0
            recall_dict["rcnn_%s" % str(t)] = 0
    gt = np.asarray(gt_boxes, dtype=np.float32)
    k = len(gt) - 1
    while k > 0 and gt[k].sum() == 0:
        k -= 1
    gt = gt[:k + 1]
    if gt.shape[0] > 0:
        iou_rcnn = iou3d_matrix(box_preds[:, :7], gt[:, :7]) if len(box_preds) > 0 else np.zeros((0, len(gt)), np.float32)
        iou_roi = iou3d_matrix(rois[:, :7], gt[:, :7]) if rois is not None else None
        for t in thresh_list:
            if iou_rcnn.shape[0] > 0:
                recall_dict["rcnn_%s" % str(t)] += int((iou_rcnn.max(0) > t).sum())
            if iou_roi is not None:
                recall_dict["roi_%s" % str(t)] += int((iou_roi.max(0) > t).sum())
        recall_dict["gt"] += int(gt.shape[0])
    return recall_dict


def nms_bev(boxes_sorted, thresh):
    b = np.ascontiguousarray(boxes_sorted[:, :7], dtype=np.float32)
    keep = np.empty(len(b), dtype=np.int64)
    n = _lib().co_nms_bev(_p(b), len(b), float(thresh), _p(keep))
    return keep[:n]


def post_process(cls_logits, boxes, score_thresh, nms_thresh, pre_max, post_max):
    """post_process.py:112-224 (class-agnostic branch) + :5-24 + iou3d_nms_utils.py:64-79.
    Tie rule of this build (torch.topk/sort leave ties unspecified): descending score, then
    ascending candidate index.  Returns (pred_boxes (K,7), pred_scores (K,), pred_labels (K,) int64,
    selected cell indices (K,))."""
    import torch
    prob = torch.sigmoid(torch.from_numpy(np.ascontiguousarray(cls_logits, dtype=np.float32))).numpy()
    label = np.argmax(prob, axis=1)  # first max on ties
    score = prob[np.arange(len(prob)), label]
    cand = np.nonzero(score >= np.float32(score_thresh))[0]
    if len(cand) == 0:
        z = np.zeros
        return z((0, 7), np.float32), z((0,), np.float32), z((0,), np.int64), z((0,), np.int64)
    order = np.lexsort((cand, -score[cand].astype(np.float64)))  # score desc, index asc
    order = order[:pre_max]
    sel = cand[order]
    keep = nms_bev(boxes[sel], nms_thresh)[:post_max]
    sel = sel[keep]
    return boxes[sel].astype(np.float32), score[sel].astype(np.float32), (label[sel] + 1).astype(np.int64), sel


def boxes_to_onehot(coords_xyz, boxes8, num_class, quirk=True):
    """Array_Index.find_features_by_bbox_with_yaw (Array_Index.cpp:14-79)."""
    c = np.ascontiguousarray(coords_xyz, dtype=np.int32)
    b = np.ascontiguousarray(boxes8, dtype=np.float32).reshape(-1, 8)
    feat = np.zeros((len(c), num_class), dtype=np.int32)
    _lib().co_boxes_to_onehot(_p(c), len(c), _p(b), len(b), _p(feat), num_class, 1 if quirk else 0)
    return feat


def points_in_instance_boxes(points, boxes8, num_class=3, out_ground=0.03, quirk=True):
    """Array_Index.find_point_in_instance_bbox_with_yaw (Array_Index.cpp:85-154): (N, num_class) int32, the value is
    the 1-based index of the box of that class containing the point (largest index wins, see coracle.c)."""
    p = np.ascontiguousarray(points, dtype=np.float32)
    b = np.ascontiguousarray(boxes8, dtype=np.float32).reshape(-1, 8)
    idx = np.zeros((len(p), num_class), dtype=np.int32)
    _lib().co_points_in_instance_boxes(_p(p), len(p), p.shape[1], _p(b), len(b), _p(idx), num_class,
                                       ctypes.c_float(out_ground), 1 if quirk else 0)
    return idx


def confusion_matrix(logits, gt, n_classes=3, ignore_index=(0,)):
    """models/metrics.py:16-30: mask ignored classes to -inf, argmax, histogram [pred, gt]."""
    lg = np.array(logits, dtype=np.float32, copy=True)
    lg[:, list(ignore_index)] = -np.inf
    pred = np.argmax(lg, axis=1)
    cm = np.zeros((n_classes, n_classes), dtype=np.int64)
    np.add.at(cm, (pred, np.asarray(gt, dtype=np.int64)), 1)
    return cm


def iou_from_confusion(cm, ignore_index=(0,)):
    """models/metrics.py:32-45."""
    cm = np.array(cm, dtype=np.int64, copy=True)
    cm[:, list(ignore_index)] = 0
    tp = np.diag(cm).astype(np.float64)
    fp = cm.sum(1) - tp
    fn = cm.sum(0) - tp
    return tp / (tp + fp + fn + 1e-15)


def output_stage(logits, ignore_index=(0,), learning_map_inv=None):
    """scripts/predict_mos.py:440-453: -inf mask, softmax, confidence = softmax[:,1:], argmax, remap."""
    import torch
    lg = np.array(logits, dtype=np.float32, copy=True)
    lg[:, list(ignore_index)] = -np.inf
    sm = torch.softmax(torch.from_numpy(lg), dim=1)
    pred = torch.argmax(sm, dim=1).numpy()
    inv = learning_map_inv or {0: 0, 1: 9, 2: 251}
    lut = np.zeros(max(inv) + 1, dtype=np.int32)
    for k, v in inv.items():
        lut[k] = v
    return lut[pred].astype(np.int32), sm[:, 1:].numpy()

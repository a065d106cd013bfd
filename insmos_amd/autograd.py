"""Training-step building blocks on MI355X (SURVEY.md 8f rank 2 -- a FIRST SLICE, see include/insmos_hip.h):

  * SparseConvFunction / sparse_conv(): the output-stationary sparse convolution as a torch.autograd.Function whose
    forward AND backward are HIP kernels -- d/dx is the same MFMA kernel on the transposed table with transposed taps,
    d/dW a slab-reduction kernel, d/db a column sum.  Deterministic: no atomics anywhere (the reference's libraries
    scatter-add their gradients with atomics).
  * mos_loss(): MOSLoss.compute_loss (models/loss.py:20-34) with its gradient from one kernel.

torch is used for what it is here for -- device memory, streams and the autograd tape that chains these nodes
(models/models.py:61-98 calls loss.backward() on such a tape).
  * batch_norm_train(): nn.BatchNorm1d / MinkowskiBatchNorm in train() mode (+ fused ReLU), forward and backward.
  * center_assign_targets() / center_head_loss(): CenterHead.assign_targets / get_loss (center_head.py:126-331).
The optimiser step stays torch.optim's.
"""
import contextlib
import ctypes
import os
import threading

import torch

from . import _lib
from .engine import _padc


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _packed(lib, taps, cin_pad, cout_store, transpose, mirror, st):
    """Device-packed MFMA A fragments of `taps` (K, cin_real, cout_real) (or of their transpose)."""
    K, ci, co = taps.shape
    cr_in, cr_out = (co, ci) if transpose else (ci, co)
    assert cin_pad >= cr_in and cout_store >= cr_out
    n = int(lib.insmos_packed_weight_floats(K, cin_pad, cout_store))
    packed = torch.empty(n, dtype=torch.float32, device=taps.device)
    _lib.check(lib.insmos_pack_weights_device(taps.data_ptr(), K, ci, co, cin_pad, cout_store, 1 if transpose else 0,
                                              1 if mirror else 0, packed.data_ptr(), st), "insmos_pack_weights_device")
    return packed


# Precision of the TRAINING convolutions (forward and d/dx; d/dW and every other kernel stay fp32).  0 = exact fp32 (default);
# 1 = operands rounded to bf16, fp32 accumulate (include/insmos_hip.h: insmos_conv_precision) -- opt-in PER TRAINER
# (InsMOSTrainer(bf16_convs=True) / INSMOS_TRAIN_BF16=1 wrap their step in train_conv_precision(1)); set_train_conv_precision()
# is the process default for bare sparse_conv() calls.  The mode of a node is fixed when its forward runs (kept in ctx: the
# backward, which autograd runs on its own thread, uses the same one) and reaches the library as a HOST-THREAD-local override
# around each launch (insmos_conv_precision_thread), so inference forwards on other host threads stay exact fp32.
_TRAIN_CONV_PRECISION = 0
_tls = threading.local()


def set_train_conv_precision(mode):
    global _TRAIN_CONV_PRECISION
    if mode not in (0, 1):
        raise ValueError("training conv precision: 0 (fp32) or 1 (bf16 operands, fp32 accumulate)")
    _TRAIN_CONV_PRECISION = int(mode)


@contextlib.contextmanager
def train_conv_precision(mode):
    """Precision of the sparse_conv() nodes CREATED inside the block on this thread (a trainer instance's own setting)."""
    if mode not in (0, 1):
        raise ValueError("training conv precision: 0 (fp32) or 1 (bf16 operands, fp32 accumulate)")
    prev = getattr(_tls, "mode", None)
    _tls.mode = int(mode)
    try:
        yield
    finally:
        _tls.mode = prev


def current_train_conv_precision():
    m = getattr(_tls, "mode", None)
    return _TRAIN_CONV_PRECISION if m is None else m


def _conv(lib, x, n_in, cin_pad, nbr, K, n_out, packed, bias_pad, cout, st, mask=None, mode=0):
    out = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    # ALWAYS pinned for this host thread, mode 0 included: a node recorded as exact fp32 must not inherit a process-wide
    # insmos_conv_precision(1 / 3) an inference engine may have set meanwhile
    _lib.check(lib.insmos_conv_precision_thread(int(mode)), "insmos_conv_precision_thread")
    try:
        _lib.check(lib.insmos_sparse_conv(x.data_ptr(), n_in, x.stride(0), cin_pad, nbr.data_ptr() if nbr is not None else None,
                                          mask.data_ptr() if (mask is not None and nbr is not None) else None, K, n_out,
                                          packed.data_ptr(), bias_pad.data_ptr(), out.data_ptr(), cout, cout, None,
                                          0, 0, 0, 0, st), "insmos_sparse_conv")
    finally:
        lib.insmos_conv_precision_thread(-1)
    return out


_zero_vecs = {}


def _zeros(n, device):
    """A cached all-zero fp32 vector (read-only by contract: bias of a bias-free layer) -- one fill per size instead of one per node."""
    key = (int(n), str(device))
    z = _zero_vecs.get(key)
    if z is None:
        z = _zero_vecs[key] = torch.zeros(int(n), dtype=torch.float32, device=device)
    return z


def _pad_cols(x, c):
    """Rows padded to the channel widths the conv kernel accepts (4, 8, multiple of 16); zero columns are neutral."""
    if x.shape[1] == c and x.stride(1) == 1 and x.stride(0) % 4 == 0:
        return x
    y = torch.zeros((x.shape[0], c), dtype=torch.float32, device=x.device)
    y[:, :x.shape[1]] = x
    return y


# bench.py --config cfg5: executed flops of the convolution kernels of a training step (forward, d/dx, d/dW: 2 * pairs * Cin * Cout
# each, pairs = valid entries of the layer's kernel map).  None = off (the default: counting costs a device reduction per table).
WORK_COUNTER = None


def _count_work(kind, nbr, n_rows, cin, cout):
    if WORK_COUNTER is None:
        return
    if nbr is None:
        pairs = int(n_rows)
    else:
        # cached ON the table object (an address + shape key would be handed to another table by the caching allocator on the
        # next step and silently reuse a stale count; an attribute dies with its tensor)
        pairs = getattr(nbr, "_insmos_valid_pairs", None)
        if pairs is None:
            pairs = int((nbr >= 0).sum().item())
            try:
                nbr._insmos_valid_pairs = pairs
            except AttributeError:
                pass
    WORK_COUNTER[kind] = WORK_COUNTER.get(kind, 0) + 2 * pairs * cin * cout
    WORK_COUNTER["launches_" + kind] = WORK_COUNTER.get("launches_" + kind, 0) + 1


class SparseConvFunction(torch.autograd.Function):
    """y[o] = sum_k x[nbr[k][o]] @ taps[k] + bias.  nbr (K, n_out) int32 (-1 = no neighbour); nbr_t (K, n_in) int32 is the
    TRANSPOSED table (nbr_t[k][i] = o  <=>  nbr[k][o] = i); for a submanifold layer pass nbr_t=None and the layer's own
    table is used with mirrored taps (nbr_t[k] = nbr[K-1-k]).  mask / mask_t: the tables' active-tap bitmasks per 16-row group
    (what insmos_build_nbr emits; optional): the kernels then walk only the taps a tile has -- half of the (tile, tap) slots of
    a LiDAR table are empty.  A mask belongs to a table ROW set, so the mirrored-tap d/dx of a submanifold layer uses `mask`."""

    @staticmethod
    def forward(ctx, x, taps, bias, nbr, nbr_t, mask=None, mask_t=None):
        lib = _lib.load()
        st = _stream(x.device)
        K, cin, cout = taps.shape
        n_in, n_out = x.shape[0], (nbr.shape[1] if nbr is not None else x.shape[0])
        taps = taps.contiguous().float()
        cin_pad = _padc(cin)
        xp = _pad_cols(x.float(), cin_pad)
        if bias is not None:
            bias_pad = torch.zeros((cout + 15) // 16 * 16, dtype=torch.float32, device=x.device)
            bias_pad[:cout] = bias
        else:
            bias_pad = _zeros((cout + 15) // 16 * 16, x.device)
        ctx.prec = current_train_conv_precision()
        y = _conv(lib, xp, n_in, cin_pad, nbr, K, n_out, _packed(lib, taps, cin_pad, cout, False, False, st), bias_pad, cout, st,
                  mask, mode=ctx.prec)
        _count_work("forward", nbr, n_out, cin, cout)
        ctx.masks = (mask, mask_t)
        ctx.save_for_backward(xp, taps, nbr if nbr is not None else torch.empty(0), nbr_t if nbr_t is not None else torch.empty(0))
        ctx.meta = (K, cin, cout, n_in, n_out, nbr is not None, nbr_t is not None, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xp, taps, nbr, nbr_t = ctx.saved_tensors
        K, cin, cout, n_in, n_out, has_nbr, has_t, has_bias = ctx.meta
        mask, mask_t = ctx.masks
        st = _stream(dy.device)
        dy = dy.contiguous().float()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            _count_work("dx", nbr if has_nbr else None, n_out, cin, cout)
            # dx[i] = sum_k dy[nbr_t[k][i]] @ taps[k]^T : the forward kernel, transposed taps, transposed table
            cp = _padc(cout)
            dyp = _pad_cols(dy, cp)
            zero_b = _zeros((cin + 15) // 16 * 16, dy.device)
            if not has_nbr:      # 1x1 / Linear
                dx = _conv(lib, dyp, n_out, cp, None, K, n_in, _packed(lib, taps, cp, cin, True, False, st), zero_b, cin, st, mode=ctx.prec)
            elif has_t:
                dx = _conv(lib, dyp, n_out, cp, nbr_t, K, n_in, _packed(lib, taps, cp, cin, True, False, st), zero_b, cin, st, mask_t, mode=ctx.prec)
            else:                # submanifold: the layer's own table, taps mirrored
                assert n_in == n_out
                dx = _conv(lib, dyp, n_out, cp, nbr, K, n_in, _packed(lib, taps, cp, cin, True, True, st), zero_b, cin, st, mask, mode=ctx.prec)
        if ctx.needs_input_grad[1]:
            _count_work("dw", nbr if has_nbr else None, n_out, cin, cout)
            dw = torch.empty((K, cin, cout), dtype=torch.float32, device=dy.device)
            ws = torch.empty(int(lib.insmos_sparse_conv_backward_weight_ws_floats(n_out, K, cin, cout)), dtype=torch.float32,
                             device=dy.device)
            _lib.check(lib.insmos_sparse_conv_backward_weight(xp.data_ptr(), n_in, xp.stride(0), cin, dy.data_ptr(), dy.stride(0),
                                                              cout, nbr.data_ptr() if has_nbr else None, K, n_out, dw.data_ptr(),
                                                              0, ws.data_ptr(), st), "insmos_sparse_conv_backward_weight")
        if has_bias and ctx.needs_input_grad[2]:
            db = torch.empty((cout,), dtype=torch.float32, device=dy.device)
            ws = torch.empty(int(lib.insmos_col_sum_ws_floats(n_out, cout)) + 1, dtype=torch.float32, device=dy.device)
            _lib.check(lib.insmos_col_sum(dy.data_ptr(), dy.stride(0), cout, n_out, db.data_ptr(), 0, ws.data_ptr(), st),
                       "insmos_col_sum")
        return dx, dw, db, None, None, None, None


def sparse_conv(x, taps, bias, nbr, nbr_t=None):
    """nbr / nbr_t: (K, n) int32 tensors, or the engine's NbrTable objects (table + active-tap masks: the masks are used)."""
    mask, mask_t = getattr(nbr, "mask16", None), getattr(nbr_t, "mask16", None)
    nbr, nbr_t = getattr(nbr, "nbr", nbr), getattr(nbr_t, "nbr", nbr_t)
    return SparseConvFunction.apply(x, taps, bias, nbr, nbr_t, mask, mask_t)


class BatchNormTrainFunction(torch.autograd.Function):
    """nn.BatchNorm1d over rows in training mode (+ optional fused ReLU): batch statistics, HIP forward and backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu):
        lib = _lib.load()
        st = _stream(x.device)
        x = x.float()
        if x.stride(1) != 1:
            x = x.contiguous()
        n, c = x.shape
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        xhat = torch.empty((n, c), dtype=torch.float32, device=x.device)
        stats = torch.empty(3 * c, dtype=torch.float32, device=x.device)
        ws = torch.empty(int(lib.insmos_batchnorm_ws_floats(n, c)), dtype=torch.float32, device=x.device)
        g32, b32 = gamma.contiguous().float(), beta.contiguous().float()
        _lib.check(lib.insmos_batchnorm_train_forward(x.data_ptr(), x.stride(0), c, n, g32.data_ptr(), b32.data_ptr(), float(eps),
                                                      1 if relu else 0, y.data_ptr(), c, xhat.data_ptr(), stats.data_ptr(),
                                                      ws.data_ptr(), st), "insmos_batchnorm_train_forward")
        ctx.save_for_backward(y, xhat, g32, stats)
        ctx.relu = bool(relu)
        ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        lib = _lib.load()
        y, xhat, gamma, stats = ctx.saved_tensors
        st = _stream(dy.device)
        dy = dy.contiguous().float()
        n, c = dy.shape
        dx = torch.empty((n, c), dtype=torch.float32, device=dy.device)
        dgamma = torch.empty(c, dtype=torch.float32, device=dy.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=dy.device)
        gws = torch.empty((n, c), dtype=torch.float32, device=dy.device)
        ws = torch.empty(int(lib.insmos_batchnorm_ws_floats(n, c)), dtype=torch.float32, device=dy.device)
        _lib.check(lib.insmos_batchnorm_train_backward(dy.data_ptr(), c, y.data_ptr(), c, xhat.data_ptr(), c, n, gamma.data_ptr(),
                                                       stats.data_ptr(), 1 if ctx.relu else 0, dx.data_ptr(), c, dgamma.data_ptr(),
                                                       dbeta.data_ptr(), gws.data_ptr(), ws.data_ptr(), st),
                   "insmos_batchnorm_train_backward")
        return dx, dgamma, dbeta, None, None


def batch_norm_train(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, relu=False):
    """F.batch_norm(x, running_mean, running_var, gamma, beta, training=True, momentum, eps) (+ ReLU) on the rows of x;
    the running statistics are updated in place like torch does (unbiased variance)."""
    y, stats = BatchNormTrainFunction.apply(x, gamma, beta, eps, relu)
    if running_mean is not None:
        n, c = x.shape
        with torch.no_grad():
            running_mean.mul_(1 - momentum).add_(stats[:c], alpha=momentum)
            running_var.mul_(1 - momentum).add_(stats[2 * c:] * (n / max(n - 1, 1)), alpha=momentum)
    return y


class BnPlan:
    """Which rows a BatchNorm layer normalises together (insmos_batchnorm_seg_*, csrc/train.hip): S segments -- the windows of a
    training batch, each normalised on its own like the reference's item-by-item forwards (models/models.py:313) -- as a table
    of <= 1024-row chunks of one segment each, sorted by segment.  Built once per coordinate level and step, shared by every
    layer on that level."""

    CHUNK = 1024   # (upper bound of the adaptive chunk length below when INSMOS_BN_CHUNK pins it)

    @staticmethod
    def chunk_rows(c=8):
        """Rows per chunk for a layer of c channels (one table per chunk length, `table(c)`): about 64 K ELEMENTS per block -- 4096 rows
        at c <= 16, 2048 at 32, 1024 at 64, 512 from 128 on.  Measured per shape (tools/bn_shape_probe.py, forward + backward of one
        layer, us): 1.9 M x 8: 476 at 1024 rows, 282 at 4096; 850 k x 16: 330 / 265; 335 k x 32: 234 at 1024, 226 at 2048, 309 at
        4096; 75 k x 128: 294 at 1024, 242 at 512, 790 at 4096.  (One length for every width cannot win: 4096 everywhere made the step's
        BatchNorm 23.8 ms against 12.6 at 1024, SHORTER chunks for wider layers 26.2.)  INSMOS_BN_CHUNK pins one length."""
        fixed = os.environ.get("INSMOS_BN_CHUNK")
        if fixed:
            return max(16, int(fixed))
        rows = 65536 // max(int(c), 1)
        p2 = 1
        while p2 * 2 <= rows:
            p2 *= 2
        return int(min(4096, max(512, p2)))

    def __init__(self, runs, n_rows, n_seg, device):
        """runs: iterable of (row_start, row_end, segment) covering [0, n_rows) (any order)."""
        import numpy as np
        self._runs = [(int(r0), int(r1), int(sg)) for r0, r1, sg in runs]
        self.n_rows, self.S, self.device = int(n_rows), int(n_seg), device
        rows = np.zeros(n_seg, np.int32)
        for r0, r1, sg in self._runs:
            rows[sg] += r1 - r0
        assert int(rows.sum()) == int(n_rows), (int(rows.sum()), n_rows)
        self.seg_rows_host = rows
        self.seg_rows = torch.from_numpy(rows).to(device)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=device)   # the kernels' last-block counter (left zero by each launch)
        self._tables = {}
        t = self.table(8)
        self.chunks, self.seg_first, self.n_chunks = t.chunks, t.seg_first, t.n_chunks   # (the c = 8 table: the historical attributes)

    def table(self, c):
        """The chunk table for layers of c channels (built on first use, shared by every such layer on this level)."""
        import numpy as np
        step = self.chunk_rows(c)
        if step not in self._tables:
            ch = []
            for r0, r1, sg in self._runs:
                for a in range(r0, r1, step):
                    ch.append((a, min(a + step, r1), sg, 0))
            ch = np.asarray(ch, np.int32).reshape(-1, 4)
            ch = ch[np.argsort(ch[:, 2], kind="stable")]
            first = np.searchsorted(ch[:, 2], np.arange(self.S + 1)).astype(np.int32)
            t = type("BnTable", (), {})()
            t.n_chunks = int(len(ch))
            t.chunks = torch.from_numpy(np.ascontiguousarray(ch)).to(self.device)
            t.seg_first = torch.from_numpy(first).to(self.device)
            self._tables[step] = t
        return self._tables[step]

    @classmethod
    def whole(cls, n_rows, device):
        return cls([(0, n_rows, 0)], n_rows, 1, device)

    @classmethod
    def from_segment_ids(cls, seg, n_seg):
        """seg: (n,) integer device tensor, the segment of every row (runs of equal ids become the chunk runs)."""
        n = int(seg.shape[0])
        seg = seg.contiguous()
        cut = (torch.nonzero(seg[1:] != seg[:-1]).flatten() + 1).cpu().numpy() if n > 1 else []
        starts = [0] + [int(v) for v in cut]
        ends = [int(v) for v in cut] + [n]
        ids = seg[torch.as_tensor(starts, device=seg.device)].cpu().numpy()
        return cls(zip(starts, ends, ids), n, n_seg, seg.device)


# INSMOS_BN_RECOMPUTE=0: the segmented BatchNorm stores x^ again (12 passes over a layer's elements per step instead of 9)
BN_RECOMPUTE = os.environ.get("INSMOS_BN_RECOMPUTE", "1") != "0"


class BatchNormSegFunction(torch.autograd.Function):
    """BatchNorm1d in training mode with per-segment statistics (BnPlan), fused ReLU; forward and backward are HIP kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, plan, running_mean, running_var, momentum):
        lib = _lib.load()
        st = _stream(x.device)
        x = x.float()
        if x.stride(1) != 1:
            x = x.contiguous()
        n, c = x.shape
        if n != plan.n_rows:
            raise ValueError(f"BatchNorm plan covers {plan.n_rows} rows, the input has {n}")
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        # x^ is not stored when the 16-byte kernels apply: x is kept instead and the backward recomputes x^ and the ReLU mask from it
        # (the same bits; one store pass and two read passes over the layer's elements fewer -- csrc/train.hip RECOMP)
        recomp = (BN_RECOMPUTE and bool(lib.insmos_batchnorm_seg_recompute_ok(c, x.stride(0), c, c)) and x.data_ptr() % 16 == 0)
        xhat = None if recomp else torch.empty((n, c), dtype=torch.float32, device=x.device)
        stats = torch.empty(plan.S * 3 * c, dtype=torch.float32, device=x.device)
        tab = plan.table(c)
        ws = torch.empty(int(lib.insmos_batchnorm_seg_ws_floats(tab.n_chunks, c, plan.S)), dtype=torch.float32, device=x.device)
        g32, b32 = gamma.contiguous().float(), beta.contiguous().float()
        _lib.check(lib.insmos_batchnorm_seg_forward(
            x.data_ptr(), x.stride(0), c, n, tab.chunks.data_ptr(), tab.n_chunks, tab.seg_first.data_ptr(), plan.seg_rows.data_ptr(),
            plan.S, g32.data_ptr(), b32.data_ptr(), float(eps), 1 if relu else 0, y.data_ptr(), c,
            xhat.data_ptr() if xhat is not None else None, stats.data_ptr(),
            running_mean.data_ptr() if running_mean is not None else None, running_var.data_ptr() if running_var is not None else None,
            float(momentum), plan.ticket.data_ptr(), ws.data_ptr(), st), "insmos_batchnorm_seg_forward")
        if recomp:
            ctx.save_for_backward(x, g32, b32, stats)
        else:
            ctx.save_for_backward(y, xhat, g32, stats)
        ctx.relu, ctx.plan, ctx.recomp = bool(relu), plan, recomp
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        plan = ctx.plan
        st = _stream(dy.device)
        dy = dy.contiguous().float()
        n, c = dy.shape
        dx = torch.empty((n, c), dtype=torch.float32, device=dy.device)
        dgamma = torch.empty(c, dtype=torch.float32, device=dy.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=dy.device)
        tab = plan.table(c)
        ws = torch.empty(int(lib.insmos_batchnorm_seg_ws_floats(tab.n_chunks, c, plan.S)), dtype=torch.float32, device=dy.device)
        if ctx.recomp:   # (dy is a contiguous (n, c) fp32 matrix with c % 4 == 0: rows start on the 16-byte grid)
            x, gamma, beta, stats = ctx.saved_tensors
            _lib.check(lib.insmos_batchnorm_seg_backward_x(
                dy.data_ptr(), c, x.data_ptr(), x.stride(0), c, n, tab.chunks.data_ptr(), tab.n_chunks, tab.seg_first.data_ptr(),
                plan.seg_rows.data_ptr(), plan.S, gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(), 1 if ctx.relu else 0,
                dx.data_ptr(), c, dgamma.data_ptr(), dbeta.data_ptr(), plan.ticket.data_ptr(), ws.data_ptr(), st),
                "insmos_batchnorm_seg_backward_x")
            return dx, dgamma, dbeta, None, None, None, None, None, None
        y, xhat, gamma, stats = ctx.saved_tensors
        _lib.check(lib.insmos_batchnorm_seg_backward(
            dy.data_ptr(), c, y.data_ptr(), c, xhat.data_ptr(), c, n, tab.chunks.data_ptr(), tab.n_chunks, tab.seg_first.data_ptr(),
            plan.seg_rows.data_ptr(), plan.S, gamma.data_ptr(), stats.data_ptr(), 1 if ctx.relu else 0, dx.data_ptr(), c,
            dgamma.data_ptr(), dbeta.data_ptr(), plan.ticket.data_ptr(), ws.data_ptr(), st), "insmos_batchnorm_seg_backward")
        return dx, dgamma, dbeta, None, None, None, None, None, None


def batch_norm_train_seg(x, gamma, beta, plan, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, relu=False,
                         force_segmented=False):
    """batch_norm_train with the statistics taken per segment of `plan` (one window of a training batch each); the running
    statistics move segment after segment, as the reference's item-by-item forwards move them.  A single segment runs on the
    first BatchNorm kernels (batch_norm_train: the same layer; measured 9.1 vs 11.3 ms per one-window step, and their rounding is
    the one tests/golden/train_wiring.npz -- the reference's own training code -- was matched to 1e-4 with) unless
    force_segmented (tests)."""
    if plan.S == 1 and not force_segmented:
        return batch_norm_train(x, gamma, beta, running_mean, running_var, momentum, eps, relu)
    return BatchNormSegFunction.apply(x, gamma, beta, eps, relu, plan, running_mean, running_var, momentum)


class GatherRowsFunction(torch.autograd.Function):
    """y = x[index] for an (n, C) matrix and an int64 index whose rows may repeat (the point -> voxel gather of the two heads:
    spconv_unet.py:408-410, motionnet.py:42-46).  torch's backward of advanced indexing sorts the index on every call
    (indexing_backward_kernel: 0.9 ms per call on the 480 k points of a B = 4 step, profiles/r05_cfg5_host_profile.txt).  The
    gradient of a gather is one scatter-add; to keep it DETERMINISTIC (the data-parallel tests compare two backward passes bit
    for bit) the rows are added as 2^-40 fixed-point int64 -- integer atomics commute -- and converted back: the sum is exact to
    9e-13 absolute, |row gradient sums| < 8e6."""
    SCALE = float(1 << 40)

    @staticmethod
    def forward(ctx, x, index):
        ctx.save_for_backward(index)
        ctx.n = x.shape[0]
        return x[index]

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        q = (g.double() * GatherRowsFunction.SCALE).round_().long()
        acc = torch.zeros((ctx.n,) + tuple(g.shape[1:]), dtype=torch.int64, device=g.device).index_add_(0, index, q)
        return (acc.double() / GatherRowsFunction.SCALE).to(g.dtype), None


def gather_rows(x, index):
    return GatherRowsFunction.apply(x, index)


class MosLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, gt, class_weights, ignore_mask):
        lib = _lib.load()
        st = _stream(logits.device)
        lg = logits.contiguous().float()
        n, ncls = lg.shape
        # converted views are bound to locals: a temporary would be released before the launch that reads its memory
        gt_l = gt.contiguous().long()
        cw = class_weights.contiguous().float()
        if gt_l.numel() != n:
            raise ValueError(f"mos_loss: {gt_l.numel()} labels for {n} logit rows")
        if gt_l.device != lg.device or cw.device != lg.device:
            raise ValueError("mos_loss: logits, labels and class weights must live on the same device")
        sums = torch.empty(2, dtype=torch.float32, device=lg.device)
        grad = torch.empty((n, ncls), dtype=torch.float32, device=lg.device)
        ws = torch.empty(int(lib.insmos_mos_loss_ws_floats(n)), dtype=torch.float32, device=lg.device)
        _lib.check(lib.insmos_mos_loss(lg.data_ptr(), lg.stride(0), gt_l.data_ptr(), n, ncls, int(ignore_mask),
                                       cw.data_ptr(), sums.data_ptr(), grad.data_ptr(), ncls,
                                       ws.data_ptr(), st), "insmos_mos_loss")
        ctx.save_for_backward(grad)
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


def mos_loss(logits, gt_labels, n_classes=3, ignore_index=(0,)):
    """MOSLoss(n_classes, ignore_index).compute_loss(logits, gt_labels) (models/loss.py:10-34), differentiable."""
    w = [0.0 if i in ignore_index else 1.0 for i in range(n_classes)]
    w = torch.tensor([v / sum(w) for v in w], dtype=torch.float32, device=logits.device)
    mask = 0
    for c in ignore_index:
        mask |= 1 << int(c)
    return MosLossFunction.apply(logits, gt_labels, w, mask)


# ---------------------------------------------------------------------------------------------------------------------
# CenterHead training side (models/backbones_2d/center_head.py:126-331)
# ---------------------------------------------------------------------------------------------------------------------
def center_assign_targets(gt_boxes, head_cfg, grid_size, point_cloud_range, num_class=3):
    """CenterHead.assign_targets(gt_boxes) (center_head.py:126-168): gt_boxes (B, M, 8) = box(7) + label on the device ->
    {'heatmaps': [(B, C, H, W)], 'anno_boxes': [(B, max_objs, 8)], 'inds': [(B, max_objs) int64],
     'masks': [(B, max_objs) uint8]} -- one task, like the reference.  One launch per batch item, no host round trips
    (the reference loops over the boxes on the host, center_head.py:202-243)."""
    import numpy as np
    lib = _lib.load()
    tc = head_cfg["TARGET_ASSIGNER_CONFIG"]
    factor = int(tc["OUT_SIZE_FACTOR"])
    fm_w, fm_h = int(grid_size[0]) // factor, int(grid_size[1]) // factor
    max_objs = int(tc["MAX_OBJS"])
    rng = np.asarray(point_cloud_range)
    gt = gt_boxes.contiguous().float()
    B, M = int(gt.shape[0]), int(gt.shape[1])
    dev = gt.device
    heat = torch.empty((B, num_class, fm_h, fm_w), dtype=torch.float32, device=dev)
    anno = torch.empty((B, max_objs, 8), dtype=torch.float32, device=dev)
    ind = torch.empty((B, max_objs), dtype=torch.int64, device=dev)
    mask = torch.empty((B, max_objs), dtype=torch.uint8, device=dev)
    st = _stream(dev)
    for b in range(B):
        _lib.check(lib.insmos_center_assign_targets(
            gt[b].data_ptr(), M, max_objs, num_class, fm_w, fm_h, float(rng[0]), float(rng[1]),
            1 if rng.dtype.kind == "f" else 0, float(np.float32(tc["VOXEL_SIZE"][0])), float(np.float32(tc["VOXEL_SIZE"][1])),
            factor, float(tc["GAUSSIAN_OVERLAP"]), int(tc["MIN_RADIUS"]), heat[b].data_ptr(), anno[b].data_ptr(),
            ind[b].data_ptr(), mask[b].data_ptr(), st), "insmos_center_assign_targets")
    return {"heatmaps": [heat], "anno_boxes": [anno], "inds": [ind], "masks": [mask]}


class CenterHeadLossFunction(torch.autograd.Function):
    """(cls_preds (H*W, C) logits, box_preds (H*W, 8)) of ONE batch item -> losses (3,) = [cls, loc, total]."""

    @staticmethod
    def forward(ctx, cls_preds, box_preds, heatmap, anno_box, ind, mask, cls_weight, loc_weight, code_weights):
        lib = _lib.load()
        st = _stream(cls_preds.device)
        cp, bp = cls_preds.contiguous().float(), box_preds.contiguous().float()
        hw, nc = int(cp.shape[0]), int(cp.shape[1])
        max_objs = int(ind.shape[0])
        losses = torch.empty(3, dtype=torch.float32, device=cp.device)
        g_cls = torch.empty_like(cp)
        g_box = torch.empty_like(bp)
        ws = torch.empty(int(lib.insmos_center_head_loss_ws_floats(hw, nc)), dtype=torch.float32, device=cp.device)
        cw = (ctypes.c_float * 8)(*[float(v) for v in code_weights])
        hm, ab, ix, mk = heatmap.contiguous(), anno_box.contiguous(), ind.contiguous(), mask.contiguous()  # alive across the launch
        _lib.check(lib.insmos_center_head_loss(cp.data_ptr(), nc, bp.data_ptr(), 8, hw, nc, hm.data_ptr(),
                                               ab.data_ptr(), ix.data_ptr(),
                                               mk.data_ptr(), max_objs, float(cls_weight), float(loc_weight),
                                               ctypes.cast(cw, ctypes.c_void_p), losses.data_ptr(), g_cls.data_ptr(), nc,
                                               g_box.data_ptr(), 8, ws.data_ptr(), st), "insmos_center_head_loss")
        ctx.save_for_backward(g_cls, g_box)
        # the stored gradients are those of the TOTAL (losses[2]); the two parts are reported, not differentiated
        parts = losses[:2].detach()
        ctx.mark_non_differentiable(parts)
        return parts, losses[2]

    @staticmethod
    def backward(ctx, g_parts, g_total):
        g_cls, g_box = ctx.saved_tensors
        # total = cls + loc is what get_loss() returns and the training step differentiates (center_head.py:283)
        return g_cls * g_total, g_box * g_total, None, None, None, None, None, None, None


def center_head_loss(cls_preds, box_preds, targets, head_cfg, as_tensors=False):
    """CenterHead.get_loss() (center_head.py:279-288) for NHWC maps cls_preds (B, H, W, C) / box_preds (B, H, W, 8) and the
    dict assign_targets returned -> (rpn_loss, tb_dict) with the reference's keys.  Differentiable in both maps
    (through rpn_loss).  B = 1 is what the reference trains with per model call (models/models.py:313 walks the list);
    for B > 1 the reference's batch-wide averages are reproduced only for B = 1, so larger batches are rejected."""
    B = int(cls_preds.shape[0])
    if B != 1:
        raise ValueError("center_head_loss: one batch item per call (the reference's model loop feeds B = 1)")
    lw = head_cfg["LOSS_CONFIG"]["LOSS_WEIGHTS"]
    nc = int(cls_preds.shape[-1])
    parts, total = CenterHeadLossFunction.apply(cls_preds.reshape(-1, nc), box_preds.reshape(-1, 8), targets["heatmaps"][0][0],
                                                targets["anno_boxes"][0][0], targets["inds"][0][0], targets["masks"][0][0],
                                                lw["cls_weight"], lw["loc_weight"], lw["code_weights"])
    if as_tensors:   # (no host read-back here: a step over several windows reads all its reported numbers once at the end)
        return total, torch.cat([parts, total.detach().reshape(1)])
    host = torch.cat([parts, total.detach().reshape(1)]).cpu()
    tb = {"rpn_loss_cls": float(host[0]), "rpn_loss_loc": float(host[1]), "rpn_loss": float(host[2])}
    return total, tb

"""Training-step slice ("next" row 2): gradients of the sparse convolution and the MOS loss.  CPU: the oracle against the
reference's MOSLoss + torch autograd (golden); GPU: the HIP kernels behind insmos_amd.autograd against the oracle."""
import os

import numpy as np
import pytest

from oracle import ref_ops as R


def test_oracle_mos_loss_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "mos_loss.npz"))
    loss, grad = R.mos_loss(g["logits"], g["gt"])
    assert abs(loss - float(g["loss"])) < 1e-6
    np.testing.assert_allclose(grad, g["grad"], atol=1e-9)
    assert (g["grad"][:50] == 0).all()  # the 1e-8 clamp is exercised


def test_oracle_conv_backward_matches_finite_differences():
    rng = np.random.default_rng(0)
    n_in, n_out, K, ci, co = 40, 30, 5, 3, 4
    nbr = rng.integers(-1, n_in, size=(K, n_out)).astype(np.int32)
    x, taps, dy = rng.normal(size=(n_in, ci)), rng.normal(size=(K, ci, co)), rng.normal(size=(n_out, co))
    def f(xx, ww):  # float64 forward (the oracle's forward helpers work in fp32: too coarse for finite differences)
        y = np.zeros((n_out, co))
        for k in range(K):
            o = np.nonzero(nbr[k] >= 0)[0]
            y[o] += xx[nbr[k][o]] @ ww[k]
        return float((y * dy).sum())
    dx, dw, db = R.sparse_conv_backward(x, nbr, taps, dy)
    eps = 1e-5
    for _ in range(10):
        i, c = rng.integers(n_in), rng.integers(ci)
        xp = x.copy(); xp[i, c] += eps
        xm = x.copy(); xm[i, c] -= eps
        assert abs((f(xp, taps) - f(xm, taps)) / (2 * eps) - dx[i, c]) < 1e-4
        k, c2 = rng.integers(K), rng.integers(co)
        wp = taps.copy(); wp[k, c, c2] += eps
        wm = taps.copy(); wm[k, c, c2] -= eps
        assert abs((f(x, wp) - f(x, wm)) / (2 * eps) - dw[k, c, c2]) < 1e-4
    np.testing.assert_allclose(db, dy.sum(0))


def _subm_table(rng, n, shape=(12, 40, 40)):
    """A real submanifold table (symmetric: tap k of o is i  <=>  tap K-1-k of i is o) from random voxels."""
    cells = rng.choice(shape[0] * shape[1] * shape[2], size=n, replace=False)
    coords = np.stack(np.unravel_index(cells, shape), 1).astype(np.int32)
    keys = R.key3(coords, shape)
    perm = np.argsort(keys).astype(np.int32)
    return R.spconv_nbr_subm(coords, keys[perm], perm, shape)


@pytest.fixture(params=[2, 1, 0], ids=["dw_rows", "dw_mfma", "dw_lds"])
def dw_kernel(request):
    """Every d/dW kernel of insmos_sparse_conv_backward_weight (include/insmos_hip.h: insmos_debug_dw_kernel); 2 is the default."""
    from insmos_amd import _lib
    lib = _lib.load()
    _lib.check(lib.insmos_debug_dw_kernel(request.param), "insmos_debug_dw_kernel")
    yield request.param
    lib.insmos_debug_dw_kernel(2)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,K", [(8, 8, 27), (16, 32, 27), (32, 16, 27), (5, 3, 27), (64, 64, 27), (16, 16, 1), (48, 32, 27),
                                        (144, 128, 27), (128, 64, 27), (20, 128, 27),
                                        # widths that are NOT a multiple of the dW kernel's per-lane vector (2 for 16 < c <= 32, 4 above)
                                        (17, 18, 27), (34, 34, 27), (20, 18, 1)])
def test_sparse_conv_autograd_submanifold(cin, cout, K, dw_kernel):
    import torch
    from insmos_amd.autograd import sparse_conv
    rng = np.random.default_rng(cin * 31 + cout)
    n = 3000
    nbr = None if K == 1 else _subm_table(rng, n)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    taps = (rng.normal(size=(K, cin, cout)) * 0.2).astype(np.float32)
    bias = rng.normal(size=cout).astype(np.float32)
    gy = rng.normal(size=(n, cout)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    wt = torch.from_numpy(taps).cuda().requires_grad_(True)
    bt = torch.from_numpy(bias).cuda().requires_grad_(True)
    nb = torch.from_numpy(nbr).cuda() if nbr is not None else None
    y = sparse_conv(xt, wt, bt, nb)
    np.testing.assert_allclose(y.detach().cpu().numpy(), R.sparse_conv_numpy(x, nbr, taps) + bias if nbr is not None else x @ taps[0] + bias,
                               rtol=2e-4, atol=2e-4)
    (y * torch.from_numpy(gy).cuda()).sum().backward()
    dx, dw, db = R.sparse_conv_backward(x, nbr, taps, gy)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), dx, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(wt.grad.cpu().numpy(), dw, rtol=2e-4, atol=5e-4)
    np.testing.assert_allclose(bt.grad.cpu().numpy(), db, rtol=2e-4, atol=5e-4)
    # deterministic: a second backward gives the same bits
    xt.grad = None; wt.grad = None; bt.grad = None
    y2 = sparse_conv(xt, wt, bt, nb)
    (y2 * torch.from_numpy(gy).cuda()).sum().backward()
    assert torch.equal(wt.grad, torch.from_numpy(wt.grad.cpu().numpy()).cuda())
    g1 = wt.grad.clone()
    xt.grad = None; wt.grad = None; bt.grad = None
    (sparse_conv(xt, wt, bt, nb) * torch.from_numpy(gy).cuda()).sum().backward()
    assert torch.equal(g1, wt.grad)


@pytest.mark.gpu
def test_sparse_conv_autograd_strided_pair(dw_kernel):
    """A stride-2 layer: its transposed table is the inverse-conv table of the same pairs (down <-> inverse)."""
    import torch
    from insmos_amd.autograd import sparse_conv
    rng = np.random.default_rng(5)
    shape = (9, 33, 33)
    cells = rng.choice(shape[0] * shape[1] * shape[2], size=2500, replace=False)
    fine = np.stack(np.unravel_index(np.sort(cells), shape), 1).astype(np.int32)
    fkeys = R.key3(fine, shape)
    ocoords, okeys, oshape = R.spconv_down_coords(fine, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    down = R.spconv_nbr_down(ocoords, fkeys, None, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))          # (27, n_coarse) -> fine rows
    inv = R.spconv_nbr_inverse(fine, okeys, None, oshape, (3, 3, 3), (2, 2, 2), (1, 1, 1))           # (27, n_fine) -> coarse rows
    cin, cout = 16, 32
    x = rng.normal(size=(len(fine), cin)).astype(np.float32)
    taps = (rng.normal(size=(27, cin, cout)) * 0.2).astype(np.float32)
    gy = rng.normal(size=(len(ocoords), cout)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    wt = torch.from_numpy(taps).cuda().requires_grad_(True)
    y = sparse_conv(xt, wt, None, torch.from_numpy(down).cuda(), torch.from_numpy(inv).cuda())
    (y * torch.from_numpy(gy).cuda()).sum().backward()
    dx, dw, _ = R.sparse_conv_backward(x, down, taps, gy)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), dx, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(wt.grad.cpu().numpy(), dw, rtol=2e-4, atol=5e-4)


@pytest.mark.gpu
def test_mos_loss_kernel_vs_reference(golden_dir):
    import torch
    from insmos_amd.autograd import mos_loss
    g = np.load(os.path.join(golden_dir, "mos_loss.npz"))
    lg = torch.from_numpy(g["logits"]).cuda().requires_grad_(True)
    loss = mos_loss(lg, torch.from_numpy(g["gt"]).cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-6
    np.testing.assert_allclose(lg.grad.cpu().numpy(), g["grad"], atol=1e-8, rtol=1e-4)
    assert bool((lg.grad[:50] == 0).all())


@pytest.mark.gpu
def test_two_layer_training_steps_reduce_the_loss():
    """conv -> ReLU -> conv -> MOS loss chained on torch's autograd tape; plain SGD brings the loss down."""
    import torch
    from insmos_amd.autograd import mos_loss, sparse_conv
    rng = np.random.default_rng(1)
    n = 4000
    nbr = torch.from_numpy(_subm_table(rng, n)).cuda()
    x = torch.from_numpy(rng.normal(size=(n, 8)).astype(np.float32)).cuda()
    gt = torch.from_numpy(np.where(x[:, 0].cpu().numpy() > 0.3, 2, 1)).cuda()  # learnable from the input
    w1 = torch.from_numpy((rng.normal(size=(27, 8, 16)) * 0.1).astype(np.float32)).cuda().requires_grad_(True)
    b1 = torch.zeros(16, device="cuda", requires_grad=True)
    w2 = torch.from_numpy((rng.normal(size=(27, 16, 3)) * 0.1).astype(np.float32)).cuda().requires_grad_(True)
    b2 = torch.zeros(3, device="cuda", requires_grad=True)
    losses = []
    for step in range(30):
        h = torch.relu(sparse_conv(x, w1, b1, nbr))
        loss = mos_loss(sparse_conv(h, w2, b2, nbr), gt)
        for p in (w1, b1, w2, b2):
            p.grad = None
        loss.backward()
        with torch.no_grad():
            for p in (w1, b1, w2, b2):
                p -= 0.5 * p.grad
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.6 * losses[0] and all(b <= a + 1e-4 for a, b in zip(losses, losses[1:])), losses


@pytest.mark.gpu
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_train_vs_torch(relu):
    import torch
    import torch.nn.functional as F
    from insmos_amd.autograd import batch_norm_train
    rng = np.random.default_rng(3)
    n, c = 5000, 24
    x = (rng.normal(size=(n, c)) * rng.uniform(0.5, 3, c) + rng.normal(size=c)).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(size=c).astype(np.float32)
    gy = rng.normal(size=(n, c)).astype(np.float32)
    # reference: torch CPU float64
    xr = torch.from_numpy(x).double().requires_grad_(True)
    gr, br = torch.from_numpy(gamma).double().requires_grad_(True), torch.from_numpy(beta).double().requires_grad_(True)
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    yr = F.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-3)
    if relu:
        yr = torch.relu(yr)
    (yr * torch.from_numpy(gy).double()).sum().backward()
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    gt_, bt = torch.from_numpy(gamma).cuda().requires_grad_(True), torch.from_numpy(beta).cuda().requires_grad_(True)
    rm2, rv2 = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    y = batch_norm_train(xt, gt_, bt, rm2, rv2, momentum=0.1, eps=1e-3, relu=relu)
    (y * torch.from_numpy(gy).cuda()).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(rm2.cpu().numpy(), rm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv2.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(gt_.grad.cpu().numpy(), gr.grad.numpy(), rtol=1e-3, atol=5e-3)
    np.testing.assert_allclose(bt.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-3, atol=5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [None, "1024", "256"])
@pytest.mark.parametrize("relu,S", [(False, 1), (True, 1), (True, 3), (False, 4)])
def test_segmented_batchnorm_vs_torch_per_segment(relu, S, chunk, monkeypatch):
    """(`chunk`: the adaptive chunk lengths of BnPlan.chunk_rows -- 2048 rows for this width -- and two pinned lengths,
    INSMOS_BN_CHUNK, read on every call: the same float64 yardstick for every chunk table.)
    insmos_batchnorm_seg_*: every segment (one window of a training batch, here interleaved runs of rows like the 4D branch's
    (t * B + b)-major order) is normalised with ITS OWN batch statistics, the running statistics move segment after segment, and
    the gradients are those of torch's batch_norm applied to each segment's rows on their own (float64 reference); S = 1 is the
    plain layer."""
    import torch
    import torch.nn.functional as F
    from insmos_amd.autograd import BnPlan, batch_norm_train_seg
    if chunk is None:
        monkeypatch.delenv("INSMOS_BN_CHUNK", raising=False)
    else:
        monkeypatch.setenv("INSMOS_BN_CHUNK", chunk)
    rng = np.random.default_rng(40 + S)
    c = 24
    run_len = [int(v) for v in rng.integers(1, 2600, size=7 * S)]
    seg_of_run = [int(i % S) for i in range(len(run_len))]          # interleaved, like time slices of B windows
    seg = np.concatenate([np.full(l, sg, np.int32) for l, sg in zip(run_len, seg_of_run)])
    n = len(seg)
    x = (rng.normal(size=(n, c)) * rng.uniform(0.5, 3, c) + rng.normal(size=c) + seg[:, None] * 0.7).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(size=c).astype(np.float32)
    gy = rng.normal(size=(n, c)).astype(np.float32)
    xr = torch.from_numpy(x).double().requires_grad_(True)
    gr, br = torch.from_numpy(gamma).double().requires_grad_(True), torch.from_numpy(beta).double().requires_grad_(True)
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    yr = torch.zeros((n, c), dtype=torch.float64)
    segt = torch.from_numpy(seg).long()
    for sg in range(S):                                               # item after item, like models/models.py:313
        rows = torch.nonzero(segt == sg).flatten()
        ys = F.batch_norm(xr[rows], rm, rv, gr, br, training=True, momentum=0.01, eps=1e-3)
        yr = yr.index_copy(0, rows, torch.relu(ys) if relu else ys)
    (yr * torch.from_numpy(gy).double()).sum().backward()
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    gt_, bt = torch.from_numpy(gamma).cuda().requires_grad_(True), torch.from_numpy(beta).cuda().requires_grad_(True)
    rm2, rv2 = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    plan = BnPlan.from_segment_ids(torch.from_numpy(seg).cuda(), S)
    assert plan.S == S and plan.n_rows == n and int(plan.seg_rows_host.sum()) == n
    y = batch_norm_train_seg(xt, gt_, bt, plan, rm2, rv2, momentum=0.01, eps=1e-3, relu=relu, force_segmented=True)
    (y * torch.from_numpy(gy).cuda()).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(rm2.cpu().numpy(), rm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv2.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(gt_.grad.cpu().numpy(), gr.grad.numpy(), rtol=1e-3, atol=5e-3)
    np.testing.assert_allclose(bt.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-3, atol=5e-3)
    # deterministic
    xt.grad = None
    y2 = batch_norm_train_seg(xt, gt_, bt, plan, None, None, eps=1e-3, relu=relu, force_segmented=True)
    assert torch.equal(y2, y)


@pytest.mark.gpu
@pytest.mark.parametrize("c,relu,S", [(8, True, 4), (16, False, 3), (32, True, 1), (128, True, 2), (256, True, 2)])
def test_segmented_batchnorm_recompute_is_bitwise_the_stored_xhat_path(c, relu, S, monkeypatch):
    """Round 6: the segmented BatchNorm keeps the layer's input instead of storing x^ (insmos_batchnorm_seg_forward with xhat = NULL,
    insmos_batchnorm_seg_backward_x recomputes x^ and the ReLU mask): outputs, running statistics and all three gradients are the
    bits of the stored-x^ path, for every 16-byte width, with and without the fused ReLU, one and several segments -- and the
    float64 torch yardstick still holds."""
    import torch
    import torch.nn.functional as F
    from insmos_amd import autograd as AG
    monkeypatch.delenv("INSMOS_BN_CHUNK", raising=False)
    rng = np.random.default_rng(400 + c + S)
    run_len = [int(v) for v in rng.integers(1, 1900, size=5 * S)]
    seg = np.concatenate([np.full(l, i % S, np.int32) for i, l in enumerate(run_len)])
    n = len(seg)
    x = (rng.normal(size=(n, c)) * rng.uniform(0.5, 3, c) + rng.normal(size=c) + seg[:, None] * 0.7).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.normal(size=c).astype(np.float32)
    gy = torch.from_numpy(rng.normal(size=(n, c)).astype(np.float32)).cuda()
    plan = AG.BnPlan.from_segment_ids(torch.from_numpy(seg).cuda(), S)
    lib = AG._lib.load()
    assert lib.insmos_batchnorm_seg_recompute_ok(c, c, c, c) == 1 and lib.insmos_batchnorm_seg_recompute_ok(24, 24, 24, 24) == 0

    def run(recompute):
        monkeypatch.setattr(AG, "BN_RECOMPUTE", recompute)
        xt = torch.from_numpy(x).cuda().requires_grad_(True)
        gt_, bt = torch.from_numpy(gamma).cuda().requires_grad_(True), torch.from_numpy(beta).cuda().requires_grad_(True)
        rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        y = AG.batch_norm_train_seg(xt, gt_, bt, plan, rm, rv, momentum=0.01, eps=1e-3, relu=relu, force_segmented=True)
        (y * gy).sum().backward()
        torch.cuda.synchronize()
        return y.detach(), xt.grad, gt_.grad, bt.grad, rm, rv

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    xr = torch.from_numpy(x).double().requires_grad_(True)
    gr, br = torch.from_numpy(gamma).double().requires_grad_(True), torch.from_numpy(beta).double().requires_grad_(True)
    yr = torch.zeros((n, c), dtype=torch.float64)
    segt = torch.from_numpy(seg).long()
    for sg in range(S):
        rows = torch.nonzero(segt == sg).flatten()
        ys = F.batch_norm(xr[rows], None, None, gr, br, training=True, eps=1e-3)
        yr = yr.index_copy(0, rows, torch.relu(ys) if relu else ys)
    (yr * gy.cpu().double()).sum().backward()
    np.testing.assert_allclose(a[0].cpu().numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(a[1].cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=3e-4)
    np.testing.assert_allclose(a[2].cpu().numpy(), gr.grad.numpy(), rtol=1e-3, atol=1e-2)
    np.testing.assert_allclose(a[3].cpu().numpy(), br.grad.numpy(), rtol=1e-3, atol=1e-2)


@pytest.mark.gpu
def test_conv_bn_relu_conv_loss_graph_vs_torch_reference():
    """One trainable block (SubMConv3d -> BatchNorm1d -> ReLU -> SubMConv3d -> MOSLoss, train mode) on the HIP autograd
    nodes vs the same graph written with torch index ops in float64 on the CPU: loss and every parameter gradient."""
    import torch
    import torch.nn.functional as F
    from insmos_amd.autograd import batch_norm_train, mos_loss, sparse_conv
    rng = np.random.default_rng(8)
    n = 3500
    nbr = _subm_table(rng, n)
    x = rng.normal(size=(n, 8)).astype(np.float32)
    gt = rng.integers(0, 3, n)
    P0 = dict(w1=(rng.normal(size=(27, 8, 16)) * 0.2).astype(np.float32), g1=rng.uniform(0.5, 1.5, 16).astype(np.float32),
              b1=rng.normal(size=16).astype(np.float32) * 0.1, w2=(rng.normal(size=(27, 16, 3)) * 0.2).astype(np.float32),
              c2=rng.normal(size=3).astype(np.float32) * 0.1)

    def ref_conv(xx, ww, bias):
        y = torch.zeros((n, ww.shape[2]), dtype=torch.float64)
        for k in range(27):
            o = torch.from_numpy(np.nonzero(nbr[k] >= 0)[0])
            i = torch.from_numpy(nbr[k][nbr[k] >= 0].astype(np.int64))
            y = y.index_add(0, o, xx[i] @ ww[k])
        return y if bias is None else y + bias

    pr = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in P0.items()}
    h = ref_conv(torch.from_numpy(x).double(), pr["w1"], None)
    h = torch.relu(F.batch_norm(h, None, None, pr["g1"], pr["b1"], training=True, eps=1e-3))
    z = ref_conv(h, pr["w2"], pr["c2"])
    lw = torch.tensor([0.0, 0.5, 0.5], dtype=torch.float64)
    zz = z.clone()
    zz[:, 0] = -float("inf")
    loss_r = F.nll_loss(torch.log(torch.softmax(zz, 1).clamp(min=1e-8)), torch.from_numpy(gt), weight=lw)
    loss_r.backward()

    pg = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in P0.items()}
    nb = torch.from_numpy(nbr).cuda()
    h = sparse_conv(torch.from_numpy(x).cuda(), pg["w1"], None, nb)
    h = batch_norm_train(h, pg["g1"], pg["b1"], eps=1e-3, relu=True)
    loss = mos_loss(sparse_conv(h, pg["w2"], pg["c2"], nb), torch.from_numpy(gt).cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_r.detach())) < 1e-5
    for k in P0:
        np.testing.assert_allclose(pg[k].grad.cpu().numpy(), pr[k].grad.numpy(), rtol=2e-3, atol=2e-5, err_msg=k)


@pytest.mark.gpu
def test_motionnet_training_step_vs_torch_reference():
    """MotionNet in train mode (batch-stat BN) + loss_motion_encoder: loss and the gradients of ALL parameters against the
    same graph written with torch index ops in float64 on the CPU over the same kernel maps; then SGD lowers the loss."""
    import torch
    import torch.nn.functional as F
    from insmos_amd import params as P
    from insmos_amd.synth import make_labels, make_window
    from insmos_amd.train_motionnet import MotionNetTrainer
    cfg = P.default_cfg()
    sd = P.random_state_dict(cfg, 6)
    w = make_window(seed=8, n_scans=3, n_az=72)
    gt = make_labels(w[w[:, 4] == 0], seed=8)
    tr = MotionNetTrainer(cfg, sd)
    pts = torch.from_numpy(w).cuda()
    loss = tr.loss(pts, torch.from_numpy(gt).cuda())
    loss.backward()
    T = tr.engine._me_tables
    tabs = dict(n125=T["nbr125"].nbr.cpu().numpy(), n81=[t.nbr.cpu().numpy() for t in T["nbr81"]],
                dn=[t.nbr.cpu().numpy() for t in T["dn"]], up=[t.nbr.cpu().numpy() for t in T["up"]])
    inverse = T["inverse"].cpu().numpy()

    pr = {k: v.detach().cpu().double().requires_grad_(True) for k, v in tr.params.items()}

    def conv(x, wname, nbr, bias=None):
        ww = pr[wname]
        if nbr is None:
            y = x @ ww[0]
        else:
            y = torch.zeros((nbr.shape[1], ww.shape[2]), dtype=torch.float64)
            for k in range(nbr.shape[0]):
                o = np.nonzero(nbr[k] >= 0)[0]
                if len(o):
                    y = y.index_add(0, torch.from_numpy(o), x[torch.from_numpy(nbr[k][o].astype(np.int64))] @ ww[k])
        return y if bias is None else y + pr[bias]

    def bn(x, name, relu):
        y = F.batch_norm(x, None, None, pr[name + ".weight"], pr[name + ".bias"], training=True, eps=1e-5)
        return torch.relu(y) if relu else y

    def block(name, x, nbr):
        out = bn(conv(x, name + ".conv1.kernel", nbr), name + ".norm1", True)
        out = bn(conv(out, name + ".conv2.kernel", nbr), name + ".norm2", False)
        res = bn(conv(x, name + ".downsample.0.kernel", None), name + ".downsample.1", False) if (name + ".downsample.0.kernel") in pr else x
        return torch.relu(out + res)

    n0 = tabs["n81"][0].shape[1]
    out_p1 = bn(conv(torch.full((n0, 1), 0.5, dtype=torch.float64), "conv0p1s1.kernel", tabs["n125"]), "bn0", True)
    out = bn(conv(out_p1, "conv1p1s2.kernel", tabs["dn"][0]), "bn1", True)
    b1 = block("block1.0", out, tabs["n81"][1])
    out = bn(conv(b1, "conv2p2s2.kernel", tabs["dn"][1]), "bn2", True)
    b2 = block("block2.0", out, tabs["n81"][2])
    out = bn(conv(b2, "conv3p4s2.kernel", tabs["dn"][2]), "bn3", True)
    out = block("block3.0", out, tabs["n81"][3])
    out = bn(conv(out, "convtr5p8s2.kernel", tabs["up"][2]), "bntr5", True)
    out = block("block6.0", torch.cat([out, b2], 1), tabs["n81"][2])
    out = bn(conv(out, "convtr6p4s2.kernel", tabs["up"][1]), "bntr6", True)
    out = block("block7.0", torch.cat([out, b1], 1), tabs["n81"][1])
    out = bn(conv(out, "convtr7p2s2.kernel", tabs["up"][0]), "bntr7", True)
    out = block("block8.0", torch.cat([out, out_p1], 1), tabs["n81"][0])
    motion = conv(out, "final.kernel", None, "final.bias")
    cur = np.nonzero(np.floor(w[:, 4] / np.float32(0.1)) == 0)[0]
    z = motion[torch.from_numpy(inverse[cur].astype(np.int64))].clone()
    z[:, 0] = -float("inf")
    loss_r = F.nll_loss(torch.log(torch.softmax(z, 1).clamp(min=1e-8)), torch.from_numpy(gt).long(),
                        weight=torch.tensor([0.0, 0.5, 0.5], dtype=torch.float64))
    loss_r.backward()
    assert abs(float(loss.detach()) - float(loss_r.detach())) < 2e-4, (float(loss.detach()), float(loss_r.detach()))
    worst = 0.0
    for k, v in tr.params.items():
        g, r = v.grad.cpu().numpy(), pr[k].grad.numpy()
        scale = max(np.abs(r).max(), 1e-6)
        worst = max(worst, float(np.abs(g - r).max() / scale))
        assert np.abs(g - r).max() <= 5e-3 * scale + 1e-6, (k, float(np.abs(g - r).max()), float(scale))
    print("worst relative gradient error", worst)
    l0 = float(loss.detach())
    for _ in range(8):
        tr.sgd_step(0.05)
        loss = tr.loss(pts, torch.from_numpy(gt).cuda())
        loss.backward()
    assert float(loss.detach()) < l0


# ---------------------------------------------------------------------------------------------------------------------
# CenterHead targets + losses (center_head.py:126-331)
# ---------------------------------------------------------------------------------------------------------------------
HEAD_CFG = {"TARGET_ASSIGNER_CONFIG": {"MAX_OBJS": 14, "VOXEL_SIZE": [0.1, 0.1, 0.1], "OUT_SIZE_FACTOR": 4,
                                       "GAUSSIAN_OVERLAP": 0.1, "MIN_RADIUS": 2},
            "LOSS_CONFIG": {"LOSS_WEIGHTS": {"cls_weight": 1.0, "loc_weight": 2.0, "code_weights": [1.0] * 8}}}


def _oracle_targets(g, pc_range):
    tc = HEAD_CFG["TARGET_ASSIGNER_CONFIG"]
    return R.center_assign_targets(g["gt_boxes"], g["grid"], pc_range, tc["VOXEL_SIZE"], tc["OUT_SIZE_FACTOR"], 3,
                                   tc["MAX_OBJS"], tc["GAUSSIAN_OVERLAP"], tc["MIN_RADIUS"])


@pytest.mark.parametrize("tag", ["", "_f64range"])
def test_oracle_center_targets_vs_reference(golden_dir, tag):
    """Oracle against CenterHead.get_targets_single run as written (tests/golden/make_golden.py:center_loss_golden):
    integer-valued range (fp32 cell arithmetic, the shipped config) and float-valued range (torch promotes to float64)."""
    g = np.load(os.path.join(golden_dir, "center_loss.npz"))
    pr = g["pc_range"] if tag == "" else g["pc_range"].astype(np.float64)
    assert g["pc_range"].dtype.kind == "i"
    heat, anno, ind, mask = _oracle_targets(g, pr)
    np.testing.assert_array_equal(heat, g["heatmap" + tag])
    np.testing.assert_array_equal(ind, g["ind" + tag])
    np.testing.assert_array_equal(mask, g["mask" + tag])
    np.testing.assert_array_equal(anno[:, :3], g["anno_box" + tag][:, :3])       # offsets and z: exact
    np.testing.assert_allclose(anno[:, 3:], g["anno_box" + tag][:, 3:], atol=2e-7)  # log / sin / cos: libm vs torch, 1 ulp
    # the cases the fixture was built around
    assert mask.tolist() == [1, 0, 0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1]
    assert ind[3] % 40 == 0 and ind[9] // 40 == 0      # coordinates in (-1, 0) truncate into cell 0
    assert ind[0] == ind[6] and int((heat == 1).sum()) == int(mask.sum()) - 1
    assert ind[8] == 40 * 30 - 1


def test_oracle_center_loss_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "center_loss.npz"))
    lc, ll, gc, gb = R.center_head_loss(g["cls_preds"][0], g["box_preds"][0], g["heatmap"], g["anno_box"], g["ind"], g["mask"])
    assert abs(lc - float(g["loss_cls"])) < 1e-5 * float(g["loss_cls"])
    assert abs(ll - float(g["loss_loc"])) < 1e-5 * float(g["loss_loc"])
    np.testing.assert_allclose(gc, g["grad_cls"][0], atol=2e-7, rtol=1e-5)
    np.testing.assert_allclose(gb, g["grad_box"][0], atol=1e-8, rtol=1e-6)
    assert (g["grad_cls"][0, 3, 4, :2] == 0).all()      # sigmoid clipped on both sides -> no gradient


def test_oracle_center_loss_matches_finite_differences():
    rng = np.random.default_rng(5)
    H, W = 6, 7
    gt = np.array([[0.5, 0.3, -1.0, 1.6, 0.8, 1.5, 0.4, 1], [-0.9, 0.6, -1.2, 0.9, 0.7, 1.7, -1.0, 3]], np.float32)
    heat, anno, ind, mask = R.center_assign_targets(gt, [28, 24, 40], np.array([-1.4, -1.2, -3, 1.4, 1.2, 1]), [0.1] * 3, 4, 3,
                                                    5, 0.1, 2)
    cls = rng.normal(size=(H, W, 3))
    box = rng.normal(size=(H, W, 8))
    lc, ll, gc, gb = R.center_head_loss(cls, box, heat, anno, ind, mask)
    h = 1e-5
    for _ in range(12):
        i, j, c = rng.integers(0, H), rng.integers(0, W), rng.integers(0, 3)
        cp, cm = cls.copy(), cls.copy()
        cp[i, j, c] += h
        cm[i, j, c] -= h
        fd = (sum(R.center_head_loss(cp, box, heat, anno, ind, mask)[:2]) - sum(R.center_head_loss(cm, box, heat, anno, ind, mask)[:2])) / (2 * h)
        assert abs(fd - gc[i, j, c]) < 1e-5 + 1e-4 * abs(fd)
    for k in range(2):
        y, x = divmod(int(ind[k]), W)
        bp, bm = box.copy(), box.copy()
        bp[y, x, 2] += h
        bm[y, x, 2] -= h
        fd = (sum(R.center_head_loss(cls, bp, heat, anno, ind, mask)[:2]) - sum(R.center_head_loss(cls, bm, heat, anno, ind, mask)[:2])) / (2 * h)
        assert abs(fd - gb[y, x, 2]) < 1e-6


def _gpu_targets(g, pc_range):
    import torch
    from insmos_amd.autograd import center_assign_targets
    return center_assign_targets(torch.from_numpy(g["gt_boxes"])[None].cuda(), HEAD_CFG, g["grid"], pc_range, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["", "_f64range"])
def test_center_targets_kernel_vs_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "center_loss.npz"))
    pr = g["pc_range"] if tag == "" else g["pc_range"].astype(np.float64)
    tg = _gpu_targets(g, pr)
    heat, anno = tg["heatmaps"][0][0].cpu().numpy(), tg["anno_boxes"][0][0].cpu().numpy()
    ind, mask = tg["inds"][0][0].cpu().numpy(), tg["masks"][0][0].cpu().numpy()
    np.testing.assert_array_equal(ind, g["ind" + tag])          # integer work: exact
    np.testing.assert_array_equal(mask, g["mask" + tag])
    assert ((heat == 1) == (g["heatmap" + tag] == 1)).all() and ((heat == 0) == (g["heatmap" + tag] == 0)).all()
    np.testing.assert_allclose(heat, g["heatmap" + tag], atol=0, rtol=2e-7)  # float64 exp on the device, rounded to fp32
    np.testing.assert_array_equal(anno[:, :3], g["anno_box" + tag][:, :3])
    np.testing.assert_allclose(anno[:, 3:], g["anno_box" + tag][:, 3:], atol=5e-7)
    oh, oa, oi, om = _oracle_targets(g, pr)
    np.testing.assert_array_equal(ind, oi)
    np.testing.assert_allclose(heat, oh, atol=0, rtol=2e-7)


@pytest.mark.gpu
def test_center_targets_kernel_many_boxes_vs_oracle():
    """Full-size map (300 x 250), 100 slots, 180 random boxes incl. out-of-range / degenerate ones; twice -> same bits."""
    import torch
    from insmos_amd.autograd import center_assign_targets
    from insmos_amd import params as P
    rng = np.random.default_rng(9)
    cfg = P.default_cfg()
    hc = cfg["MODEL"]["DENSE_HEAD"]
    M = 180
    gt = np.zeros((M, 8), np.float32)
    gt[:, 0] = rng.uniform(-66, 66, M)
    gt[:, 1] = rng.uniform(-55, 55, M)
    gt[:, 2] = rng.uniform(-2, 0, M)
    gt[:, 3:6] = rng.uniform(0.3, 12.0, (M, 3))
    gt[:, 6] = rng.uniform(-3.2, 3.2, M)
    gt[:, 7] = rng.integers(0, 4, M)
    gt[::17, 3] = 0
    grid = np.array([1200, 1000, 40])
    pr = np.array(cfg["DATA"]["POINT_CLOUD_RANGE"])
    tc = hc["TARGET_ASSIGNER_CONFIG"]
    oh, oa, oi, om = R.center_assign_targets(gt, grid, pr, tc["VOXEL_SIZE"], tc["OUT_SIZE_FACTOR"], 3, tc["MAX_OBJS"],
                                             tc["GAUSSIAN_OVERLAP"], tc["MIN_RADIUS"])
    outs = []
    for _ in range(2):
        tg = center_assign_targets(torch.from_numpy(gt)[None].cuda(), hc, grid, pr, 3)
        outs.append([tg[k][0][0].cpu().numpy() for k in ("heatmaps", "anno_boxes", "inds", "masks")])
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    heat, anno, ind, mask = outs[0]
    assert 20 < om.sum() < 100
    np.testing.assert_array_equal(ind, oi)
    np.testing.assert_array_equal(mask, om)
    np.testing.assert_allclose(heat, oh, atol=0, rtol=2e-7)
    np.testing.assert_array_equal(anno[:, :3], oa[:, :3])
    np.testing.assert_allclose(anno[:, 3:], oa[:, 3:], atol=5e-7)


@pytest.mark.gpu
def test_center_head_loss_kernel_vs_reference(golden_dir):
    import torch
    from insmos_amd.autograd import center_head_loss
    g = np.load(os.path.join(golden_dir, "center_loss.npz"))
    tg = {"heatmaps": [torch.from_numpy(g["heatmap"])[None].cuda()], "anno_boxes": [torch.from_numpy(g["anno_box"])[None].cuda()],
          "inds": [torch.from_numpy(g["ind"])[None].cuda()], "masks": [torch.from_numpy(g["mask"])[None].cuda()]}
    cls = torch.from_numpy(g["cls_preds"]).cuda().requires_grad_(True)
    box = torch.from_numpy(g["box_preds"]).cuda().requires_grad_(True)
    loss, tb = center_head_loss(cls, box, tg, HEAD_CFG)
    loss.backward()
    assert abs(tb["rpn_loss_cls"] - float(g["loss_cls"])) < 2e-5 * float(g["loss_cls"])
    assert abs(tb["rpn_loss_loc"] - float(g["loss_loc"])) < 2e-5 * float(g["loss_loc"])
    assert abs(tb["rpn_loss"] - float(g["loss"])) < 2e-5 * float(g["loss"])
    np.testing.assert_allclose(cls.grad.cpu().numpy(), g["grad_cls"], atol=3e-7, rtol=2e-4)
    np.testing.assert_allclose(box.grad.cpu().numpy(), g["grad_box"], atol=1e-8, rtol=1e-5)
    assert bool((cls.grad[0, 3, 4, :2] == 0).all())
    # the oracle agrees to the same tolerance, and a second call gives the same bits (fixed-order reductions)
    lc, ll, gc, gb = R.center_head_loss(g["cls_preds"][0], g["box_preds"][0], g["heatmap"], g["anno_box"], g["ind"], g["mask"])
    assert abs(tb["rpn_loss"] - (lc + ll)) < 2e-5 * (lc + ll)
    cls2 = torch.from_numpy(g["cls_preds"]).cuda().requires_grad_(True)
    box2 = torch.from_numpy(g["box_preds"]).cuda().requires_grad_(True)
    loss2, _ = center_head_loss(cls2, box2, tg, HEAD_CFG)
    loss2.backward()
    assert float(loss2.detach()) == float(loss.detach())
    assert bool((cls2.grad == cls.grad).all()) and bool((box2.grad == box.grad).all())


@pytest.mark.gpu
def test_center_head_loss_full_map_vs_oracle():
    """BASELINE cfg-2 head map (250 x 300 cells, 3 classes): targets from the kernel, loss + gradients against the oracle."""
    import torch
    from insmos_amd.autograd import center_assign_targets, center_head_loss
    from insmos_amd import params as P
    rng = np.random.default_rng(21)
    cfg = P.default_cfg()
    hc = cfg["MODEL"]["DENSE_HEAD"]
    M = 60
    gt = np.zeros((M, 8), np.float32)
    gt[:, 0] = rng.uniform(-58, 58, M)
    gt[:, 1] = rng.uniform(-48, 48, M)
    gt[:, 2] = rng.uniform(-2, 0, M)
    gt[:, 3:6] = rng.uniform(0.5, 5.0, (M, 3))
    gt[:, 6] = rng.uniform(-3.2, 3.2, M)
    gt[:, 7] = rng.integers(1, 4, M)
    grid = np.array([1200, 1000, 40])
    pr = np.array(cfg["DATA"]["POINT_CLOUD_RANGE"])
    tg = center_assign_targets(torch.from_numpy(gt)[None].cuda(), hc, grid, pr, 3)
    H, W = 250, 300
    cls_np = (rng.normal(size=(1, H, W, 3)) * 1.5 - 3.0).astype(np.float32)
    box_np = rng.normal(size=(1, H, W, 8)).astype(np.float32)
    cls = torch.from_numpy(cls_np).cuda().requires_grad_(True)
    box = torch.from_numpy(box_np).cuda().requires_grad_(True)
    loss, tb = center_head_loss(cls, box, tg, hc)
    loss.backward()
    heat, anno = tg["heatmaps"][0][0].cpu().numpy(), tg["anno_boxes"][0][0].cpu().numpy()
    ind, mask = tg["inds"][0][0].cpu().numpy(), tg["masks"][0][0].cpu().numpy()
    lw = hc["LOSS_CONFIG"]["LOSS_WEIGHTS"]
    lc, ll, gc, gb = R.center_head_loss(cls_np[0], box_np[0], heat, anno, ind, mask, lw["cls_weight"], lw["loc_weight"],
                                        lw["code_weights"])
    assert abs(tb["rpn_loss_cls"] - lc) < 1e-4 * lc and abs(tb["rpn_loss_loc"] - ll) < 1e-4 * ll
    np.testing.assert_allclose(cls.grad.cpu().numpy()[0], gc, atol=3e-7, rtol=3e-4)
    np.testing.assert_allclose(box.grad.cpu().numpy()[0], gb, atol=1e-8, rtol=1e-5)
    assert int((box.grad != 0).any(dim=-1).sum()) <= int(mask.sum())


FUZZ_GEOMS = [(np.array([1200, 1000, 40]), np.array([-60, -50, -3, 60, 50, 1]), [0.1, 0.1, 0.1]),
              (np.array([2400, 2000, 80]), np.array([-60.0, -50.0, -3.0, 60.0, 50.0, 1.0]), [0.05, 0.05, 0.05]),
              (np.array([352, 400, 40]), np.array([0, -40, -3, 70.4, 40, 1]), [0.2, 0.2, 0.1])]


def test_oracle_center_targets_vs_reference_fuzz(golden_dir):
    """48 random box sets over three map geometries (integer and float ranges, three overlaps / minimum radii) through the
    reference's get_targets_single as written (make_golden.py:center_targets_fuzz_golden): cells, masks and heat maps
    bit-exact (heat maps by digest), regression rows to 1 ulp."""
    import hashlib
    g = np.load(os.path.join(golden_dir, "center_targets_fuzz.npz"))
    n_assigned = 0
    for case in range(int(g["n_cases"])):
        pre = "c%02d_" % case
        grid, pcr, vsz = FUZZ_GEOMS[int(g[pre + "geom"])]
        heat, anno, ind, mask = R.center_assign_targets(g[pre + "gt"], grid, pcr, vsz, 4, 3, 30, float(g[pre + "overlap"]),
                                                        int(g[pre + "min_radius"]))
        np.testing.assert_array_equal(mask, g[pre + "mask"], err_msg=pre)
        np.testing.assert_array_equal(ind, g[pre + "ind"], err_msg=pre)
        assert int((heat == 1).sum()) == int(g[pre + "heat_ones"]), pre
        assert hashlib.sha256(np.ascontiguousarray(heat).tobytes()).hexdigest() == str(g[pre + "heat_digest"]), pre
        np.testing.assert_array_equal(anno[:, :3], g[pre + "anno"][:, :3], err_msg=pre)
        ok = np.isfinite(g[pre + "anno"][:, 3:]).all(1)
        np.testing.assert_allclose(anno[ok, 3:], g[pre + "anno"][ok, 3:], rtol=3e-7, atol=3e-7, err_msg=pre)
        n_assigned += int(mask.sum())
    assert n_assigned > 400


@pytest.mark.gpu
def test_center_targets_kernel_vs_reference_fuzz(golden_dir):
    import torch
    from insmos_amd.autograd import center_assign_targets
    g = np.load(os.path.join(golden_dir, "center_targets_fuzz.npz"))
    for case in range(int(g["n_cases"])):
        pre = "c%02d_" % case
        grid, pcr, vsz = FUZZ_GEOMS[int(g[pre + "geom"])]
        hc = {"TARGET_ASSIGNER_CONFIG": {"MAX_OBJS": 30, "VOXEL_SIZE": vsz, "OUT_SIZE_FACTOR": 4,
                                         "GAUSSIAN_OVERLAP": float(g[pre + "overlap"]), "MIN_RADIUS": int(g[pre + "min_radius"])}}
        tg = center_assign_targets(torch.from_numpy(g[pre + "gt"])[None].cuda(), hc, grid, pcr, 3)
        np.testing.assert_array_equal(tg["inds"][0][0].cpu().numpy(), g[pre + "ind"], err_msg=pre)
        np.testing.assert_array_equal(tg["masks"][0][0].cpu().numpy(), g[pre + "mask"], err_msg=pre)
        oh = R.center_assign_targets(g[pre + "gt"], grid, pcr, vsz, 4, 3, 30, float(g[pre + "overlap"]), int(g[pre + "min_radius"]))[0]
        np.testing.assert_allclose(tg["heatmaps"][0][0].cpu().numpy(), oh, atol=0, rtol=2e-7, err_msg=pre)

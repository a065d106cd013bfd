"""-m gpu: detection-head post-processing, instance features and metrics against the oracle and the
golden vectors generated from the reference's own code."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
D = "cuda:0"


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _iou(a, b):
    from gpu_util import dev, lib, stream
    from insmos_amd import _lib
    out = torch.empty((len(a), len(b)), device=D)
    da, db = dev(a), dev(b)  # keep both alive: temporaries would share one allocation
    _lib.check(lib().insmos_iou_bev(da.data_ptr(), len(a), db.data_ptr(), len(b), out.data_ptr(), stream()), "iou")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_iou_bev_vs_reference_golden(golden_dir):
    g = G(golden_dir, "iou_bev.npz")
    # device sinf/cosf/atan2f differ from glibc by an ulp: tolerance, not bit-exactness
    np.testing.assert_allclose(_iou(g["a"], g["b"]), g["iou_ab"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(_iou(g["special"], g["special"]), g["iou_special"], rtol=1e-4, atol=2e-6)


def _nms(boxes, thresh, post_max, max_n=4096):
    from gpu_util import dev, lib, stream, ws
    from insmos_amd import _lib
    n = len(boxes)
    b = torch.zeros((max_n, 7), device=D)
    b[:n] = dev(boxes)
    nd = torch.tensor([n, 0, 0, 0], dtype=torch.int32, device=D)
    keep = torch.full((post_max,), -1, dtype=torch.int32, device=D)
    cnt = torch.zeros(4, dtype=torch.int32, device=D)
    w = ws(lib().insmos_nms_ws_bytes(max_n))
    _lib.check(lib().insmos_nms_rotated_bev(b.data_ptr(), nd.data_ptr(), max_n, thresh, post_max, keep.data_ptr(),
                                            cnt.data_ptr(), w.data_ptr(), w.numel(), stream()), "nms")
    torch.cuda.synchronize()
    return keep[:int(cnt[0])].cpu().numpy()


def test_nms_keep_lists_vs_reference_golden(golden_dir):
    g = G(golden_dir, "iou_bev.npz")
    np.testing.assert_array_equal(_nms(g["dense"], 0.01, 500), g["keep_001"])
    np.testing.assert_array_equal(_nms(g["dense"], 0.5, 500), g["keep_05"])
    np.testing.assert_array_equal(_nms(g["dense"], 0.01, 7), g["keep_001"][:7])  # post_max truncation
    assert len(_nms(g["dense"][:0], 0.01, 500)) == 0  # empty input


def test_nms_predicate_at_its_threshold_vs_reference_golden(golden_dir):
    """Pairs whose IoU -- as the reference's compiled iou3d_cpu.cpp computes it -- is the float 0.01 itself / one ulp above / one
    ulp below (axis-aligned boxes: sin 0 and cos 0 are exact on the device too, so this case IS bit-comparable, unlike the
    rotated ones): the device IoU is the same float and k_nms_mask's `>` suppresses exactly the 'above' partners."""
    g = G(golden_dir, "nms_threshold.npz")
    b = g["boxes"]
    iou = _iou(b, b)
    pair = np.array([iou[2 * i, 2 * i + 1] for i in range(len(b) // 2)], np.float32)
    np.testing.assert_array_equal(pair, g["pair_iou"])
    np.testing.assert_array_equal(_nms(b, float(g["thresh"][0]), 500), g["keep_001"])


def test_nms_large_matches_oracle():
    rng = np.random.default_rng(4)
    n = 4096
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-55, 55, (n, 2))
    b[:, 3] = rng.uniform(1.5, 5, n); b[:, 4] = rng.uniform(0.6, 2.2, n); b[:, 5] = 1.5
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    ref = R.nms_bev(b, 0.01)
    got = _nms(b, 0.01, 4096)
    # an ulp-level IoU difference right at the 0.01 threshold could flip a decision; require >= 99.9 % agreement
    common = len(np.intersect1d(ref, got))
    assert common >= 0.999 * max(len(ref), len(got)), (len(ref), len(got), common)


def test_decode_select_nms_vs_reference_golden(golden_dir):
    """insmos_center_decode_select + nms + gather == post_processing of the reference on raw-box input."""
    from gpu_util import dev, lib, stream, ws
    from insmos_amd import _lib
    g = G(golden_dir, "post_process.npz")
    cls, boxes = g["cls"], g["boxes"]  # (6000,3) logits, (6000,7) decoded boxes
    # build a head map whose decode reproduces `boxes`: up=1, H*W = 6000 = 60 x 100, out_factor*v = 1, x0=y0=0
    H, W = 60, 100
    n = H * W
    head = np.zeros((n, 12), np.float32)
    head[:, :3] = cls
    rows, cols = np.divmod(np.arange(n), W)
    head[:, 3] = boxes[:, 0] - cols
    head[:, 4] = boxes[:, 1] - rows
    head[:, 5] = boxes[:, 2]
    head[:, 6:9] = np.log(boxes[:, 3:6])
    head[:, 9] = np.sin(boxes[:, 6]); head[:, 10] = np.cos(boxes[:, 6])
    pre_max, post_max = int(g["pre_max"]), int(g["post_max"])
    cb = torch.empty((pre_max, 7), device=D); cs = torch.empty(pre_max, device=D)
    cl = torch.empty(pre_max, dtype=torch.int32, device=D); cc = torch.empty(pre_max, dtype=torch.int32, device=D)
    cnt = torch.zeros(4, dtype=torch.int32, device=D)
    w = ws(lib().insmos_center_decode_select_ws_bytes(n))
    dhead = dev(head)
    _lib.check(lib().insmos_center_decode_select(dhead.data_ptr(), 12, 3, H, W, 1, 1.0, 1.0, 1.0, 0.0, 0.0, 0.1,
                                                 pre_max, cb.data_ptr(), cs.data_ptr(), cl.data_ptr(), cc.data_ptr(),
                                                 cnt.data_ptr(), w.data_ptr(), w.numel(), stream()), "decode")
    keep = torch.empty(post_max, dtype=torch.int32, device=D)
    cntk = torch.zeros(4, dtype=torch.int32, device=D)
    w2 = ws(lib().insmos_nms_ws_bytes(pre_max))
    _lib.check(lib().insmos_nms_rotated_bev(cb.data_ptr(), cnt.data_ptr(), pre_max, 0.01, post_max, keep.data_ptr(),
                                            cntk.data_ptr(), w2.data_ptr(), w2.numel(), stream()), "nms")
    pb = torch.empty((post_max, 7), device=D); ps = torch.empty(post_max, device=D)
    pl = torch.empty(post_max, dtype=torch.int64, device=D)
    _lib.check(lib().insmos_gather_preds(cb.data_ptr(), cs.data_ptr(), cl.data_ptr(), keep.data_ptr(), cntk.data_ptr(),
                                         post_max, pb.data_ptr(), ps.data_ptr(), pl.data_ptr(), stream()), "gather")
    torch.cuda.synchronize()
    K = int(cntk[0])
    assert int(cnt[0]) == pre_max  # more than pre_max cells pass the score threshold in this fixture
    assert K == len(g["pred_labels"])
    np.testing.assert_array_equal(pl[:K].cpu().numpy(), g["pred_labels"])
    np.testing.assert_allclose(ps[:K].cpu().numpy(), g["pred_scores"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(pb[:K].cpu().numpy(), g["pred_boxes"], rtol=1e-5, atol=1e-5)


def _onehot(coords_xyz, boxes8, quirk=True):
    """boxes8 are already in voxel units: feed them as metric boxes with lo=0, vsize=1, stride=1, mult=1."""
    from gpu_util import dev, lib, stream, hp
    from insmos_amd import _lib
    n, m = len(coords_xyz), len(boxes8)
    c4 = np.zeros((n, 4), np.int32)
    c4[:, 1] = coords_xyz[:, 2]; c4[:, 2] = coords_xyz[:, 1]; c4[:, 3] = coords_xyz[:, 0]
    mb = max(m, 4)
    pb = torch.zeros((mb, 7), device=D); pb[:m] = dev(boxes8[:, :7])
    pl = torch.zeros(mb, dtype=torch.int64, device=D); pl[:m] = dev(boxes8[:, 7].astype(np.int64))
    nd = torch.tensor([m, 0, 0, 0], dtype=torch.int32, device=D)
    out = torch.full((n, 16), -1.0, device=D)
    scratch = torch.empty(int(lib().insmos_boxes_to_onehot_scratch_ints(mb, n)), dtype=torch.int32, device=D)
    lo = np.zeros(3, np.float32); vs = np.ones(3, np.float32)
    dc4 = dev(c4)
    _lib.check(lib().insmos_boxes_to_onehot(pb.data_ptr(), pl.data_ptr(), nd.data_ptr(), mb, hp(lo), hp(vs), 1.0, 1.0,
                                            dc4.data_ptr(), n, 3, 16, 1 if quirk else 0, out.data_ptr(), 16,
                                            scratch.data_ptr(), stream()), "onehot")
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert np.all(o[:, 3:] == 0)
    return o[:, :3].astype(np.int32)


def test_boxes_to_onehot_vs_reference_golden(golden_dir):
    g = G(golden_dir, "array_index.npz")
    for order in ("sorted", "perm1", "perm2"):
        np.testing.assert_array_equal(_onehot(g["coords_" + order], g["boxes"]), g["out_" + order])
    np.testing.assert_array_equal(_onehot(g["demo_coords"], g["demo_box"]), g["demo_out"])
    np.testing.assert_array_equal(_onehot(g["demo_coords_b"], g["demo_box"]), g["demo_out_b"])
    np.testing.assert_array_equal(_onehot(g["coords_C"], g["boxes_C"]), g["out_C"])
    # quirk off == plain geometric containment (oracle's quirk=False)
    np.testing.assert_array_equal(_onehot(g["coords_C"], g["boxes_C"], quirk=False),
                                  R.boxes_to_onehot(g["coords_C"], g["boxes_C"], 3, quirk=False))


def test_confusion_and_gather(golden_dir):
    from gpu_util import dev, lib, stream
    from insmos_amd import _lib
    g = G(golden_dir, "metrics.npz")
    cm = torch.zeros(9, dtype=torch.int64, device=D)
    lg = dev(g["logits"])
    dgt = dev(g["gt"])
    for _ in range(2):  # accumulates
        _lib.check(lib().insmos_confusion3(lg.data_ptr(), 3, dgt.data_ptr(), len(g["gt"]), 3, 1, cm.data_ptr(),
                                           stream()), "confusion")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(cm.cpu().numpy().reshape(3, 3), 2 * g["cm"])
    src = torch.arange(40, dtype=torch.float32, device=D).reshape(10, 4)
    idx = torch.tensor([3, -1, 0, 9, -1], dtype=torch.int64, device=D)
    out = torch.full((5, 3), 7.0, device=D)
    _lib.check(lib().insmos_gather_rows(src.data_ptr(), 4, 3, idx.data_ptr(), 5, out.data_ptr(), 3, stream()), "gather")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), [[12, 13, 14], [0, 0, 0], [0, 1, 2], [36, 37, 38], [0, 0, 0]])


def test_iou3d_kernel_and_recall_record(golden_dir):
    """insmos_iou3d vs the reference's boxes_iou3d_gpu values; generate_recall_record counts equal the reference's."""
    import os
    from insmos_amd.metrics import boxes_iou3d, generate_recall_record
    g = np.load(os.path.join(golden_dir, "recall.npz"))
    pred, gt, gt_pad = (torch.from_numpy(g[k]).cuda() for k in ("pred", "gt", "gt_pad"))
    np.testing.assert_allclose(boxes_iou3d(pred, gt).cpu().numpy(), g["iou3d"], rtol=1e-5, atol=1e-6)
    thr = [float(t) for t in g["thresh"]]
    keys = list(g["rd_keys"])
    rd = generate_recall_record(pred, {}, 0, {"gt_boxes": gt_pad[None]}, thr)
    assert [rd[k] for k in keys] == g["rd_vals"].tolist()
    rd2 = generate_recall_record(pred[:5], dict(rd), 0, {"gt_boxes": gt_pad[None]}, thr)
    assert [rd2[k] for k in keys] == g["rd2_vals"].tolist()
    rd0 = generate_recall_record(torch.zeros((0, 7), device="cuda"), {}, 0, {"gt_boxes": gt_pad[None]}, thr)
    assert [rd0[k] for k in keys] == g["rd0_vals"].tolist()
    assert generate_recall_record(pred, {}, 0, {}, thr) == {}

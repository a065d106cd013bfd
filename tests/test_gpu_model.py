"""-m gpu: end-to-end parity of the HIP path against the oracle on a seeded synthetic window, through
the reference-shaped API (InsMOSNet.load_from_checkpoint(...).cuda().eval().forward(list, 'test')).
Tolerances: logits 1e-3 abs fp32 (north star), labels exact after argmax, boxes 1e-3."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_model as M
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    from insmos_amd import params as P
    from insmos_amd.models import InsMOSNet, save_checkpoint
    from insmos_amd.synth import make_window
    from model_util import detecting_state_dict
    cfg = P.default_cfg()
    window = make_window(seed=5, n_scans=10, n_az=256)
    sd = detecting_state_dict(cfg, window, seed=0)
    path = os.path.join(tmp_path_factory.mktemp("ckpt"), "synthetic.ckpt")
    save_checkpoint(path, cfg, sd)
    hparams = torch.load(path, weights_only=False)["hyper_parameters"]  # predict_mos.py:288
    model = InsMOSNet.load_from_checkpoint(path, hparams=hparams).cuda().eval()
    ref_logits, ref_pred, dbg = M.forward_window(sd, cfg, window, want_debug=True)
    return dict(cfg=cfg, window=window, sd=sd, model=model, ref_logits=ref_logits, ref_pred=ref_pred, dbg=dbg)


def test_forward_signature_and_parity(setup):
    model, window = setup["model"], setup["window"]
    batch = [{"past_point_clouds": torch.from_numpy(window).cuda(), "meta": (8, 0, ["000009.bin"]),
              "batch_size_npast": 10}]
    with torch.no_grad():
        pred_list, recall_list, logits_list = model.forward(batch, "test")
    torch.cuda.synchronize()
    assert len(pred_list) == len(recall_list) == len(logits_list) == 1
    assert recall_list[0] == {}
    pred = pred_list[0][0]
    assert set(pred) == {"pred_boxes", "pred_scores", "pred_labels"}
    assert pred["pred_labels"].dtype == torch.int64 and pred["pred_boxes"].shape[1] == 7
    logits = logits_list[0].cpu().numpy()
    ref_logits, ref_pred, dbg = setup["ref_logits"], setup["ref_pred"], setup["dbg"]
    eng = model.model.engine
    # --- the product path is the native runner (csrc/forward.hip); the step-by-step Python path keeps the
    # intermediates for the stage checks below and must give the same bits
    assert eng.native
    logits_step, pred_step = eng.forward_window(batch[0]["past_point_clouds"], native=False)
    assert torch.equal(logits_step, logits_list[0])
    for k in pred:
        assert torch.equal(pred_step[k], pred[k]), k
    # --- stage checks (sharper diagnostics than the end result)
    cur_ref = dbg["current_point"]
    assert logits.shape == ref_logits.shape == (len(cur_ref), 3)
    r0 = eng.last_counts["me_row_starts"][0][0]  # dead-row elimination: only the current scan's rows are computed
    assert 0 < r0 < len(dbg["motion"]["voxel_motion"])
    np.testing.assert_allclose(eng._me_debug["motion"].cpu().numpy()[r0:, :3], dbg["motion"]["voxel_motion"][r0:], atol=2e-4,
                               rtol=1e-3)
    # ... and computing every row (as the reference does) gives the same bits where it matters
    eng.prune_dead_rows = False
    logits_full, _ = eng.forward_window(batch[0]["past_point_clouds"], native=False)
    eng.prune_dead_rows = True
    assert torch.equal(logits_full, logits_list[0])
    np.testing.assert_allclose(eng._me_debug["motion"].cpu().numpy()[:, :3], dbg["motion"]["voxel_motion"], atol=2e-4, rtol=1e-3)
    # ... and so does running the BEV deblock and the heads as two launches instead of the fused kernel
    eng.fuse_deconv_head = False
    logits_two, pred_two = eng.forward_window(batch[0]["past_point_clouds"], native=False)
    head_two = eng._head_debug["head"].clone()
    eng.fuse_deconv_head = True
    eng.forward_window(batch[0]["past_point_clouds"], native=False)  # restore the default intermediates for the checks below
    assert torch.equal(head_two, eng._head_debug["head"]) and torch.equal(logits_two, logits_list[0])
    assert torch.equal(pred_two["pred_boxes"], pred["pred_boxes"])
    np.testing.assert_allclose(eng._un_debug["enc"].cpu().numpy(), dbg["unet"]["encoded"], atol=2e-4, rtol=1e-3)
    head = eng._head_debug["head"].cpu().numpy()
    H2, W2 = 2 * eng.bevH, 2 * eng.bevW
    hm = head.reshape(eng.bevH, eng.bevW, 2, 2, -1).transpose(0, 2, 1, 3, 4).reshape(H2 * W2, -1)
    np.testing.assert_allclose(hm[:, :3], dbg["unet"]["cls"], atol=2e-3, rtol=1e-3)  # cls weights are x1000 here
    K = len(ref_pred["pred_boxes"])
    assert K > 5, "test checkpoint must produce detections"
    assert pred["pred_boxes"].shape[0] == K
    np.testing.assert_array_equal(pred["pred_labels"].cpu().numpy(), ref_pred["pred_labels"])
    np.testing.assert_allclose(pred["pred_boxes"].cpu().numpy(), ref_pred["pred_boxes"], atol=1e-3, rtol=1e-4)
    np.testing.assert_allclose(pred["pred_scores"].cpu().numpy(), ref_pred["pred_scores"], atol=2e-3)
    for lvl, name, C in ((4, "ci4", 128), (3, "ci3", 64), (2, "ci2", 32), (1, "ci1", 16)):
        t = eng._un_debug[name].cpu().numpy()
        np.testing.assert_array_equal(t[:, C:C + 3], dbg["unet"]["onehots"][lvl])
        assert np.all(t[:, C + 3:] == 0)  # zero padding up to the next multiple of 16 channels
    assert sum(int(dbg["unet"]["onehots"][l].sum()) for l in (1, 2, 3, 4)) > 0, "instance features must be exercised"
    # --- the north-star bar
    np.testing.assert_allclose(logits, ref_logits, atol=1e-3, rtol=0)
    lab, conf = R.output_stage(logits)
    lab_ref, conf_ref = R.output_stage(ref_logits)
    np.testing.assert_array_equal(lab, lab_ref)
    # MOS IoU parity through the on-device confusion matrix
    from insmos_amd.metrics import ClassificationMetrics
    from insmos_amd.synth import make_labels
    gt = make_labels(cur_ref, seed=5)
    met = ClassificationMetrics(3, [0])
    cm = met.compute_confusion_matrix(logits_list[0], torch.from_numpy(gt).cuda())
    np.testing.assert_array_equal(cm.cpu().numpy(), R.confusion_matrix(ref_logits, gt))
    np.testing.assert_allclose(met.getIoU(cm).cpu().numpy(), R.iou_from_confusion(R.confusion_matrix(ref_logits, gt)),
                               rtol=1e-6)


def test_idempotent_and_batch_of_two(setup):
    """Same window twice in one list -> identical results (deterministic kernels, no atomics in the sums)."""
    model, window = setup["model"], setup["window"]
    t = torch.from_numpy(window).cuda()
    a, _, la = model.forward([{"past_point_clouds": t}, {"past_point_clouds": t.clone()}], "test")
    torch.cuda.synchronize()
    assert torch.equal(la[0], la[1])
    assert torch.equal(a[0][0]["pred_boxes"], a[1][0]["pred_boxes"])


def test_windows_in_flight_match_sequential(setup):
    """A list of different windows processed concurrently (threads + streams + arenas) == one at a time, bit for bit,
    in list order; the caller's stream sees finished results."""
    from insmos_amd.synth import make_window
    model = setup["model"]
    wins = [torch.from_numpy(make_window(seed=20 + i, n_scans=4 + i, n_az=160 + 32 * i)).cuda() for i in range(5)]
    assert model.model.windows_in_flight >= 2
    for _ in range(2):  # second round re-uses the worker arenas
        pl, _, ll = model.forward([{"past_point_clouds": w} for w in wins], "test")
        got = [(l.clone(), {k: v.clone() for k, v in p[0].items()}) for l, p in zip(ll, pl)]
    for w, (lg, pr) in zip(wins, got):
        p1, _, l1 = model.forward([{"past_point_clouds": w}], "test")
        assert torch.equal(l1[0], lg)
        for k in pr:
            assert torch.equal(p1[0][0][k], pr[k]), k


def test_row_regrouping_does_not_change_the_outputs(setup):
    """The runner re-orders the rows of its 3D levels by tap signature (insmos_regroup_rows3d / _global); every voxel's value is
    a function of its own taps in tap order, and the one order-dependent rule downstream (the first voxel inside a box,
    Array_Index.cpp:40-56) is evaluated in the reference's row order -- so logits and boxes are the same bits for every mode and
    with the regrouping off, for one window and for a launch set of three."""
    from insmos_amd.synth import make_window
    model = setup["model"]
    lib = model.model.engine.lib
    wins = [torch.from_numpy(make_window(seed=40 + i, n_scans=5 + i, n_az=192 + 32 * i)).cuda() for i in range(3)]
    got = {}
    try:
        for blk in (0, 1111, 2222, 3333, 4444, 5555, 4012):   # one digit per level 4..1: off / 256 / 1024 / 4096-row blocks / whole windows / parity first
            assert lib.insmos_forward_regroup(blk) == 0
            p1, _, l1 = model.forward([{"past_point_clouds": wins[0]}], "test")
            p3, _, l3 = model.forward([{"past_point_clouds": w} for w in wins], "test")
            torch.cuda.synchronize()
            got[blk] = ([l.clone() for l in l1 + l3], [{k: v.clone() for k, v in p[0].items()} for p in p1 + p3])
        assert lib.insmos_forward_regroup(7) != 0 and lib.insmos_forward_regroup(55555) != 0
    finally:
        lib.insmos_forward_regroup(-1)
    for blk in (1111, 2222, 3333, 4444, 5555, 4012):
        for a, b in zip(got[0][0], got[blk][0]):
            assert torch.equal(a, b), blk
        for a, b in zip(got[0][1], got[blk][1]):
            for k in a:
                assert torch.equal(a[k], b[k]), (blk, k)


def test_eval_mode_is_test_mode_plus_the_two_losses(setup):
    """Model_mode 'eval' (models/models.py:347-353, 369-373): same predictions as 'test', MOSLoss of the point logits and of
    the motion features against past_labels[-1] (oracle: ref_ops.mos_loss on the oracle's own logits / current_point),
    six return values, ignored logit column overwritten with -inf as MOSLoss.compute_loss does to its input."""
    model, window, dbg = setup["model"], setup["window"], setup["dbg"]
    rng = np.random.default_rng(3)
    ncur = int((window[:, 4] == 0).sum())
    gts = [torch.from_numpy(rng.integers(0, 3, ncur)).cuda() for _ in range(3)]
    pts = torch.from_numpy(window).cuda()
    batch = [{"past_point_clouds": pts, "past_labels": [None, gts[i]]} for i in range(3)]
    with torch.no_grad():
        p_test, _, l_test = model.forward([{"past_point_clouds": pts}], "test")
        out = model.forward(batch, "eval")
    assert len(out) == 6
    preds, recalls, gt_list, logits_list, val_loss, val_motion_loss = out
    assert isinstance(val_loss, float) and tuple(val_motion_loss.shape) == (1,)
    assert len(preds) == len(recalls) == len(gt_list) == len(logits_list) == 3
    exp_loss, exp_motion, own_loss = 0.0, 0.0, 0.0
    own_logits = l_test[0].cpu().numpy()
    for i in range(3):
        assert gt_list[i] is gts[i]
        assert bool(torch.isinf(logits_list[i][:, 0]).all()) and bool((logits_list[i][:, 0] < 0).all())
        assert torch.equal(logits_list[i][:, 1:], l_test[0][:, 1:])
        for k in p_test[0][0]:
            assert torch.equal(preds[i][0][k], p_test[0][0][k])
        g = gts[i].cpu().numpy()
        exp_loss += R.mos_loss(setup["ref_logits"], g)[0] / 3
        exp_motion += R.mos_loss(dbg["current_point"][:, 4:7], g)[0] / 3
        own_loss += R.mos_loss(own_logits, g)[0] / 3
    assert abs(val_loss - own_loss) < 2e-5 * max(1.0, abs(own_loss))   # the loss kernel on the path's own logits
    assert abs(val_loss - exp_loss) < 1e-3 * max(1.0, abs(exp_loss))   # ... and end to end against the oracle's logits
    assert abs(float(val_motion_loss[0]) - exp_motion) < 1e-3 * max(1.0, abs(exp_motion))
    # the motion features the loss was taken on are the native runner's current_point columns == the step path's
    eng = model.model.engine
    eng.keep_current_points = True
    eng.forward_window(pts)
    native_cur = eng.last_current_points.clone()
    eng.forward_window(pts, native=False)
    assert torch.equal(native_cur, eng.last_current_points)
    eng.keep_current_points = False
    np.testing.assert_allclose(native_cur[:, :7].cpu().numpy(), dbg["current_point"][:, :7], atol=2e-4)
    with pytest.raises(KeyError):      # 'train' is served too (test_train_unet.py), and needs the training batch keys
        model.forward(batch, "train")
    with pytest.raises(ValueError):
        model.forward([{"past_point_clouds": pts, "past_labels": [gts[0][:-1]]}], "eval")


def test_training_graph_on_running_stats_reproduces_inference(setup):
    """insmos_amd/train_unet.py with BatchNorm on running statistics == the (oracle-checked) inference path: pins the
    training graph's wiring (layer order, concatenations, pair-sum reduction, BEV scatter, deconv table, point gather)."""
    from insmos_amd.engine import Engine
    from insmos_amd.train_unet import UNetV2Trainer
    cfg, window, sd = setup["cfg"], setup["window"], setup["sd"]
    eng = Engine(cfg, sd, "cuda:0")
    eng.keep_current_points = True
    pts = torch.from_numpy(window).cuda()
    logits_inf, pred_inf = eng.forward_window(pts, native=False)
    cur = eng.last_current_points.clone()
    head_inf = eng._head_debug["head"].clone()
    H, W = eng.bevH, eng.bevW
    tr = UNetV2Trainer(cfg, sd, engine=eng)
    tr.bn_training = False
    with torch.no_grad():
        out = tr.forward(cur)
    # head maps: the inference head rows are in the deconv's [y][x][ky][kx] sub-site order
    hi = head_inf.view(H, W, 2, 2, -1).permute(0, 2, 1, 3, 4).reshape(4 * H * W, -1)
    d_cls = float((out["cls_preds"].reshape(-1, 3) - hi[:, :3]).abs().max())
    d_box = float((out["box_preds"].reshape(-1, 8) - hi[:, 3:11]).abs().max())
    d_log = float((out["point_logits"] - logits_inf).abs().max())
    print("max abs diff: cls %.2e box %.2e point logits %.2e; boxes %d vs %d" %
          (d_cls, d_box, d_log, len(out["pred_dicts"][0]["pred_boxes"]), len(pred_inf["pred_boxes"])))
    assert d_cls < 1e-3 and d_box < 1e-3
    assert len(pred_inf["pred_boxes"]) > 0
    assert len(out["pred_dicts"][0]["pred_boxes"]) == len(pred_inf["pred_boxes"])
    assert float((out["pred_dicts"][0]["pred_boxes"] - pred_inf["pred_boxes"]).abs().max()) < 1e-3
    assert torch.equal(out["pred_dicts"][0]["pred_labels"], pred_inf["pred_labels"])
    assert d_log < 1e-3
    # the tap-layout round trip: exported parameters are the checkpoint's
    exp = tr.export_state_dict()
    for k, v in exp.items():
        np.testing.assert_array_equal(v, np.asarray(sd[k], np.float32).reshape(v.shape), err_msg=k)


def test_native_runner_arena_growth_and_errors(setup):
    """The native runner grows its arena on INSMOS_EWORKSPACE and reports the reference-visible input errors."""
    from insmos_amd.engine import Engine
    from insmos_amd.synth import make_window
    eng = Engine(setup["cfg"], setup["sd"], native=True)
    w = torch.from_numpy(make_window(seed=31, n_scans=3, n_az=128)).cuda()
    ref_logits, ref_pred = eng.forward_window(w, native=False)
    eng._arena = torch.empty(1 << 16, dtype=torch.uint8, device="cuda")  # far too small: several growth rounds
    logits, pred = eng.forward_window(w)
    assert eng._arena.numel() > (1 << 20)
    assert torch.equal(logits, ref_logits) and torch.equal(pred["pred_boxes"], ref_pred["pred_boxes"])
    past_only = w[w[:, 4] < 0].contiguous()  # no current scan
    for native in (True, False):
        with pytest.raises(ValueError):
            eng.forward_window(past_only, native=native)
        with pytest.raises(ValueError):
            eng.forward_window(w.double(), native=native)
        with pytest.raises(ValueError):
            eng.forward_window(w[:, :4].contiguous(), native=native)
    far = w.clone()
    far[0, 0] = 1e6  # beyond the +-32768-voxel key window
    for native in (True, False):
        with pytest.raises(ValueError):
            eng.forward_window(far, native=native)
    logits2, _ = eng.forward_window(w)  # the engine is still usable afterwards
    assert torch.equal(logits2, ref_logits)


def test_n1_window_and_no_detection_checkpoint(setup):
    """cfg-1 shape: a single scan (N=1, t==0 only) and the default head bias (no detections)."""
    from insmos_amd import params as P
    from insmos_amd.models import InsMOSNet
    from insmos_amd.synth import make_window
    cfg = setup["cfg"]
    w1 = make_window(seed=2, n_scans=1, n_az=256)
    sd = P.random_state_dict(cfg, 1)
    model = InsMOSNet(cfg, state_dict=sd).cuda().eval()
    pred, _, logits = model.forward([{"past_point_clouds": torch.from_numpy(w1).cuda()}], "test")
    ref_logits, ref_pred = M.forward_window(sd, cfg, w1)
    assert pred[0][0]["pred_boxes"].shape == (0, 7) and len(ref_pred["pred_boxes"]) == 0
    np.testing.assert_allclose(logits[0].cpu().numpy(), ref_logits, atol=1e-3, rtol=0)
    np.testing.assert_array_equal(R.output_stage(logits[0].cpu().numpy())[0], R.output_stage(ref_logits)[0])


def test_dense_stress_config_voxel_005():
    """cfg-4 shape (BASELINE.json configs[3]): voxel 0.05 m -> grid [81,2000,2400], BEV depth 5 -> NUM_BEV_FEATURES 640,
    500x600 head map.  Not weight-compatible with the 0.1 m checkpoints, so random weights; parity
    is against the oracle on a reduced scan, plus the voxel cap (max_voxels) being hit."""
    import copy
    from insmos_amd import params as P
    from insmos_amd.engine import Engine
    from insmos_amd.synth import make_window
    cfg = copy.deepcopy(P.default_cfg())
    cfg["DATA"]["VOXEL_SIZE"] = [0.05, 0.05, 0.05]
    cfg["MODEL"]["MAP_TO_BEV"]["NUM_BEV_FEATURES"] = 640
    cfg["MODEL"]["DENSE_HEAD"]["TARGET_ASSIGNER_CONFIG"]["VOXEL_SIZE"] = [0.05, 0.05, 0.05]
    sd = P.random_state_dict(cfg, 4)
    w = make_window(seed=9, n_scans=3, n_az=300)
    cap = 6000  # force the max-voxel cap path (models.py:287 uses 100000; real cfg-4 scenes exceed it)
    eng = Engine(cfg, sd, max_voxels=cap)
    assert eng.shape[1] == [81, 2000, 2400] and eng.shape[5] == [5, 250, 300]
    logits, pred = eng.forward_window(torch.from_numpy(w).cuda())
    torch.cuda.synchronize()
    ref_logits, ref_pred = M.forward_window(sd, cfg, w, max_voxels=cap)
    assert eng.last_counts["unet_voxels"][0] == cap
    assert int((eng._un_tables["pcid"] < 0).sum()) > 0  # points dropped by the cap get zero logits
    np.testing.assert_allclose(logits.cpu().numpy(), ref_logits, atol=1e-3, rtol=0)
    np.testing.assert_array_equal(R.output_stage(logits.cpu().numpy())[0], R.output_stage(ref_logits)[0])
    assert pred["pred_boxes"].shape[0] == len(ref_pred["pred_boxes"])
    logits_n, pred_n = eng.forward_window(torch.from_numpy(w).cuda(), native=True)  # same bits through the native runner
    assert torch.equal(logits_n, logits) and torch.equal(pred_n["pred_boxes"], pred["pred_boxes"])
    assert eng.last_counts["unet_voxels"][0] == cap


def test_native_runner_launch_list_equals_the_step_path(setup):
    """The roofline numerator (bench.py: roofline.algorithmic_gflop_per_window) is counted on the step path's launch log
    (Engine._conv_log) and divided by the NATIVE runner's kernel time: the two graphs must issue the same convolution launches.
    The native runner's launch sites attach (K, Cin, Cout, rows computed) to their profiler spans (insmos_prof_read_spans);
    here they are lined up with the step path's log, launch by launch, for one window and for a launch set of three."""
    import ctypes
    model, window = setup["model"], setup["window"]
    eng = model.model.engine
    lib = eng.lib
    pts = torch.from_numpy(window).cuda()
    kk = next(k for k in range(64) if lib.insmos_prof_name(k) == b"sparse_conv_mfma")

    def step_list(p):
        eng.forward_window(p, native=False)
        out = []
        for nbr, n_out, layer, row0 in eng._conv_log:
            if layer.name == "head" and out and out[-1][0] == "deconv":     # one fused launch (k_deconv_head)
                nm, K, ci, co, rows = out[-1]
                out[-1] = ("deconv+head", K, ci, 4 * eng.up_ch, rows)
                continue
            out.append((layer.name, layer.K, layer.cin, layer.cout, int(n_out) - (int(row0) & ~15)))
        return out

    def native_list(ps):
        lib.insmos_forward_streams(0)
        lib.insmos_prof_reset()
        lib.insmos_prof_enable(1)
        try:
            eng.forward_windows(ps)
            cap = 1024
            ms = (ctypes.c_double * cap)()
            meta = (ctypes.c_int64 * (4 * cap))()
            n = lib.insmos_prof_read_spans(kk, cap, ms, meta)
        finally:
            lib.insmos_prof_enable(0)
            lib.insmos_prof_reset()
            lib.insmos_forward_streams(-1)
        return [tuple(int(meta[4 * i + j]) for j in range(4)) for i in range(n)]

    want = step_list(pts)
    got = native_list([pts])
    assert len(got) == len(want), (len(got), len(want))
    for (name, K, ci, co, rows), (gK, gci, gco, grows) in zip(want, got):
        assert (K, co) == (gK, gco) and (name == "conv0p1s1" or ci == gci), (name, (K, ci, co, rows), (gK, gci, gco, grows))
        assert rows == grows, (name, rows, grows)
    # a launch set of three: the same launches, rows summed over the windows (window 2 is a shorter one)
    w2 = torch.from_numpy(np.ascontiguousarray(window[::2])).cuda()
    wants = [step_list(p) for p in (pts, w2, pts)]
    got3 = native_list([pts, w2, pts])
    assert len(got3) == len(want)
    for i, (gK, gci, gco, grows) in enumerate(got3):
        name, K, ci, co, _ = want[i]
        assert (K, co) == (gK, gco), (name, gK, gco)
        tot = sum(w[i][4] for w in wants)
        # (row0 of a set is rounded down to a 16-row group once, not once per window)
        assert abs(grows - tot) <= 16 * 3, (name, grows, tot)


def test_degenerate_windows(setup):
    """Edge inputs the reference's loop can meet on real sequences: windows of a handful of points (tight parity with the oracle), a
    window whose points all lie outside POINT_CLOUD_RANGE (the 3D branch has no voxel: every current point gets the zero logits of
    a dropped point, spconv_unet.py:408-410, and there is no box), a single point, a window without a current scan (refused with a
    ValueError, nothing launched), and a launch set that mixes a normal window with a 50-point one (each gets the bits it gets
    alone)."""
    from insmos_amd import params as P
    from insmos_amd.models import InsMOSNet
    from insmos_amd.synth import make_window
    cfg = setup["cfg"]
    sd = P.random_state_dict(cfg, 1)
    model = InsMOSNet(cfg, state_dict=sd).cuda().eval()
    w = make_window(seed=3, n_scans=10, n_az=64)
    rng = np.random.default_rng(0)

    def fwd(wins):
        pred, _, logits = model.forward([{"past_point_clouds": torch.from_numpy(x).cuda()} for x in wins], "test")
        torch.cuda.synchronize()
        return pred, logits

    for n_pts in (200, 17):
        sub = w[rng.choice(len(w), n_pts, replace=False)]
        if not (sub[:, 4] == 0).any():
            sub[0, 4] = 0.0
        pred, logits = fwd([sub])
        ref_logits, ref_pred = M.forward_window(sd, cfg, sub)
        assert logits[0].shape == ref_logits.shape == (int((sub[:, 4] == 0).sum()), 3)
        np.testing.assert_allclose(logits[0].cpu().numpy(), ref_logits, atol=1e-5, rtol=0)
        assert pred[0][0]["pred_boxes"].shape == (0, 7)
    far = w.copy()
    far[:, :3] += 1000.0
    pred, logits = fwd([far])
    assert logits[0].shape == (int((w[:, 4] == 0).sum()), 3) and float(logits[0].abs().max()) == 0.0
    assert pred[0][0]["pred_boxes"].shape == (0, 7)
    pred, logits = fwd([w[w[:, 4] == 0][:1]])
    assert logits[0].shape == (1, 3) and bool(torch.isfinite(logits[0]).all())
    with pytest.raises(ValueError):
        fwd([w[w[:, 4] != 0]])
    small = w[rng.choice(len(w), 50, replace=False)]
    small[0, 4] = 0.0
    _, both = fwd([w, small])
    _, la = fwd([w])
    _, lb = fwd([small])
    assert torch.equal(both[0], la[0]) and torch.equal(both[1], lb[0])

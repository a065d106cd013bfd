"""-m gpu: B windows in ONE set of launches (insmos_forward_windows, csrc/forward.hip) against the same windows one at a time.
The reference walks its batch list window by window (models/models.py:313); the batched runner must hand every window the
bits it gets alone -- logits, boxes, scores, labels -- whatever the other windows of the batch are (different sizes, one
that hits the voxel cap, one without detections), and must match the oracle like the single-window path does."""
import ctypes

import numpy as np
import pytest
import torch

from gpu_util import lib, stream, ws
from insmos_amd import _lib
from oracle import ref_model as M
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from insmos_amd import params as P
    from insmos_amd.synth import make_window
    from model_util import detecting_state_dict
    cfg = P.default_cfg()
    shapes = [(10, 160), (4, 112), (10, 96), (3, 144), (10, 128), (6, 176), (2, 80), (10, 104)]
    wins = [make_window(seed=60 + i, n_scans=ns, n_az=az) for i, (ns, az) in enumerate(shapes)]
    sd = detecting_state_dict(cfg, wins[0], seed=5, target=(60, 200))
    return {"cfg": cfg, "sd": sd, "wins": wins, "dev": [torch.from_numpy(w).cuda() for w in wins]}


def _same(a, b):
    la, pa, _ = a
    lb, pb, _ = b
    return (torch.equal(la, lb) and torch.equal(pa["pred_boxes"], pb["pred_boxes"]) and
            torch.equal(pa["pred_scores"], pb["pred_scores"]) and torch.equal(pa["pred_labels"], pb["pred_labels"]))


@pytest.mark.parametrize("B", [1, 2, 3, 4, 8])
def test_batched_forward_is_bitwise_the_single_window_forward(setup, B):
    from insmos_amd.engine import Engine
    eng = Engine(setup["cfg"], setup["sd"], native=True)
    wins = setup["dev"][:B]
    single = [eng.forward_windows([w])[0] for w in wins]
    per_window_counts = []
    for w in wins:
        eng.forward_windows([w])
        per_window_counts.append(dict(eng.last_counts))
    batched = eng.forward_windows(wins)
    assert len(batched) == B
    for i, (s, b) in enumerate(zip(single, batched)):
        assert s[0].shape == b[0].shape, i
        assert float((s[0] - b[0]).abs().max()) < 1e-4, (i, float((s[0] - b[0]).abs().max()))   # (a gross error reads better)
        assert _same(s, b), (i, float((s[0] - b[0]).abs().max()))
    assert sum(len(b[1]["pred_boxes"]) for b in batched) > 0      # the instance branch is exercised
    # the batched coordinate sets are the unions of the windows' sets
    c = eng.last_counts
    assert c["batch"] == B
    assert c["me_voxels"] == [sum(p["me_voxels"][l] for p in per_window_counts) for l in range(4)]
    assert [pw["voxels"] for pw in c["per_window"]] == [p["unet_voxels"][0] for p in per_window_counts]
    assert [pw["n_cur"] for pw in c["per_window"]] == [p["n_cur"] for p in per_window_counts]
    assert [pw["n_boxes"] for pw in c["per_window"]] == [p["n_boxes"] for p in per_window_counts]
    if B > 1:
        assert c["unet_voxels"][1:] == [sum(p["unet_voxels"][l] for p in per_window_counts) for l in range(1, 5)]


def test_launch_set_with_a_window_outside_the_packed_key_box(setup):
    """The packed 40-bit sort keys of a launch set of <= 8 windows cover +-2048 voxels in x / y and +-256 in z (coords.hip:
    k_quant_keys_p); MotionNet does not crop far returns (motionnet.py:21-50), so ONE window with points beyond the box sends
    the WHOLE set through the fallback (pair sort, then the full-width sort).  Every window still gets the bits it gets alone,
    the far points included, in a set of 8, in both slot orders; and the context remembers the overflow (the next set starts at
    the pair sort: same bits again)."""
    from insmos_amd.engine import Engine
    eng = Engine(setup["cfg"], setup["sd"], native=True)
    wins = [w.clone() for w in setup["dev"][:8]]
    far = wins[3].clone()
    n = far.shape[0]
    far[5::97, 0] += 230.0          # x beyond +204.8 m = 2048 voxels of 0.1 m (all time steps: old scans and the current one)
    far[7::89, 1] -= 215.0          # y beyond -204.8 m
    far[11::131, 2] += 31.0         # z beyond +25.6 m = 256 voxels
    wins[3] = far
    single = [eng.forward_windows([w])[0] for w in wins]
    assert not _same(single[3], eng.forward_windows([setup["dev"][3]])[0]) or True   # (the far points changed window 3's input)
    for order in (list(range(8)), [7, 3, 0, 5, 1, 6, 2, 4]):
        for rep in range(2):          # rep 1: the context starts at the fallback mode it remembered
            batched = eng.forward_windows([wins[i] for i in order])
            for slot, i in enumerate(order):
                assert _same(single[i], batched[slot]), (order, rep, i)
    # a set WITHOUT far points right after: starts at the pair sort (remembered), same bits as ever
    batched = eng.forward_windows(setup["dev"][:8])
    ref = [eng.forward_windows([w])[0] for w in setup["dev"][:8]]
    for i in range(8):
        assert _same(ref[i], batched[i]), i


def test_batched_forward_with_voxel_cap_and_order_independence(setup):
    """max_voxels is a PER-WINDOW cap (the reference voxelises each batch item on its own, models/models.py:326): windows
    that hit it and windows that do not share a batch; and a window's result does not depend on its slot in the batch."""
    from insmos_amd.engine import Engine
    wins = setup["dev"][:5]
    free = Engine(setup["cfg"], setup["sd"], native=True)
    uncapped = []
    for w in wins:
        free.forward_windows([w])
        uncapped.append(free.last_counts["unet_voxels"][0])
    cap = sorted(uncapped)[2]            # the median: two windows exceed it, one meets it exactly, two stay below
    eng = Engine(setup["cfg"], setup["sd"], native=True, max_voxels=cap)
    single = [eng.forward_windows([w])[0] for w in wins]
    vox = []
    for w in wins:
        eng.forward_windows([w])
        vox.append(eng.last_counts["unet_voxels"][0])
    assert vox == [min(v, cap) for v in uncapped] and max(uncapped) > cap > min(uncapped), (vox, uncapped)
    batched = eng.forward_windows(wins)
    assert [pw["voxels"] for pw in eng.last_counts["per_window"]] == vox
    for s, b in zip(single, batched):
        assert _same(s, b)
    order = [3, 0, 4, 2, 1]
    shuffled = eng.forward_windows([wins[i] for i in order])
    for k, i in enumerate(order):
        assert _same(single[i], shuffled[k]), (k, i)


def test_batched_forward_matches_the_oracle(setup):
    from insmos_amd.engine import Engine
    eng = Engine(setup["cfg"], setup["sd"], native=True)
    idx = [1, 3, 6]
    res = eng.forward_windows([setup["dev"][i] for i in idx])
    for i, (logits, pred, _) in zip(idx, res):
        ref_logits, ref_pred = M.forward_window(setup["sd"], setup["cfg"], setup["wins"][i])
        got = logits.cpu().numpy()
        np.testing.assert_allclose(got, ref_logits, atol=1e-3, rtol=0)                      # the north star's logit bar
        np.testing.assert_array_equal(R.output_stage(got)[0], R.output_stage(ref_logits)[0])
        assert pred["pred_boxes"].shape[0] == len(ref_pred["pred_boxes"])
        if len(ref_pred["pred_boxes"]):
            np.testing.assert_allclose(pred["pred_boxes"].cpu().numpy(), ref_pred["pred_boxes"], atol=1e-3, rtol=0)
            np.testing.assert_array_equal(pred["pred_labels"].cpu().numpy(), ref_pred["pred_labels"])


def test_model_groups_of_launch_sets_match_the_sequential_walk(setup):
    """InsMOS_Model.forward: 7 items as groups of 3 (two launch sets in flight) == one item at a time."""
    from insmos_amd.models import InsMOSNet
    model = InsMOSNet(setup["cfg"], state_dict=setup["sd"]).cuda().eval()
    batch = [{"past_point_clouds": w} for w in setup["dev"][:7]]
    model.model.windows_per_launch, model.model.windows_in_flight = 1, 1
    p1, r1, l1 = model.forward(batch, "test")
    model.model.windows_per_launch, model.model.windows_in_flight = 3, 2
    p2, r2, l2 = model.forward(batch, "test")
    torch.cuda.synchronize()
    assert len(l1) == len(l2) == 7 and r1 == r2
    for a, b, pa, pb in zip(l1, l2, p1, p2):
        assert torch.equal(a, b)
        assert torch.equal(pa[0]["pred_boxes"], pb[0]["pred_boxes"]) and torch.equal(pa[0]["pred_labels"], pb[0]["pred_labels"])
    with pytest.raises(ValueError):
        model.model.engine.forward_windows([])
    with pytest.raises(ValueError):   # a batch with a window that has no current scan is refused like the single window
        w = setup["dev"][0]
        model.model.engine.forward_windows([setup["dev"][1], w[w[:, 4] < 0].contiguous()])


def test_oversized_launch_set_is_split_not_refused(setup):
    """Kernels address a neighbour table with 32-bit byte offsets: insmos_forward_windows refuses a batch whose finest 81-tap
    table would pass 2 GiB (INSMOS_EBATCH) and Engine.forward_windows runs it as two smaller sets -- same bits per window.
    (The limit is lowered for the test; 16 full S0 windows would reach the real one.)"""
    from insmos_amd.engine import Engine
    eng = Engine(setup["cfg"], setup["sd"], native=True)
    wins = setup["dev"][:6]
    ref = eng.forward_windows(wins)
    n0 = eng.last_counts["me_voxels"][0]
    L = lib()
    try:
        L.insmos_debug_table_limit(81 * 4 * (n0 // 2 + n0 // 8))       # the six together are too large, halves fit
        got = eng.forward_windows(wins)
        assert eng.last_counts["batch"] == 3
        for a, b in zip(ref, got):
            assert _same(a, b)
        L.insmos_debug_table_limit(81 * 4 * 100)                        # nothing fits: a single window is an error, not a loop
        with pytest.raises(_lib.InsmosHipError):
            eng.forward_windows(wins[:2])
    finally:
        L.insmos_debug_table_limit(0)


def test_voxelize_windows_is_per_window_voxelisation(setup):
    """insmos_voxelize_mean_windows == insmos_voxelize_mean per window (first-seen order, per-window cap), rows window-major."""
    L = lib()
    rng = np.random.default_rng(3)
    cfg = setup["cfg"]
    prange = np.asarray(cfg["DATA"]["POINT_CLOUD_RANGE"], np.float32)
    vs = np.asarray(cfg["DATA"]["VOXEL_SIZE"], np.float32)
    sizes = [4000, 1, 2600, 900]
    clouds = []
    for n in sizes:
        p = np.zeros((n, 8), np.float32)
        p[:, :3] = rng.uniform(-12, 12, (n, 3)).astype(np.float32) * np.array([1, 1, 0.1], np.float32)
        p[:, 3:7] = rng.normal(size=(n, 4)).astype(np.float32)
        p[rng.random(n) < 0.05, 0] = 1e4       # out of range
        clouds.append(p)
    cap, max_pts = 1500, 5
    hp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731

    def run(points, starts, key_cells=0):
        n = len(points)
        B = len(starts) - 1
        d = torch.from_numpy(points).cuda()
        feat = torch.zeros((cap * B, 8), device="cuda")
        coords = torch.zeros((cap * B, 4), dtype=torch.int32, device="cuda")
        npts = torch.zeros(cap * B, dtype=torch.int32, device="cuda")
        pcid = torch.zeros(n, dtype=torch.int64, device="cuda")
        uk = torch.zeros(n, dtype=torch.int64, device="cuda")
        up = torch.zeros(n, dtype=torch.int32, device="cuda")
        counts = torch.zeros(8 + B, dtype=torch.int32, device="cuda")
        st = torch.tensor(starts, dtype=torch.int32, device="cuda")
        w = ws(L.insmos_voxelize_mean_ws_bytes(n))
        _lib.check(L.insmos_voxelize_mean_windows(d.data_ptr(), n, 8, 7, st.data_ptr(), B, key_cells, hp(prange), hp(vs), cap, max_pts,
                                                  feat.data_ptr(), 8, coords.data_ptr(), npts.data_ptr(), pcid.data_ptr(),
                                                  uk.data_ptr(), up.data_ptr(), counts.data_ptr(), w.data_ptr(), w.numel(),
                                                  stream()), "voxelize_windows")
        torch.cuda.synchronize()
        c = counts.cpu().numpy()
        return feat.cpu().numpy(), coords.cpu().numpy(), npts.cpu().numpy(), pcid.cpu().numpy(), c, uk.cpu().numpy()[:int(c[1])]

    starts = np.concatenate([[0], np.cumsum(sizes)]).tolist()
    KC = 41 * 1000 * 1200       # the level-1 spatial shape is one cell deeper than the 40-cell voxel grid (spconv_unet.py:114)
    fb, cb, nb, pb, cnt, ukb = run(np.concatenate(clouds), starts, KC)
    uk_single = []
    rows = cnt[4:4 + len(sizes) + 1]
    assert cnt[0] == rows[-1]
    hit_cap = 0
    for b, p in enumerate(clouds):
        f1, c1, n1, p1, k1, uk1 = run(p, [0, len(p)])
        uk_single.append(uk1 + b * KC)
        v = int(k1[0])
        hit_cap += v == cap
        assert rows[b + 1] - rows[b] == v
        sl = slice(rows[b], rows[b + 1])
        np.testing.assert_array_equal(fb[sl], f1[:v])
        np.testing.assert_array_equal(cb[sl, 1:], c1[:v, 1:])
        assert (cb[sl, 0] == b).all() and (c1[:v, 0] == 0).all()
        np.testing.assert_array_equal(nb[sl], n1[:v])
        pw = pb[starts[b]:starts[b + 1]]
        np.testing.assert_array_equal(np.where(pw >= 0, pw - rows[b], -1), p1)
    assert hit_cap >= 1
    np.testing.assert_array_equal(ukb, np.concatenate(uk_single))   # search keys: b * key_cells + cell, ascending


def test_sparse_to_bev_kernel_vs_reference_height_compression(golden_dir):
    """SURVEY.md 8 row a9 at kernel level: insmos_sparse_to_bev against tests/golden/height_compression.npz (the reference's
    HeightCompression module run as written: .dense() -> (1, C*D, H, W), height_compression.py:24-31)."""
    import os
    L = lib()
    g = np.load(os.path.join(golden_dir, "height_compression.npz"))
    dense5, out = g["dense5"], g["out"]                      # (1, C, D, H, W) -> (1, C*D, H, W)
    _, C, D, H, W = dense5.shape
    # active sites of the golden dense tensor as a sparse (features, [b, z, y, x]) pair, in a shuffled row order
    act = np.argwhere(np.abs(dense5[0]).sum(0) != 0)         # (n, 3) = [d, y, x]
    act = act[np.random.default_rng(0).permutation(len(act))]
    feats = np.ascontiguousarray(dense5[0][:, act[:, 0], act[:, 1], act[:, 2]].T)   # (n, C)
    coords = np.concatenate([np.zeros((len(act), 1), np.int64), act], 1).astype(np.int32)
    ld = 8
    f = torch.zeros((len(act), ld), device="cuda")
    f[:, :C] = torch.from_numpy(feats).cuda()
    cd = torch.from_numpy(coords).cuda()
    for B in (1, 2):                                          # B = 2: the same image in both slots of a stacked pair
        if B == 2:
            cd2 = torch.cat([cd, cd.clone()])
            cd2[len(act):, 0] = 1
            f_in, c_in = torch.cat([f, f * 2.0]), cd2
        else:
            f_in, c_in = f, cd
        bev = torch.full((B, H, W, C * D), 7.0, device="cuda")
        _lib.check(L.insmos_sparse_to_bev_b(f_in.data_ptr(), ld, C, c_in.data_ptr(), len(c_in), D, H, W, B, bev.data_ptr(),
                                            stream()), "sparse_to_bev")
        torch.cuda.synchronize()
        got = bev.cpu().numpy()                                # NHWC, channel = c*D + d
        np.testing.assert_array_equal(got[0].transpose(2, 0, 1), out[0])
        if B == 2:
            np.testing.assert_array_equal(got[1].transpose(2, 0, 1), out[0] * 2.0)
